"""Host side of the backbone engine (`fpc_net_*` in include/fpc.h, csrc/net.hip).

A `NetEngine` binds one PoseRegressor's parameters to a native plan for a fixed (B, H, W): the C
side repacks the weights once (OHWI, padded; BatchNorm folded), then every `forward` is a single C
call that enqueues the whole frame's kernels on torch's current stream.  torch only owns the
memory: parameters, the workspace and the output tensors.
"""
import ctypes
import operator

import torch

from fastposecnn_amd import _native as nat


_DATA_PTR = torch.Tensor.data_ptr
_VERSION = operator.attrgetter('_version')

class NetEngine:

    def __init__(self, model, B, H, W, device, autotune=True, tune_mode=0, graph=False, split_precision=False):
        L = nat.lib()
        self._lib = L
        self.B, self.H, self.W, self.device = B, H, W, device
        self.classes = model.classes
        h = ctypes.c_void_p()
        enc = model.encoder.name.encode()
        nat.check(L.fpc_net_create(enc, self.classes, B, H, W, ctypes.byref(h)), "fpc_net_create")
        self._h = h
        self._names = [L.fpc_net_param_name(h, i).decode() for i in range(L.fpc_net_param_count(h))]
        nbytes = L.fpc_net_workspace_bytes(h)
        with torch.cuda.device(device):
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            assert self._ws.data_ptr() % 256 == 0
        self.reloads = 0
        self.generation = 0          # the owner's count of unversioned writes when this plan was last bound (pose_regressor.py)
        self.bind(model)
        if split_precision:
            nat.check(L.fpc_net_set_split_precision(h, int(split_precision)), "fpc_net_set_split_precision")
        if autotune:
            # one (discarded) forward that times every candidate tiling per convolution on this device
            nat.check(L.fpc_net_autotune_next(h, int(tune_mode)), "fpc_net_autotune_next")
            self.forward(torch.zeros((B, 3, H, W), dtype=torch.float32, device=device), want_logits=False)
        if graph:
            nat.check(L.fpc_net_set_graph(h, 1), "fpc_net_set_graph")

    def bind(self, model):
        """(Re)pack the model's current parameters into the plan's workspace.  The tuned tilings are kept; a
        recorded graph is dropped by the library and re-captured on the next forward."""
        L, h, device = self._lib, self._h, self.device
        tensors = dict(model.named_parameters())
        tensors.update(dict(model.named_buffers()))
        params = []                 # keeps the tensors alive: the plan reads some of them in place
        for i, name in enumerate(self._names):
            if name not in tensors:
                raise RuntimeError(f"fastposecnn_amd: the model has no parameter {name!r} (smp naming expected)")
            t = tensors[name]
            if t.device != device or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError(f"fastposecnn_amd: parameter {name!r} must be a contiguous f32 tensor on {device}")
            if t.numel() != L.fpc_net_param_numel(h, i):
                raise RuntimeError(f"fastposecnn_amd: parameter {name!r} has {t.numel()} elements, expected "
                                   f"{L.fpc_net_param_numel(h, i)}")
            params.append(t)
        self._params = params
        n = len(params)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in params])
        with torch.cuda.device(device):
            nat.check(L.fpc_net_load_params(h, ptrs, n, self._ws.data_ptr(), self._ws.numel(), nat.stream()),
                      "fpc_net_load_params")
        self._stamp = self._fingerprint()
        self.reloads += 1

    def _fingerprint(self):
        # (storage address, in-place version) of every bound tensor: changes on load_state_dict (of the model or of
        # any sub-module), optimizer / EMA steps, p.copy_(), .to() — anything but a write through `p.data`, which
        # PyTorch itself does not version
        # (two C-level maps: a Python-level loop over the ~200 tensors cost 35 us of every streamed frame's 390 on the host)
        return (tuple(map(_DATA_PTR, self._params)), tuple(map(_VERSION, self._params)))

    def stale(self):
        """True when a bound parameter was rewritten or replaced since it was packed (packed conv weights and folded
        BatchNorm are a snapshot; biases / GroupNorm / head weights are read in place)."""
        return self._fingerprint() != self._stamp

    def conv_plans(self):
        out = []
        buf = (ctypes.c_int * 5)()
        for i in range(self._lib.fpc_net_conv_count(self._h)):
            self._lib.fpc_net_conv_plan(self._h, i, buf)
            out.append(tuple(buf))
        return out

    def copy_plans_from(self, other):
        """Run this engine on the plans `other` (same network and frame size, another batch) was autotuned to."""
        nat.check(self._lib.fpc_net_copy_plans(self._h, other._h), "fpc_net_copy_plans")

    def force_winograd(self, form):
        """Every 3x3 / stride-1 site on Winograd form `form` (8 = fp16 x 2 pieces); returns the number of sites changed."""
        rc = self._lib.fpc_net_force_winograd(self._h, int(form))
        if rc < 0:
            nat.check(rc, "fpc_net_force_winograd")
        return rc

    def flops(self):
        """(direct-convolution FLOP, FLOP the current plans execute, Winograd share) of one forward over the batch."""
        buf = (ctypes.c_double * 3)()
        nat.check(self._lib.fpc_net_flops(self._h, buf), "fpc_net_flops")
        return tuple(buf)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.fpc_net_destroy(h)
            self._h = None

    def forward(self, x, want_logits=True):
        """x f32 [B,3,H,W] -> (logits dict | None, categorical dict incl. 'mask')."""
        B, H, W, C, dev = self.B, self.H, self.W, self.classes, self.device
        G = C - 1
        if tuple(x.shape) != (B, 3, H, W) or x.device != dev:
            raise RuntimeError("NetEngine.forward: input shape / device does not match the plan")
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.float().contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        logits = None
        lp = [None] * 5
        if want_logits:
            logits = {'mask': torch.empty((B, C, H, W), **f32), 'quaternion': torch.empty((B, 4 * G, H, W), **f32),
                      'scales': torch.empty((B, 3 * G, H, W), **f32), 'xy': torch.empty((B, 2 * G, H, W), **f32),
                      'z': torch.empty((B, G, H, W), **f32)}
            lp = [logits[k].data_ptr() for k in ('mask', 'quaternion', 'scales', 'xy', 'z')]
        cat = {'mask': torch.empty((B, H, W), dtype=torch.int64, device=dev),
               'quaternion': torch.empty((B, 4, H, W), **f32), 'scales': torch.empty((B, 3, H, W), **f32),
               'xy': torch.empty((B, 2, H, W), **f32), 'z': torch.empty((B, H, W), **f32)}
        # the foreground of the class mask also as bit words (1/64 of its bytes): the connected-component labelling reads
        # those instead of the i64 mask; they ride on the mask tensor (aggregation_layer.fg_bits_of), not in the dict
        bits = None
        if W % 64 == 0 and self._lib.fpc_cc_bits_supported(B, H, W):
            bits = torch.empty((B, self._lib.fpc_mask_bits_words(H, W)), dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            nat.check(self._lib.fpc_net_forward_bits(self._h, x.data_ptr(), *lp, cat['mask'].data_ptr(),
                                                     cat['quaternion'].data_ptr(), cat['scales'].data_ptr(),
                                                     cat['xy'].data_ptr(), cat['z'].data_ptr(), nat.ptr(bits), nat.stream()),
                      "fpc_net_forward_bits")
        if bits is not None:
            cat['mask']._fpc_fg_bits = (bits, cat['mask']._version)
        return logits, cat

    def tensor(self, name):
        """Intermediate activation as an NHWC view into the workspace (tests)."""
        p = ctypes.c_void_p()
        H, W, C = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        nat.check(self._lib.fpc_net_tensor(self._h, name.encode(), ctypes.byref(p), ctypes.byref(H), ctypes.byref(W),
                                           ctypes.byref(C)), "fpc_net_tensor")
        off = p.value - self._ws.data_ptr()
        n = self.B * H.value * W.value * C.value
        return self._ws[off:off + 4 * n].view(torch.float32).view(self.B, H.value, W.value, C.value)
