"""ctypes binding of libfpc_hip.so (C ABI in include/fpc.h).

PyTorch is used for device memory and streams only: every call passes raw device
pointers (`tensor.data_ptr()`) and the current HIP stream.  There is NO fallback: if the
library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfpc_hip.so")

_lib = None

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_u64 = ctypes.c_uint64
_f = ctypes.c_float
_sz = ctypes.c_size_t

_SIGNATURES = {
    "fpc_abi_version": (ctypes.c_int, []),
    "fpc_error_string": (ctypes.c_char_p, [_i]),
    "fpc_last_hip_error": (ctypes.c_char_p, []),
    "fpc_generate_hypothesis": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "fpc_voting_for_hypothesis": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "fpc_ransac_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "fpc_ransac_voting_v3": (_i, [_vp, _vp, _i64, _i64, _i64, _i64, _i, _vp, _i, _i, _i, _vp, _vp, _u64, _f, _i, _i,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "fpc_ransac_voting_v3_bits": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i, _vp, _i, _i, _i, _vp, _vp, _u64, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "fpc_ransac_voting_v3_pose": (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i, _vp, _i, _i, _i, _vp, _vp, _u64, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "fpc_vote_set_prune": (_i, [_i, _i, ctypes.POINTER(ctypes.c_int32)]),
    "fpc_vote_prune_info": (_i, [_vp, _sz, _i, _i, _i, _i, _vp, _vp]),
    "fpc_class_compress": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fpc_class_compress_bits": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fpc_cc_workspace_bytes": (_sz, [_i, _i, _i]),
    "fpc_cc_bits_supported": (_i, [_i, _i, _i]),
    "fpc_fg_bits": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "fpc_cc_label_bits": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "fpc_cc_label": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "fpc_aggregate_workspace_bytes": (_sz, [_i]),
    "fpc_aggregate": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                           _vp]),
    "fpc_aggregate_bits": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "fpc_mask_bits_words": (_sz, [_i, _i]),
    "fpc_pose_errors": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "fpc_post_network_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _f, _i, _u64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fpc_vote_refine_backward": (_i, [_vp, _vp, _i64, _i64, _i64, _i64, _i, _i, _i, _vp, _f, _i, _u64, _vp, _vp, _vp]),
    "fpc_class_compress_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "fpc_mask_losses": (_i, [_vp, _vp, _i, _i, _i, _i64, _i64, _f, _f, _vp, _vp, _vp, _vp]),
    "fpc_grad_sumsq": (_i, [_vp, _sz, _vp, _vp]),
    "fpc_lookahead_radam_step": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _i64, _i, _f, _vp, _vp]),
    "fpc_pose_rt": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "fpc_pack_pose_records": (_i, [_vp] * 9 + [_i, _i, _i, _vp, _vp]),
    "fpc_mask_iou_workspace_bytes": (_sz, [_i, _i, _i64]),
    "fpc_mask_iou": (_i, [_vp, _i, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "fpc_preprocess_workspace_bytes": (_sz, [_i]),
    "fpc_preprocess_u8": (_i, [_vp, _i, _i, _i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _i, _vp, _vp,
                               _sz, _vp]),
    "fpc_png_info": (_i, [_vp, _sz, ctypes.POINTER(ctypes.c_int32)]),
    "fpc_png_decode": (_i, [_vp, _sz, _vp, _sz, _i]),
    "fpc_png_decode_batch": (_i, [ctypes.POINTER(_vp), ctypes.POINTER(_sz), _i, _vp, _i, _i, _i]),
    "fpc_net_create": (_i, [ctypes.c_char_p, _i, _i, _i, _i, ctypes.POINTER(_vp)]),
    "fpc_net_destroy": (None, [_vp]),
    "fpc_net_set_graph": (_i, [_vp, _i]),
    "fpc_net_set_split_precision": (_i, [_vp, _i]),
    "fpc_net_param_count": (_i, [_vp]),
    "fpc_net_param_name": (ctypes.c_char_p, [_vp, _i]),
    "fpc_net_param_numel": (_i64, [_vp, _i]),
    "fpc_net_workspace_bytes": (_sz, [_vp]),
    "fpc_net_load_params": (_i, [_vp, _vp, _i, _vp, _sz, _vp]),
    "fpc_net_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fpc_net_forward_bits": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fpc_net_autotune_next": (_i, [_vp, _i]),
    "fpc_net_conv_count": (_i, [_vp]),
    "fpc_net_conv_plan": (_i, [_vp, _i, ctypes.POINTER(_i)]),
    "fpc_net_copy_plans": (_i, [_vp, _vp]),
    "fpc_net_force_winograd": (_i, [_vp, _i]),
    "fpc_net_flops": (_i, [_vp, ctypes.POINTER(ctypes.c_double)]),
    "fpc_net_tensor": (_i, [_vp, ctypes.c_char_p, ctypes.POINTER(_vp), ctypes.POINTER(_i), ctypes.POINTER(_i),
                            ctypes.POINTER(_i)]),
    "fpc_conv2d_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "fpc_conv2d_workspace_bytes_for": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "fpc_conv2d_plan": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i, ctypes.POINTER(_i)]),
    "fpc_conv2d": (_i, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i,
                        _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "fpc_conv2d_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "fpc_upsample_bilinear_fwd": (_i, [_vp, _i64, _i64, _i64, _i64, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "fpc_upsample_bilinear_bwd_scratch_floats": (_sz, [_i, _i, _i, _i, _i]),
    "fpc_upsample_bilinear_bwd": (_i, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "fpc_groupnorm4_relu_scratch_floats": (_sz, [_i, _i, _i]),
    "fpc_groupnorm4_relu_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _vp]),
    "fpc_groupnorm4_relu_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "fpc_conv2d_wgrad": (_i, [_vp, _i64, _i64, _i64, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "fpc_conv2d_wgrad_split": (_i, [_vp, _i64, _i64, _i64, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
}

EXPORTED = tuple(_SIGNATURES)


def lib():
    """Load the HIP library or raise.  Never returns a substitute."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"fastposecnn_amd: {LIB_PATH} is missing — build it with `python -m fastposecnn_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the HIP hot path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.fpc_abi_version() != 11:
            raise RuntimeError("fastposecnn_amd: libfpc_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        L = lib()
        msg = L.fpc_error_string(rc).decode()
        hip = L.fpc_last_hip_error().decode() if rc == -3 else ""      # FPC_ELAUNCH: the HIP error of THIS call
        raise RuntimeError(f"fastposecnn_amd: {what} failed: {msg} (code {rc}){' — HIP: ' + hip if hip else ''}")


def ptr(t):
    """Raw device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """The current device's current HIP stream as the integer handle the C ABI takes.  torch.cuda.current_stream() builds a Stream
    object and resolves the device through four Python layers (~8 us; a streamed frame asks 14 times): the two C entry points torch
    itself uses for this are taken when they exist."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*tensors, what="this op"):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(f"fastposecnn_amd: {what} runs only on GPU tensors (HIP kernels, no CPU fallback)")


_workspaces = {}


def workspace(tag, device, nbytes):
    """Persistent, grow-only, 256-byte-aligned device scratch per (tag, device, stream)."""
    key = (tag, device.index, stream())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        assert ws.data_ptr() % 256 == 0
        _workspaces[key] = ws
    return ws
