"""Convolutions of the TRAINING step on the native kernels (BASELINE.json configs[4]).

The reference trains through torch.nn.Conv2d, i.e. cuDNN forward and backward under autograd
(F/lib/pose_regressor.py:709-743 inside Lightning's training_step).  Here a convolution in training mode is one
autograd.Function over three native pieces, all on channel-last (NHWC) activations:

  forward        fpc_conv2d: the inference engine's implicit-GEMM / Winograd kernels (csrc/net_kernels.hip), bias fused
  data gradient  stride 1: the SAME kernels on the flipped, transposed weights (a convolution of dy);
                 stride 2 (three encoder convolutions, their 1x1 shortcuts): the same kernels once per output parity
                 (round 4: dx[2a + r] only sees the taps of parity r — 1, 2, 2 and 4 of a 3x3 kernel's nine — so four small
                 stride-1 convolutions of dy write the four interleaved quarter planes; no zero-insertion, 13/9 of the
                 minimal multiply-adds);  heads whose width is not a multiple of 32: dy and W zero-padded to 32 channels
  weight gradient  fpc_conv2d_wgrad (csrc/conv_wgrad.hip): pixels-as-K GEMM on the f32 matrix cores, fixed-order
                 split over the pixels; a width that is not a multiple of 4 (mask / scales / xyz heads) is padded likewise
The 7x7 stem (Cin = 3) is the one convolution left to torch, forward and backward: its input needs no gradient and its
weight gradient is a K = 147 GEMM no tile of these kernels fits (counted neither as native nor as aten below: it never
enters this module's autograd functions).

The tiling / Winograd form of a shape is chosen once, by timing the candidates on the first call with that shape (this
synchronises: it happens in the warm-up steps).  f32 operands, accumulation and results, products in plain f32 or as the
exact bf16 x 3 split (FPC_SPLIT_PRECISION, default on), like the inference path.
`ENABLED = False` (or FPC_TRAIN_NATIVE_CONV=0) returns every convolution to torch's own kernels.
"""
import os

import torch

from fastposecnn_amd import _native as nat

ENABLED = bool(int(os.environ.get("FPC_TRAIN_NATIVE_CONV", "1")))
SPLIT_PRECISION = bool(int(os.environ.get("FPC_SPLIT_PRECISION", "1")))      # the bf16 x 3 product forms may be chosen
SPLIT_F16 = SPLIT_PRECISION and bool(int(os.environ.get("FPC_SPLIT_F16", "1")))      # ... and, for forward convolutions, the fp16 x 2 Winograd form
_plan_cache = {}            # (device index, B, Cin, H, W, Cout, k, stride, pad) -> nsplit code of fpc_conv2d
counters = {"fwd_native": 0, "dgrad_native": 0, "wgrad_native": 0, "dgrad_aten": 0, "wgrad_aten": 0, "wgrad_direct": 0}


# Weight-gradient sinks (round 6): an optimiser that keeps `p.grad` as views of one flat buffer (train_parallel.py) registers
# {weight data pointer -> GradSink}; the weight-gradient kernel then writes STRAIGHT into that view and the autograd function returns
# None for the weight — no temporary, no AccumulateGrad add launch per convolution (~70 of the ~270 add launches of a step), the
# sink's callback stands in for the post-accumulate hook.  A weight used twice in a step falls back to the ordinary path (the
# second gradient is returned and accumulated by autograd).
class GradSink:
    __slots__ = ("view", "callback", "written", "owner")

    def __init__(self, view, callback, owner):
        import weakref
        self.view, self.callback, self.written, self.owner = view, callback, False, weakref.ref(owner)


grad_sinks = {}


def _channels_last(t):
    return t if t.stride(1) == 1 and t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


def _run(x, w, bias, stride, pad, code, out, up=None):
    B, Cin, H, W = x.shape
    Cout, _, Kh, Kw = w.shape
    Ho, Wo = out.shape[2], out.shape[3]
    L = nat.lib()
    sb, sc, sh, sw = x.stride()
    ws = nat.workspace("train_conv", x.device, L.fpc_conv2d_workspace_bytes_for(B, Ho, Wo, Cin, Cout, Kh, Kw, 0, 0, code))
    # (the library resolves its zero page — a device global — for the CURRENT device: make that the tensor's)
    with torch.cuda.device(x.device):
        nat.check(L.fpc_conv2d(x.data_ptr(), sb, sh, sw, sc, w.data_ptr(), None, nat.ptr(bias), None, nat.ptr(up), out.data_ptr(), None,
                               B, H, W, Cin, Cout, Kh, Kw, stride, pad, 0, 0, 0, code, ws.data_ptr(), ws.numel(), nat.stream()),
                  "fpc_conv2d (training)")


def conv_nhwc(x, w, bias, stride, pad, up=None, allow_f16=False):
    """x [B,Cin,H,W] channel-last, w OIHW contiguous -> [B,Cout,Ho,Wo] channel-last, on the engine's kernels.
    up [B,Cout,Ho/2,Wo/2] channel-last contiguous: added nearest-x2 upsampled in the kernel's epilogue (the FPN top-down merge)."""
    B, Cin, H, W = x.shape
    Cout, _, Kh, Kw = w.shape
    Ho, Wo = (H + 2 * pad - Kh) // stride + 1, (W + 2 * pad - Kw) // stride + 1
    out = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    key = (x.device.index, B, Cin, H, W, Cout, Kh, stride, pad, bool(allow_f16))
    code = _plan_cache.get(key)
    if code is None:
        cands = [0] + ([1000] if SPLIT_PRECISION else [])      # heuristic tiling; the same with split-precision products
        if Kh == 3 and Kw == 3 and stride == 1 and pad == 1 and Cin % 8 == 0 and Cout % 64 == 0:
            cands += [-1, -2, -4]       # Winograd F(2x2,3x3): 4 waves, 8 waves, 8 waves all-DMA
            if SPLIT_PRECISION:
                cands.append(-5)        # 8 waves, split-precision products
                cands.append(-7)        # the same products as four waves of 512 registers, weights straight into registers (wino_w4.hip)
                if Cout % 128 == 0:
                    cands.append(-6)    # 128 output channels per workgroup (wino128.hip)
                if allow_f16 and SPLIT_F16:
                    cands.append(-8)    # two fp16 pieces per operand (wino_h2.hip): FORWARD convolutions only — their operand is an
                    #                     activation of ordinary scale; a data gradient's operand is dy, whose values are tiny
                    #                     (the three-product form, -9, measured on the batch-8 step: forward 10.3-10.5 ms against 8.6-9.3 — its
                    #                     entry stages three K-steps and sixteen weight fragments; not a candidate here)
        if SPLIT_PRECISION and Kh == 1 and Kw == 1 and stride == 1 and pad == 0 and Cin in (64, 128) and Cout % 32 == 0:
            cands += [2000 + p for p in (1, 2, 4) if (Cout // 32) % p == 0]      # pixel-resident lateral product (lateral.hip)
        best = (float("inf"), 0)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for c in cands:
            _run(x, w, bias, stride, pad, c, out)
            ev[0].record()
            for _ in range(3):
                _run(x, w, bias, stride, pad, c, out)
            ev[1].record()
            ev[1].synchronize()
            best = min(best, (ev[0].elapsed_time(ev[1]), c))
        code = _plan_cache[key] = best[1]
    _run(x, w, bias, stride, pad, code, out, up)
    return out


def _native_forward_ok(x, w):
    return w.shape[1] % 32 == 0 and w.shape[2] == w.shape[3]


def _dgrad_stride2(gy, w, H, W, pad):
    """dx [B,Cin,H,W] of a stride-2 convolution (1x1 pad 0, or 3x3 pad 1; H, W even) from dy [B,Cout,H/2,W/2]:
    y[i] = sum_k x[2 i + k - pad] w[k]  =>  dx[2 a] = dy[a] w[1],  dx[2 a + 1] = dy[a] w[2] + dy[a + 1] w[0]  per axis (3x3), i.e.
    per output parity (ry, rx) a 2-tap correlation of dy (zero past its end) with taps g_even = (w[1], 0), g_odd = (w[2], w[0])."""
    B, Cout, Ho, Wo = gy.shape
    Cin, K = w.shape[1], w.shape[2]
    gx = torch.empty((B, Cin, H, W), dtype=torch.float32, device=gy.device, memory_format=torch.channels_last)
    wt = w.transpose(0, 1)                                              # [Cin, Cout, K, K]: output channels of the dgrad first
    if K == 1:
        gx.zero_()
        gx[:, :, ::2, ::2] = conv_nhwc(gy, wt.contiguous(), None, 1, 0)
        return gx
    gyp = torch.nn.functional.pad(gy, (0, 1, 0, 1))                     # one zero row / column past the end (channel-last stays)
    gyp = _channels_last(gyp)
    taps = ((1, None), (2, 0))                                           # parity -> kernel index of tap t = 0, 1 (None: no tap)
    for ry in range(2):
        for rx in range(2):
            if ry == 0 and rx == 0:
                out = conv_nhwc(gy, wt[:, :, 1:2, 1:2].contiguous(), None, 1, 0)
            else:
                g = torch.zeros((Cin, Cout, 2, 2), dtype=torch.float32, device=gy.device)
                for ty, ky in enumerate(taps[ry]):
                    for tx, kx in enumerate(taps[rx]):
                        if ky is not None and kx is not None:
                            g[:, :, ty, tx] = wt[:, :, ky, kx]
                out = conv_nhwc(gyp, g, None, 1, 0)
            gx[:, :, ry::2, rx::2] = out
    return gx


def _conv_backward(x, w, gy, stride, pad, needs, has_bias):
    """(dx, dW, db) of a convolution whose forward ran on the native kernels; needs = (x, W, bias) wanted."""
    Cout, Cin, Kh, Kw = w.shape
    gy = _channels_last(gy)
    need_x, need_w = needs[0], needs[1]
    gx = gw = gb = None
    if has_bias and needs[2]:
        gb = gy.sum((0, 2, 3))
    # a width the kernels' tiles do not divide (the heads: 7, 18, 24 channels): zero-padded copies of dy and W — the extra
    # channels contribute nothing to dx and their rows of dW are dropped
    w_full = w
    if (need_x and Cout % 32 != 0) or (need_w and Cout % 4 != 0):
        Cp = (Cout + 31) // 32 * 32
        gyp = torch.empty((gy.shape[0], Cp, gy.shape[2], gy.shape[3]), dtype=torch.float32, device=gy.device, memory_format=torch.channels_last)
        gyp[:, Cout:] = 0.0
        gyp[:, :Cout] = gy
        wp = torch.zeros((Cp, Cin, Kh, Kw), dtype=torch.float32, device=w.device)
        wp[:Cout] = w
        gy, w = gyp, wp
    Cw = w.shape[0]
    s2_native = (stride == 2 and pad == (Kh - 1) // 2 and Kh in (1, 3) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)
    aten_x = need_x and not ((stride == 1 and pad <= Kh - 1) or s2_native)
    aten_w = need_w and Cin % 64 != 0
    if need_x and not aten_x:
        if stride == 1:
            # dx = conv(dy, W'), W'[ci][co][kh][kw] = W[co][ci][K-1-kh][K-1-kw], padding K-1-pad
            w2 = w.flip(2, 3).transpose(0, 1).contiguous() if Kh > 1 else w.transpose(0, 1).contiguous()
            gx = conv_nhwc(gy, w2, None, 1, Kh - 1 - pad)
        else:
            gx = _dgrad_stride2(gy, w, x.shape[2], x.shape[3], pad)
        counters["dgrad_native"] += 1
    if need_w and not aten_w:
        L = nat.lib()
        B, _, H, W = x.shape
        Ho, Wo = gy.shape[2], gy.shape[3]
        sink = grad_sinks.get(w.data_ptr()) if Cw == Cout else None
        if sink is not None and sink.owner() is None:        # its optimiser is gone: the address may belong to another tensor now
            del grad_sinks[w.data_ptr()]
            sink = None
        if sink is not None and (sink.written or sink.view.shape != w.shape or not sink.view.is_contiguous()):
            sink = None
        gw = sink.view if sink is not None else torch.empty_like(w)
        sb, sc, sh, sw = x.stride()
        ws = nat.workspace("train_wgrad", x.device, L.fpc_conv2d_wgrad_workspace_bytes(B, Ho, Wo, Cin, Cw, Kh, Kw))
        wgrad = L.fpc_conv2d_wgrad_split if SPLIT_PRECISION else L.fpc_conv2d_wgrad
        nat.check(wgrad(x.data_ptr(), sb, sh, sw, gy.data_ptr(), gw.data_ptr(), B, H, W, Cin, Cw, Kh, Kw, stride,
                        pad, ws.data_ptr(), ws.numel(), nat.stream()), "fpc_conv2d_wgrad")
        if Cw != Cout:
            gw = gw[:Cout].contiguous()
        counters["wgrad_native"] += 1
        if sink is not None:                # written in place: nothing for autograd to accumulate
            sink.written = True
            sink.callback()
            gw = None
            counters["wgrad_direct"] += 1
    if aten_x or aten_w:
        ax, aw, _ = torch.ops.aten.convolution_backward(gy[:, :Cout] if Cw != Cout else gy, x, w_full, None, [stride, stride], [pad, pad],
                                                        [1, 1], False, [0, 0], 1, [aten_x, aten_w, False])
        if aten_x:
            gx = ax; counters["dgrad_aten"] += 1
        if aten_w:
            gw = aw; counters["wgrad_aten"] += 1
    return gx, gw, gb


class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, pad):
        x = _channels_last(x)
        w = w.contiguous()
        ctx.stride, ctx.pad, ctx.has_bias = stride, pad, bias is not None
        ctx.save_for_backward(x, w)
        counters["fwd_native"] += 1
        return conv_nhwc(x, w, bias, stride, pad, allow_f16=True)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx, gw, gb = _conv_backward(x, w, gy, ctx.stride, ctx.pad, ctx.needs_input_grad, ctx.has_bias)
        return gx, gw, gb, None, None


class _ConvUpAddFn(torch.autograd.Function):
    """conv(x, w, bias) + nearest_x2(top): the FPN block's lateral 1x1 convolution and top-down merge
    (smp FPNBlock: F.interpolate(top, scale_factor=2, mode="nearest") + skip_conv(skip)) as ONE native launch — the
    engine's convolution epilogue adds the upsampled pyramid level, as on the inference path.  Backward: the convolution's
    as _Conv2dFn; the gradient of `top` is the 2 x 2 block sum of the output gradient."""

    @staticmethod
    def forward(ctx, x, w, bias, top, pad):
        x = _channels_last(x)
        w = w.contiguous()
        top = _channels_last(top)
        ctx.pad, ctx.has_bias = pad, bias is not None
        ctx.save_for_backward(x, w)
        counters["fwd_native"] += 1
        return conv_nhwc(x, w, bias, 1, pad, up=top)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = _channels_last(gy)
        gx, gw, gb = _conv_backward(x, w, gy, 1, ctx.pad, ctx.needs_input_grad, ctx.has_bias)
        gt = torch.nn.functional.avg_pool2d(gy, 2, divisor_override=1) if ctx.needs_input_grad[3] else None
        return gx, gw, gb, gt, None


def conv2d_up_add(x, w, bias, top, pad):
    """conv(x) + nearest-x2(top) for the FPN merge; None when the shapes do not fit the fused form (the caller keeps the
    separate ops)."""
    if not (ENABLED and x.is_cuda and x.dtype == torch.float32 and top.dtype == torch.float32 and _native_forward_ok(x, w)):
        return None
    Cout = w.shape[0]
    Ho, Wo = x.shape[2] + 2 * pad - w.shape[2] + 1, x.shape[3] + 2 * pad - w.shape[3] + 1
    if Cout % 4 != 0 or Ho % 2 or Wo % 2 or tuple(top.shape) != (x.shape[0], Cout, Ho // 2, Wo // 2):
        return None
    return _ConvUpAddFn.apply(x, w, bias, top, int(pad))


def conv2d(x, w, bias, stride, pad):
    """Training-mode convolution: native when the shapes allow it, torch otherwise (same signature either way)."""
    if ENABLED and x.is_cuda and x.dtype == torch.float32 and _native_forward_ok(x, w):
        return _Conv2dFn.apply(x, w, bias, int(stride), int(pad))
    if ENABLED and x.is_cuda and x.dim() == 4:
        x = _channels_last(x)        # the stem: keeps the rest of the network channel-last
    return torch.nn.functional.conv2d(x, w, bias, stride, pad)


# ---- bilinear upsampling (align_corners = True): the x2 of the segmentation blocks, the x4 of the heads ----------------

class _UpBilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, out_nhwc):
        B, C, h, w = x.shape
        fmt = torch.channels_last if out_nhwc else torch.contiguous_format
        out = torch.empty((B, C, h * scale, w * scale), dtype=torch.float32, device=x.device, memory_format=fmt)
        sb, sc, sh, sw = x.stride()
        L = nat.lib()
        nat.check(L.fpc_upsample_bilinear_fwd(x.data_ptr(), sb, sc, sh, sw, out.data_ptr(), B, C, h, w, scale, int(out_nhwc), nat.stream()),
                  "fpc_upsample_bilinear_fwd")
        ctx.scale, ctx.shape = scale, (B, C, h, w)
        ctx.in_nhwc = C > 1 and x.stride(1) == 1 and x.is_contiguous(memory_format=torch.channels_last)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, h, w = ctx.shape
        fmt = torch.channels_last if ctx.in_nhwc else torch.contiguous_format
        gx = torch.empty((B, C, h, w), dtype=torch.float32, device=g.device, memory_format=fmt)
        sb, sc, sh, sw = g.stride()
        L = nat.lib()
        ws = nat.workspace("train_up_bwd", g.device, 4 * L.fpc_upsample_bilinear_bwd_scratch_floats(B, C, h, w, ctx.scale))
        nat.check(L.fpc_upsample_bilinear_bwd(g.data_ptr(), sb, sc, sh, sw, gx.data_ptr(), ws.data_ptr(), B, C, h, w, ctx.scale,
                                              int(ctx.in_nhwc), nat.stream()), "fpc_upsample_bilinear_bwd")
        return gx, None, None


def upsample_bilinear(x, scale, out_nchw=False):
    """F.interpolate(x, scale_factor=scale, mode='bilinear', align_corners=True) on the native kernels (forward and backward)
    for f32 GPU tensors with scale 2 or 4; torch otherwise.  The result keeps x's memory format unless out_nchw (the heads:
    full-resolution logits in the layout the losses and the post-network kernels read)."""
    if ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and scale in (2, 4) and x.shape[0] * x.shape[1] <= 65535:
        nhwc = (not out_nchw) and x.shape[1] > 1 and x.stride(1) == 1 and x.is_contiguous(memory_format=torch.channels_last)
        if any(s < 0 for s in x.stride()) or x.stride(1) == 0:
            x = x.contiguous()
        return _UpBilinearFn.apply(x, int(scale), bool(nhwc))
    return torch.nn.functional.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=True)


# ---- GroupNorm(32 groups of 4 channels) + ReLU on channel-last activations (the decoder's Conv3x3GNReLU blocks) -----------

class _GroupNormReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps):
        B, C, H, W = x.shape
        L = nat.lib()
        y = torch.empty_like(x, memory_format=torch.channels_last)
        stats = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
        part = torch.empty(L.fpc_groupnorm4_relu_scratch_floats(B, H * W, C), dtype=torch.float32, device=x.device)
        gamma, beta = gamma.contiguous(), beta.contiguous()
        nat.check(L.fpc_groupnorm4_relu_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(), part.data_ptr(),
                                            B, H * W, C, groups, float(eps), nat.stream()), "fpc_groupnorm4_relu_fwd")
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.groups = groups
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, beta, stats = ctx.saved_tensors
        B, C, H, W = x.shape
        L = nat.lib()
        gy = _channels_last(gy)
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        chunks = L.fpc_groupnorm4_relu_scratch_floats(B, H * W, C) // (B * C * 2)
        part = torch.empty((B, chunks, C, 2), dtype=torch.float32, device=x.device)
        nat.check(L.fpc_groupnorm4_relu_bwd(x.data_ptr(), gy.data_ptr(), gamma.data_ptr(), beta.data_ptr(), stats.data_ptr(), dx.data_ptr(),
                                            part.data_ptr(), B, H * W, C, ctx.groups, nat.stream()), "fpc_groupnorm4_relu_bwd")
        sums = part.sum((0, 1))                      # [C, 2]: dbeta, dgamma
        return dx, sums[:, 1].contiguous(), sums[:, 0].contiguous(), None, None


def groupnorm_relu(x, gn):
    """relu(gn(x)) for a torch.nn.GroupNorm `gn`: native on channel-last f32 GPU tensors whose groups are 4 channels wide
    (the decoder: GroupNorm(32, 128)), torch otherwise."""
    C, G = gn.num_channels, gn.num_groups
    if (ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and gn.affine and C == 4 * G and 256 % G == 0 and G <= 64
            and x.shape[0] <= 65535 and x.stride(1) == 1 and x.is_contiguous(memory_format=torch.channels_last)):
        return _GroupNormReLUFn.apply(x, gn.weight, gn.bias, G, gn.eps)
    return torch.relu(gn(x))
