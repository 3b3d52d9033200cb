"""GPU parity of the backbone engine (fpc_net_*, fpc_conv2d) — floating point, so the bar is the
north-star tolerance (1e-4 relative to the tensor's scale, fp32) against
  * torch in float64 on the CPU (small shapes: the tight check), and
  * the torch-ROCm modules (the path the engine replaces) at full 640x480.
The conv stack's oracle is torch itself: segmentation_models_pytorch is not vendored in the
reference, so every op is checked against torch.nn.functional.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def lib(dev):
    import fastposecnn_amd.lib as L
    from fastposecnn_amd import _native
    _native.lib()
    return L


def _conv2d(dev, x_nchw, w, stride, pad, scale=None, shift=None, res=None, up=None, relu=False, gn=False, bm=0, bn=0,
            nsplit=0, nchw_input=False):
    """Runs fpc_conv2d; x is given NCHW (CPU), fed as NHWC unless nchw_input. Returns (out NCHW cpu, gn_part)."""
    from fastposecnn_amd import _native as nat
    L = nat.lib()
    B, Cin, Hi, Wi = x_nchw.shape
    Cout, _, Kh, Kw = w.shape
    Ho = (Hi + 2 * pad - Kh) // stride + 1
    Wo = (Wi + 2 * pad - Kw) // stride + 1
    if nchw_input:
        xin = x_nchw.contiguous().to(dev)
        sb, sc, sh, sw = xin.stride()
    else:
        xin = x_nchw.permute(0, 2, 3, 1).contiguous().to(dev)
        sb, sh, sw, sc = xin.stride()
    wd = w.contiguous().to(dev)
    out = torch.full((B, Ho, Wo, Cout), float("nan"), device=dev)
    plan = (ctypes.c_int * 4)()
    nat.check(L.fpc_conv2d_plan(B, Ho, Wo, Cin, Cout, Kh, Kw, bm, bn, nsplit, plan), "plan")
    P32 = plan[3]
    gpart = torch.zeros((B, P32, Cout, 2), device=dev) if gn else None
    ws = torch.empty(L.fpc_conv2d_workspace_bytes(B, Ho, Wo, Cin, Cout, Kh, Kw), dtype=torch.uint8, device=dev)
    t = lambda a: None if a is None else a.contiguous().to(dev)
    scale_d, shift_d = t(scale), t(shift)
    res_d = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(dev)
    up_d = None if up is None else up.permute(0, 2, 3, 1).contiguous().to(dev)
    nat.check(L.fpc_conv2d(xin.data_ptr(), sb, sh, sw, sc, wd.data_ptr(), nat.ptr(scale_d), nat.ptr(shift_d),
                           nat.ptr(res_d), nat.ptr(up_d), out.data_ptr(), nat.ptr(gpart), B, Hi, Wi, Cin, Cout, Kh, Kw,
                           stride, pad, int(relu), bm, bn, nsplit, ws.data_ptr(), ws.numel(), nat.stream()), "conv2d")
    torch.cuda.synchronize()
    return out.permute(0, 3, 1, 2).cpu(), (None if gpart is None else gpart.cpu()), tuple(plan)


def _ref_conv(x, w, stride, pad, scale=None, shift=None, res=None, up=None, relu=False):
    y = F.conv2d(x.double(), w.double(), stride=stride, padding=pad)
    if scale is not None:
        y = y * scale.double().view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.double().view(1, -1, 1, 1)
    if res is not None:
        y = y + res.double()
    if up is not None:
        y = y + F.interpolate(up.double(), scale_factor=2, mode="nearest")
    if relu:
        y = y.relu()
    return y


CONV_CASES = [
    # B, Cin, Hi, Wi, Cout, k, stride, pad, (bm, bn, nsplit), extras
    (1, 64, 30, 40, 64, 3, 1, 1, (0, 0, 0), "bn_relu"),
    (2, 64, 15, 20, 128, 3, 2, 1, (64, 64, 1), "bn_relu_res"),
    (1, 128, 16, 24, 128, 3, 1, 1, (128, 128, 1), "gn"),
    (1, 128, 16, 24, 128, 3, 1, 1, (128, 64, 1), "gn"),
    (1, 128, 16, 24, 128, 3, 1, 1, (64, 128, 1), "gn"),
    (2, 256, 15, 20, 128, 3, 1, 1, (64, 64, 4), "gn"),          # split-K + GroupNorm partials, ragged M (300)
    (1, 512, 15, 20, 512, 3, 1, 1, (0, 0, 0), "bn_relu_res"),   # planner picks split-K
    # split-K summed by the second launch (nsplit = 100 + split; the default is the fused last-arriver form)
    (2, 256, 15, 20, 128, 3, 1, 1, (64, 64, 104), "gn"),
    (1, 512, 15, 20, 512, 3, 1, 1, (128, 64, 108), "bn_relu_res"),
    (2, 256, 30, 40, 128, 3, 1, 1, (128, 128, 6), "gn"),         # fused, 128 x 128 tiles
    (2, 64, 24, 32, 256, 1, 1, 0, (64, 128, 2), "bias_up"),      # fused, FPN lateral epilogue
    (2, 64, 24, 32, 256, 1, 1, 0, (0, 0, 0), "bias_up"),        # FPN lateral: 1x1 + bias + nearest-x2 add
    (1, 64, 24, 32, 128, 1, 2, 0, (0, 0, 0), "bn"),             # downsample 1x1 stride 2
    (2, 3, 64, 96, 64, 7, 2, 3, (0, 0, 0), "stem"),             # 7x7/2 on the NCHW image (generic loader)
    (1, 128, 12, 16, 7, 1, 1, 0, (64, 64, 1), "bias"),          # Cout not a multiple of anything
    (1, 32, 9, 11, 24, 3, 1, 1, (64, 64, 1), "bias_relu"),
    # Winograd F(2x2,3x3) kernel (nsplit = -1 requests it)
    (1, 64, 30, 40, 64, 3, 1, 1, (0, 0, -1), "bn_relu_res"),
    (2, 256, 15, 20, 128, 3, 1, 1, (0, 0, -1), "gn"),            # odd height: half-empty last tile row
    (1, 128, 34, 50, 128, 3, 1, 1, (0, 0, -1), "gn"),            # ragged tile patches in both directions
    (1, 8, 7, 9, 64, 3, 1, 1, (0, 0, -1), "bias_relu"),
    # wave-private, barrier-free K loop (nsplit = -3)
    (1, 64, 30, 40, 64, 3, 1, 1, (0, 0, -3), "bn_relu_res"),
    (2, 256, 15, 20, 128, 3, 1, 1, (0, 0, -3), "gn"),
    (1, 128, 34, 50, 128, 3, 1, 1, (0, 0, -3), "gn"),
    # 8-wave all-DMA three-stage form (nsplit = -4)
    (1, 64, 30, 40, 64, 3, 1, 1, (0, 0, -4), "bn_relu_res"),
    (2, 256, 15, 20, 128, 3, 1, 1, (0, 0, -4), "gn"),
    (1, 128, 34, 50, 128, 3, 1, 1, (0, 0, -4), "gn"),
    (1, 8, 7, 9, 64, 3, 1, 1, (0, 0, -4), "bias_relu"),       # a single K-step
    (1, 16, 20, 24, 64, 3, 1, 1, (0, 0, -4), "bias_relu"),    # two K-steps
    # split-precision matrix products (nsplit = 1000 + split): bf16 x 3 planes, six MFMAs per tile, f32 accumulation
    (1, 64, 30, 40, 64, 3, 1, 1, (64, 64, 1001), "bn_relu_res"),
    (2, 256, 15, 20, 128, 3, 1, 1, (64, 64, 1004), "gn"),         # split-K + GroupNorm partials, ragged M
    (1, 128, 16, 24, 128, 3, 1, 1, (128, 128, 1001), "gn"),
    (1, 128, 16, 24, 128, 3, 1, 1, (64, 128, 1001), "gn"),
    (2, 64, 24, 32, 256, 1, 1, 0, (64, 64, 1001), "bias_up"),     # FPN lateral
    (1, 64, 24, 32, 128, 1, 2, 0, (128, 64, 1001), "bn"),         # 1x1 stride 2
    (1, 96, 20, 24, 64, 3, 1, 1, (64, 64, 1002), "bias_relu"),    # 27 K-steps in slices of 14 + 13: both phase parities end a slice (DMA-staged weight planes)
    (1, 32, 12, 16, 64, 3, 1, 1, (64, 64, 1001), "bias_relu"),    # 9 K-steps, one slice
    (1, 32, 12, 16, 64, 1, 1, 0, (64, 64, 1001), "bias"),         # a single K-step: prologue only
    # split-precision Winograd (nsplit = -5): transformed tiles split into three bf16 pieces, three bf16 MFMAs per K-step, 8 waves
    (1, 64, 30, 40, 64, 3, 1, 1, (0, 0, -5), "bn_relu_res"),
    (2, 256, 15, 20, 128, 3, 1, 1, (0, 0, -5), "gn"),
    (1, 128, 34, 50, 128, 3, 1, 1, (0, 0, -5), "gn"),
    (1, 8, 7, 9, 64, 3, 1, 1, (0, 0, -5), "bias_relu"),       # a single K-step
    (1, 16, 20, 24, 64, 3, 1, 1, (0, 0, -5), "bias_relu"),    # two K-steps
    # pixel-resident FPN lateral product, k_lateral1x1 (nsplit = 2000 + workgroups per 128-pixel tile): K = 64 / 128, bf16 x 3 planes
    (2, 64, 24, 32, 256, 1, 1, 0, (0, 0, 2001), "bias_up"),       # 768 pixels = 6 tiles, one workgroup walks all 8 weight tiles
    (2, 64, 24, 32, 256, 1, 1, 0, (0, 0, 2004), "bias_up"),
    (1, 128, 30, 40, 256, 1, 1, 0, (0, 0, 2002), "bias_up"),      # K = 128; 1200 pixels: ragged last tile (48 rows)
    (1, 128, 30, 40, 256, 1, 1, 0, (0, 0, 2008), "bias"),         # one weight tile per workgroup, no top-down addend
    (3, 64, 10, 14, 96, 1, 1, 0, (0, 0, 2003), "bias_relu"),      # 140 pixels: waves 1-3 of the second tile partly / fully empty; Cout = 96
    (1, 64, 6, 8, 32, 1, 1, 0, (0, 0, 2001), "none"),             # no bias: accumulation starts from zero
    # weight-resident 7x7 / s2 stem, k_stem7x7 (nsplit = 3000): NHWC4 image (4th channel 0), output rows of 64-pixel segments
    (2, 4, 96, 128, 64, 7, 2, 3, (0, 0, 3000), "stem4"),          # 48 x 64 outputs: every border case of a row / column
    (3, 4, 40, 256, 64, 7, 2, 3, (0, 0, 3000), "stem4"),          # two segments per row, more tiles than waves of one workgroup
    # 8-wave form (8x8 tile patch per workgroup, nsplit = -2)
    (1, 64, 30, 40, 64, 3, 1, 1, (0, 0, -2), "bn_relu_res"),
    (2, 256, 15, 20, 128, 3, 1, 1, (0, 0, -2), "gn"),
    (1, 128, 34, 50, 128, 3, 1, 1, (0, 0, -2), "gn"),
    # split-precision Winograd with 128 output channels per workgroup (nsplit = -6, wino128.hip): 8 x 4 tile patches, 4 waves, weights
    # straight into the operand registers.  Ragged patches on both axes, every image border, one / two / many K-steps, two
    # 128-channel blocks, residual + folded BatchNorm, GroupNorm partial sums
    (1, 64, 30, 40, 128, 3, 1, 1, (0, 0, -6), "bn_relu_res"),
    (2, 256, 15, 20, 128, 3, 1, 1, (0, 0, -6), "gn"),
    (1, 128, 34, 50, 256, 3, 1, 1, (0, 0, -6), "gn"),
    (1, 8, 7, 9, 128, 3, 1, 1, (0, 0, -6), "bias_relu"),        # a single K-step
    (1, 16, 20, 24, 128, 3, 1, 1, (0, 0, -6), "bias_relu"),     # two K-steps
    (3, 24, 16, 32, 128, 3, 1, 1, (0, 0, -6), "gn"),            # three K-steps, patches that tile the image exactly
    # split-precision Winograd as four waves of 512 registers (nsplit = -7, wino_w4.hip): 8 x 8 tile patches x 64 channels, weights
    # straight into the operand registers, next step's pieces split in place
    (1, 64, 30, 40, 64, 3, 1, 1, (0, 0, -7), "bn_relu_res"),
    (2, 256, 15, 20, 128, 3, 1, 1, (0, 0, -7), "gn"),
    (1, 128, 34, 50, 128, 3, 1, 1, (0, 0, -7), "gn"),
    (1, 8, 7, 9, 64, 3, 1, 1, (0, 0, -7), "bias_relu"),         # a single K-step
    (1, 16, 20, 24, 64, 3, 1, 1, (0, 0, -7), "bias_relu"),      # two K-steps
    (3, 24, 32, 32, 192, 3, 1, 1, (0, 0, -7), "gn"),            # three K-steps, patches that tile the image exactly, three channel blocks
    # the four-wave form on two fp16 pieces per operand (nsplit = -8, wino_h2.hip): weights scaled by a power of two on the device,
    # activations as they come (N(0, 1) here) — held to the same bar as every other form
    (1, 64, 30, 40, 64, 3, 1, 1, (0, 0, -8), "bn_relu_res"),
    (2, 256, 15, 20, 128, 3, 1, 1, (0, 0, -8), "gn"),
    (1, 128, 34, 50, 128, 3, 1, 1, (0, 0, -8), "gn"),
    (1, 8, 7, 9, 64, 3, 1, 1, (0, 0, -8), "bias_relu"),         # a single K-step
    (1, 16, 20, 24, 64, 3, 1, 1, (0, 0, -8), "bias_relu"),      # two K-steps
    (3, 24, 32, 32, 192, 3, 1, 1, (0, 0, -8), "gn"),            # three K-steps, three channel blocks
    # ... with three of its four piece products, over PAIRS of K-steps (nsplit = -9, wino_h3.hip; Cin a multiple of 16): the same bar
    (1, 64, 30, 40, 64, 3, 1, 1, (0, 0, -9), "bn_relu_res"),
    (2, 256, 15, 20, 128, 3, 1, 1, (0, 0, -9), "gn"),
    (1, 128, 34, 50, 128, 3, 1, 1, (0, 0, -9), "gn"),
    (1, 16, 7, 9, 64, 3, 1, 1, (0, 0, -9), "bias_relu"),        # a single pair
    (1, 32, 20, 24, 64, 3, 1, 1, (0, 0, -9), "bias_relu"),      # two pairs
    (3, 48, 32, 32, 192, 3, 1, 1, (0, 0, -9), "gn"),            # three pairs (the staging ring wraps), three channel blocks
    (2, 80, 33, 47, 64, 3, 1, 1, (0, 0, -9), "bn_relu_res"),    # five pairs, ragged border patches
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: f"{c[1]}x{c[2]}x{c[3]}-{c[4]}-k{c[5]}s{c[6]}-{c[8]}-{c[9]}")
def test_conv2d_vs_float64(lib, dev, case):
    B, Cin, Hi, Wi, Cout, k, stride, pad, (bm, bn, ns), extra = case
    g = torch.Generator().manual_seed(CONV_CASES.index(case))
    x = torch.randn((B, Cin, Hi, Wi), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) / (Cin * k * k) ** 0.5
    Ho, Wo = (Hi + 2 * pad - k) // stride + 1, (Wi + 2 * pad - k) // stride + 1
    kw = {}
    if "bn" in extra:
        kw["scale"] = torch.rand(Cout, generator=g) + 0.5
        kw["shift"] = torch.randn(Cout, generator=g)
    if "bias" in extra:
        kw["shift"] = torch.randn(Cout, generator=g)
    if "res" in extra:
        kw["res"] = torch.randn((B, Cout, Ho, Wo), generator=g)
    if "up" in extra:
        kw["up"] = torch.randn((B, Cout, Ho // 2, Wo // 2), generator=g)
    if "relu" in extra or extra == "stem":
        kw["relu"] = True
    if extra in ("stem", "stem4"):
        kw["scale"] = torch.rand(Cout, generator=g) + 0.5
        kw["shift"] = torch.randn(Cout, generator=g)
    if extra == "stem4":
        x[:, 3] = 0.0                                              # the engine's NHWC4 image: the fourth channel is zero
        kw["relu"] = True
    out, gpart, plan = _conv2d(dev, x, w, stride, pad, gn=("gn" in extra), bm=bm, bn=bn, nsplit=ns,
                               nchw_input=(extra == "stem"), **kw)
    ref = _ref_conv(x, w, stride, pad, **kw)
    assert not torch.isnan(out).any(), "unwritten outputs"
    err = (out.double() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), (err, plan)
    if gpart is not None:
        s = gpart.double().sum(1)                                   # [B, Cout, 2] over the row tiles
        np.testing.assert_allclose(s[..., 0].numpy(), ref.sum((2, 3)).numpy(), rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(s[..., 1].numpy(), (ref * ref).sum((2, 3)).numpy(), rtol=1e-4, atol=1e-3)


def test_fused_split_k_equals_two_launch_form_bitwise(lib, dev):
    """The last-arriver fix-up sums the partials in split order, like k_conv_splitk_epilogue: the outputs of the two forms
    agree bit for bit, run after run (the arrival counters are left at zero by every launch).  The GroupNorm partial sums
    are reduced over a tile's rows in another order by the two kernels: equal to rounding."""
    g = torch.Generator().manual_seed(77)
    x = torch.randn((2, 256, 15, 20), generator=g)
    w = torch.randn((128, 256, 3, 3), generator=g) / 48.0
    res = torch.randn((2, 128, 15, 20), generator=g)
    for bm, bn, ns in ((64, 64, 12), (128, 64, 3), (64, 128, 24)):
        two, gp2, _ = _conv2d(dev, x, w, 1, 1, res=res, relu=True, gn=True, bm=bm, bn=bn, nsplit=100 + ns)
        for _ in range(3):
            one, gp1, plan = _conv2d(dev, x, w, 1, 1, res=res, relu=True, gn=True, bm=bm, bn=bn, nsplit=ns)
            assert plan[2] == ns and torch.equal(one, two), plan
            torch.testing.assert_close(gp1, gp2, rtol=1e-5, atol=1e-4)


def _model(lib, dev, encoder, seed=0):
    from fastposecnn_amd import config
    hp = config.INFERENCE()
    hp.RUNTIME_TIMING = False
    hp.ENCODER = encoder
    hp.PERFORM_AGGREGATION = False
    torch.manual_seed(seed)
    m = lib.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(None, hp)
    # non-trivial BatchNorm statistics / affine parameters so that folding is exercised
    g = torch.Generator().manual_seed(seed + 1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
            mod.weight.data.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
            mod.bias.data.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
        if isinstance(mod, torch.nn.GroupNorm):
            mod.weight.data.copy_(torch.rand(mod.num_channels, generator=g) + 0.5)
            mod.bias.data.copy_(torch.randn(mod.num_channels, generator=g) * 0.1)
    return m.eval(), hp


def _torch_path(m, x):
    """The torch-module path of the same model (what the engine replaces)."""
    m.HPARAM.USE_NATIVE_ENGINE = False
    try:
        with torch.no_grad():
            return m.pure_model_forward(x)
    finally:
        m.HPARAM.USE_NATIVE_ENGINE = True


@pytest.mark.parametrize("encoder,B,H,W", [("resnet18", 2, 64, 96), ("resnet34", 1, 96, 64)])
def test_net_vs_float64_cpu(lib, dev, encoder, B, H, W):
    from fastposecnn_amd import synth
    m, hp = _model(lib, dev, encoder)
    x = torch.stack([synth.make_image(i, H, W) for i in range(B)])
    import copy
    ref_m = copy.deepcopy(m).double()
    ref_m.HPARAM = copy.copy(hp); ref_m.HPARAM.USE_NATIVE_ENGINE = False
    with torch.no_grad():
        ref = ref_m.pure_model_forward(x.double())
    m = m.to(dev)
    with torch.no_grad():
        out = m(x.to(dev))
    assert m._engines, "the native engine did not run"
    eng = next(iter(m._engines.values()))
    # encoder features localise a failure
    with torch.no_grad():
        feats = ref_m.encoder(x.double())
    for name, f in zip(("c2", "c3", "c4", "c5"), feats[2:]):
        got = eng.tensor(name).permute(0, 3, 1, 2).cpu().double()
        err = (got - f).abs().max().item()
        assert err <= 1e-4 * max(1.0, f.abs().max().item()), (name, err)
    for k in ("mask", "quaternion", "scales", "xy", "z"):
        got = out["logits"][k].cpu().double()
        assert got.shape == ref[k].shape, k
        err = (got - ref[k]).abs().max().item()
        assert err <= 1e-4 * max(1.0, ref[k].abs().max().item()), (k, err)


@pytest.mark.parametrize("encoder,B", [("resnet18", 1), ("resnet34", 32)], ids=["config2-r18-b1", "config3-r34-b32"])
def test_net_fullsize_vs_torch_modules(lib, dev, encoder, B):
    """640x480 at the two single-GPU configurations of BASELINE.json (configs[1]: ResNet18 B = 1; configs[2]: ResNet34
    B = 32, per-image seeds 0..31 — the plan's tilings and split-K choices depend on B): engine logits vs the torch-ROCm
    module path; the fused class compression is bit-identical to the stand-alone kernel on the engine's own logits."""
    import gpu_tensor_funcs as gtf
    from fastposecnn_amd import synth
    m, hp = _model(lib, dev, encoder)
    m = m.to(dev)
    x = torch.stack([synth.make_image(i) for i in range(B)]).to(dev)
    with torch.no_grad():
        out = m(x)
    assert m._engines
    ref = _torch_path(m, x)
    for k in ("mask", "quaternion", "scales", "xy", "z"):
        scale = max(1.0, ref[k].abs().max().item())
        err = (out["logits"][k] - ref[k]).abs().max().item()
        assert err <= 2e-4 * scale, (k, err, scale)       # MIOpen's Winograd convs carry their own f32 error
    cat = gtf.class_compression_fused(7, out["logits"])
    assert torch.equal(cat["mask"], out["categorical"]["mask"])
    for k in ("quaternion", "scales", "xy", "z"):
        assert torch.equal(cat[k], out["categorical"][k]), k
    # schema of the reference's forward
    assert set(out) == {"logits", "categorical", "aggregated"} and out["aggregated"] is None
    assert out["categorical"]["mask"].dtype == torch.int64 and tuple(out["categorical"]["z"].shape) == (B, 480, 640)


def test_net_fullsize_config2_vs_float64(lib, dev):
    """BASELINE.json configs[1] at the size the bench runs (ResNet18, B = 1, 640 x 480): the engine's logits against the
    float64 CPU module path at north_star's 1e-4 of each tensor's scale — with the split-precision products the autotuner
    may pick (the default) AND with plain f32 products only.  (The MIOpen comparison above is 2e-4 against a backend with
    its own Winograd error; this is the reference arithmetic itself, ~15 s of float64 convolutions on the host.)"""
    import copy
    from fastposecnn_amd import synth
    x = synth.make_image(0)[None]
    m, hp = _model(lib, dev, "resnet18")
    ref_m = copy.deepcopy(m).double()
    ref_m.HPARAM = copy.copy(hp); ref_m.HPARAM.USE_NATIVE_ENGINE = False
    with torch.no_grad():
        ref = ref_m.pure_model_forward(x.double())
    assert tuple(ref["mask"].shape) == (1, 7, 480, 640)
    for split in (True, False):
        m2, hp2 = _model(lib, dev, "resnet18")
        hp2.ENGINE_SPLIT_PRECISION = split
        m2 = m2.to(dev)
        with torch.no_grad():
            out = m2(x.to(dev))
        assert m2._engines, "the native engine did not run"
        plans = next(iter(m2._engines.values())).conv_plans()
        assert split or not any(p[2] == -5 for p in plans)
        for k in ("mask", "quaternion", "scales", "xy", "z"):
            got = out["logits"][k].cpu().double()
            err = (got - ref[k]).abs().max().item()
            assert err <= 1e-4 * max(1.0, ref[k].abs().max().item()), (split, k, err, ref[k].abs().max().item())
        del m2, out
        torch.cuda.empty_cache()


def test_net_fullsize_config3_vs_float64(lib, dev):
    """BASELINE.json configs[2] at the size and batch the bench's top-level record runs (ResNet34, B = 32, 640 x 480, per-image
    seeds 0..31): the logits of frames 0 and 31 OF THE 32-FRAME BATCH — i.e. produced by the batch-32 plan set, whose autotuner
    picks kernels the batch-1 plans do not (k_stem7x7, k_lateral1x1 `parts`, k_head_part tiles, the four-wave Winograd form) —
    against the float64 CPU module path of those two frames at north_star's 1e-4 of each tensor's scale (the
    headline's own kernels were held to 2e-4 against MIOpen only).  Then the same two frames as a batch of 2 ON the batch-32
    plans (fpc_net_copy_plans): the same bar.  ~90 s of float64 convolutions on the host."""
    import copy
    from fastposecnn_amd import synth
    m, hp = _model(lib, dev, "resnet34")
    x2 = torch.stack([synth.make_image(0), synth.make_image(31)])
    ref_m = copy.deepcopy(m).double()
    ref_m.HPARAM = copy.copy(hp); ref_m.HPARAM.USE_NATIVE_ENGINE = False
    with torch.no_grad():
        ref = ref_m.pure_model_forward(x2.double())
    del ref_m
    assert tuple(ref["mask"].shape) == (2, 7, 480, 640)
    m = m.to(dev)
    x32 = torch.stack([synth.make_image(i) for i in range(32)]).to(dev)
    with torch.no_grad():
        out = m(x32)
    e32 = m._engines[(32, 480, 640, x32.device)]
    plans32 = e32.conv_plans()
    for k in ("mask", "quaternion", "scales", "xy", "z"):
        got = out["logits"][k][[0, 31]].cpu().double()
        scale = max(1.0, ref[k].abs().max().item())
        err = (got - ref[k]).abs().max().item()
        assert err <= 1e-4 * scale, ("batch-32 forward", k, err, scale)
    del out
    torch.cuda.empty_cache()
    with torch.no_grad():
        m(x2.to(dev))                                   # builds the batch-2 engine (its own plans)
        e2 = m._engines[(2, 480, 640, x32.device)]
        e2.copy_plans_from(e32)
        assert e2.conv_plans() == plans32, "a batch-32 plan did not fit the batch-2 engine"
        out2 = m(x2.to(dev))
    for k in ("mask", "quaternion", "scales", "xy", "z"):
        got = out2["logits"][k].cpu().double()
        scale = max(1.0, ref[k].abs().max().item())
        err = (got - ref[k]).abs().max().item()
        assert err <= 1e-4 * scale, ("batch 2 on the batch-32 plans", k, err, scale)
    del m._engines[(32, 480, 640, x32.device)]
    torch.cuda.empty_cache()


def test_engine_invalidation(lib, dev):
    """Packed weights are a snapshot keyed on (address, version) of every bound tensor: in-place updates, a
    load_state_dict on the model OR on a sub-module, and FrameStreamer's plan copies all pick the new values up
    without losing the tuned plan; train()/eval() alone do not repack; .to() drops the plans."""
    import copy
    from fastposecnn_amd import synth
    from fastposecnn_amd.streaming import FrameStreamer
    m, hp = _model(lib, dev, "resnet18")
    m = m.to(dev)
    x = synth.make_image(1, 64, 64)[None].to(dev)
    with torch.no_grad():
        a = m(x)["logits"]["mask"].clone()
        assert m._engines
        eng = next(iter(m._engines.values()))
        plans = eng.conv_plans()
        m.train(); m.eval()                                   # no parameter changed: same plan, not repacked
        assert next(iter(m._engines.values())) is eng and not eng.stale() and eng.reloads == 1
        m.segmentation_head[0].bias.add_(1.0)                 # read in place by the plan; still flagged
        assert eng.stale()
        b = m(x)["logits"]["mask"].clone()
        assert eng.reloads == 2 and eng.conv_plans() == plans # repacked, tilings kept
        assert torch.allclose(b, a + 1.0, atol=1e-5)
        # a SNAPSHOTTED tensor (conv weight, folded BN) changed through a sub-module's load_state_dict
        sd = copy.deepcopy(m.encoder.state_dict())
        sd["conv1.weight"] = sd["conv1.weight"] * 0.5
        sd["bn1.running_var"] = sd["bn1.running_var"] * 4.0
        m.encoder.load_state_dict(sd)
        c = m(x)["logits"]["mask"].clone()
        assert eng.reloads == 3
        ref = _torch_path(m, x)["mask"]
        assert (c - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
        assert (c - b).abs().max().item() > 1e-3              # the change is visible at all
        # FrameStreamer copies own their plans and re-check for themselves
        st = FrameStreamer(m, net_streams=2)
        for _ in range(2):
            st.collect(st.submit(x))
        m.segmentation_head[0].bias.sub_(1.0)
        outs = [st.collect(st.submit(x))["logits"]["mask"] for _ in range(2)]
        for o in outs:
            assert torch.allclose(o, c - 1.0, atol=1e-5)
    m.train()
    out = m(x)                                                # training mode: torch modules, autograd works
    assert out["logits"]["mask"].requires_grad
    m2 = m.eval().to("cpu").to(dev)                           # .to(): parameters are new tensors, plans dropped
    assert not m2._engines


def test_engine_follows_training_steps(lib, dev):
    """eval (plan built) -> training steps with the sharded native optimiser -> eval: the optimiser kernel writes the
    parameters through raw pointers and a training-mode forward moves BatchNorm's running statistics in place, neither
    of which PyTorch versions; the plan (and a FrameStreamer copy's) must still serve the NEW weights."""
    from fastposecnn_amd import synth
    from fastposecnn_amd.streaming import FrameStreamer
    from fastposecnn_amd.train_parallel import ShardedLookaheadRAdam
    m, hp = _model(lib, dev, "resnet18")
    m = m.to(dev)
    x = synth.make_image(1, 64, 64)[None].to(dev)
    with torch.no_grad():
        before = m(x)["logits"]["mask"].clone()
    eng = next(iter(m._engines.values()))
    st = FrameStreamer(m, net_streams=2)
    for _ in range(2):
        st.collect(st.submit(x))
    m.train()
    opt = ShardedLookaheadRAdam(m, lr=1e-2, weight_decay=0.0)
    for step in range(3):
        opt.zero_grad()
        out = m.pure_model_forward(x)
        sum(v.square().mean() for v in out.values()).backward()
        opt.step()
    m.eval()
    with torch.no_grad():
        after = m(x)["logits"]["mask"].clone()
        ref = _torch_path(m, x)["mask"]
        outs = [st.collect(st.submit(x))["logits"]["mask"] for _ in range(2)]
    assert next(iter(m._engines.values())) is eng and eng.reloads >= 2            # repacked, not rebuilt
    assert (after - before).abs().max().item() > 1e-3                             # the steps changed the output at all
    tol = 2e-4 * max(1.0, ref.abs().max().item())
    assert (after - ref).abs().max().item() <= tol
    for o in outs:
        assert (o - ref).abs().max().item() <= tol


@pytest.mark.parametrize("kw", [dict(), dict(net_streams=2, post_inline=False), dict(net_streams=2, tune_mode=1)],
                         ids=["4-streams-inline", "2-streams+post-stream", "2-streams-throughput-tuned"])
def test_frame_streamer_matches_forward(lib, dev, kw):
    """Three frames in flight on the streaming runtime give, frame by frame, what forward() gives.  tune_mode=1: every plan
    of the runtime is a copy tuned for several frames in flight; the caller's model keeps its own plan."""
    from fastposecnn_amd import config, synth
    from fastposecnn_amd.streaming import FrameStreamer
    hp = config.INFERENCE()
    hp.RUNTIME_TIMING = False
    hp.HV_NUM_OF_HYPOTHESES = 128
    torch.manual_seed(0)
    m = lib.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(None, hp).to(dev).eval()
    H, W = 96, 128
    xs = [synth.make_image(i, H, W)[None].to(dev) for i in range(5)]
    cats = []
    for i in range(5):
        c, _ = synth.make_vote_frame(i, K=3, H=H, W=W, rmin=8, rmax=20)
        cats.append({k: v.to(dev) for k, v in c.items()})
    ref = []
    with torch.no_grad():
        for i in range(5):
            logits = m.pure_model_forward(xs[i])
            cat = m.class_compression(logits)
            torch.manual_seed(100 + i)
            ref.append((logits, cat, m.agg_hough_and_generate_RT(cats[i])))
    st = FrameStreamer(m, **kw)
    torch.manual_seed(99)
    st.prepare(xs[0], categorical_override=cats[0])        # builds every stream's plan up front (bench.py's set-up)
    assert len(st._warm) == len(st.models)
    if kw.get("tune_mode") is not None:
        assert all(mm is not m and mm.HPARAM.ENGINE_TUNE_MODE == kw["tune_mode"] for mm in st.models)
        assert getattr(m.HPARAM, "ENGINE_TUNE_MODE", 0) == 0
    tickets = []
    for i in range(5):
        torch.manual_seed(100 + i)                     # the vote's sampler seed is drawn at submit time
        tickets.append(st.submit(xs[i], categorical_override=cats[i]))
    for i in range(5):
        out = st.collect(tickets[i])
        # the two plans are autotuned separately (different split-K / tilings -> different f32 summation
        # order), so the network outputs agree to rounding, not bit for bit
        for k in ("mask", "quaternion", "scales", "xy", "z"):
            a, b = out["logits"][k], ref[i][0][k]
            assert (a - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item()), (i, k)
        assert (out["categorical"]["mask"] != ref[i][1]["mask"]).float().mean().item() < 1e-3
        assert set(out["aggregated"]) == set(ref[i][2])
        for k, v in ref[i][2].items():
            assert torch.equal(out["aggregated"][k], v), (i, k)
        assert out["aggregated"]["class_ids"].shape[0] == 3


def test_frame_streamer_keeps_the_fastest_of_several_plan_sets(lib, dev):
    """prepare(tune_trials=N): the runtime's plans are built N times, each set streamed, the fastest kept — the kept plans
    are live (graphs recorded on their streams) and give the model's own forward to rounding."""
    from fastposecnn_amd import config, synth
    from fastposecnn_amd.streaming import FrameStreamer
    hp = config.INFERENCE()
    hp.RUNTIME_TIMING = False
    hp.HV_NUM_OF_HYPOTHESES = 128
    torch.manual_seed(0)
    m = lib.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(None, hp).to(dev).eval()
    H, W = 96, 128
    x = synth.make_image(3, H, W)[None].to(dev)
    c, _ = synth.make_vote_frame(3, K=3, H=H, W=W, rmin=8, rmax=20)
    cat = {k: v.to(dev) for k, v in c.items()}
    with torch.no_grad():
        ref = m.pure_model_forward(x)
    st = FrameStreamer(m, net_streams=2, tune_mode=1)
    st.prepare(x, categorical_override=cat, tune_trials=3, trial_frames=8)
    assert len(st.trial_rates) == 3 and all(r > 0 for r in st.trial_rates)
    assert all(len(mm._engines) == 1 for mm in st.models)
    kept = [next(iter(mm._engines.values())) for mm in st.models]
    outs = [st.collect(st.submit(x, categorical_override=cat)) for _ in range(4)]
    assert [next(iter(mm._engines.values())) for mm in st.models] == kept          # no plan was rebuilt by the frames after prepare
    for out in outs:
        for k in ("mask", "quaternion", "scales", "xy", "z"):
            a, b = out["logits"][k], ref[k]
            assert (a - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item()), k
        assert out["aggregated"]["class_ids"].shape[0] == 3


def test_frame_streamer_async_consumers_and_dropped_inputs(lib, dev):
    """The ownership rule of FrameStreamer.collect under allocator pressure: every input tensor is created right before
    submit() and dropped right after it, every output is consumed by kernels queued asynchronously on the caller's stream
    and dropped at once, and other allocations churn the caller's pool in between.  The accumulated device-side checksums
    must equal those of a run that synchronises after every frame (a block handed out again while a queued reader or
    the frame's first kernel still uses it would change them)."""
    from fastposecnn_amd import config, synth
    from fastposecnn_amd.streaming import FrameStreamer
    hp = config.INFERENCE()
    hp.RUNTIME_TIMING = False
    hp.HV_NUM_OF_HYPOTHESES = 64
    torch.manual_seed(0)
    m = lib.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(None, hp).to(dev).eval()
    H, W = 96, 128
    imgs = [synth.make_image(i, H, W)[None] for i in range(3)]
    cats = []
    for i in range(3):
        c, _ = synth.make_vote_frame(i, K=3, H=H, W=W, rmin=8, rmax=20)
        cats.append({k: v.to(dev) for k, v in c.items()})
    st = FrameStreamer(m)
    st.prepare(imgs[0].to(dev), categorical_override=cats[0])
    nplan = len(st.models)

    def run(n, sync_each):
        acc = torch.zeros(4, dtype=torch.float64, device=dev)
        pending = []
        st._n = 0                                               # the same frame -> plan assignment in both runs

        def consume(out):
            # asynchronous readers on the caller's stream; the outputs are dropped when this returns
            acc[0] += out["logits"]["quaternion"].double().sum()
            acc[1] += out["categorical"]["mask"].double().sum()
            acc[2] += out["aggregated"]["xy"].double().sum()
            acc[3] += out["aggregated"]["RT"].double().sum()

        for i in range(n):
            torch.manual_seed(1000 + i)
            x = imgs[i % 3].to(dev, non_blocking=True)          # fresh block every frame ...
            pending.append(st.submit(x, categorical_override=cats[i % 3]))
            del x                                               # ... dropped while its frame is still queued
            junk = [torch.full((H * W * 24,), float(i), device=dev) for _ in range(3)]      # churn: the sizes of the logits
            del junk
            if len(pending) > nplan:
                consume(st.collect(pending.pop(0)))
                if sync_each:
                    torch.cuda.synchronize()
        while pending:
            consume(st.collect(pending.pop(0)))
        torch.cuda.synchronize()
        return acc.cpu()

    want = run(60, True)
    got = run(60, False)
    assert torch.equal(got, want), (got, want)


def test_forward_with_runtime_timing_and_report(lib, dev, capsys):
    """config.INFERENCE defaults (RUNTIME_TIMING=True): the stage-by-stage path with the reference's six
    timers; whole forward() on a small frame with random weights (many tiny instances), schema check."""
    from fastposecnn_amd import config, synth
    hp = config.INFERENCE()
    hp.HV_NUM_OF_HYPOTHESES = 32
    torch.manual_seed(0)
    m = lib.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(None, hp).to(dev).eval()
    x = torch.stack([synth.make_image(i, 64, 96) for i in range(2)]).to(dev)
    with torch.no_grad():
        out = m(x)
        out2 = m(x)
    agg = out["aggregated"]
    n = agg["class_ids"].shape[0]
    assert n == out2["aggregated"]["class_ids"].shape[0]
    for k, shp in (("instance_masks", (n, 64, 96)), ("quaternion", (n, 4)), ("scales", (n, 3)), ("xy", (n, 2)),
                   ("z", (n, 1)), ("R", (n, 3, 3)), ("T", (n, 3)), ("RT", (n, 4, 4)), ("hypothesis", (n, 1, 2)),
                   ("xy_mask", (n, 2, 64, 96)), ("sample_ids", (n,))):
        assert tuple(agg[k].shape) == shp, k
    if n:
        assert int(agg["sample_ids"].max()) <= 1 and int(agg["class_ids"].min()) >= 1
        # the instance masks partition the foreground of their images
        fg = (out["categorical"]["mask"] != 0).float()
        cover = torch.zeros_like(fg).index_add_(0, agg["sample_ids"], agg["instance_masks"])
        assert torch.equal(cover, fg)
    m.report_runtime()
    printed = capsys.readouterr().out
    for name in ("forward", "model", "Aggregation", "Hough Voting", "RT Calculation", "Class Compression"):
        assert name + ":" in printed
    hp.RUNTIME_TIMING = False
    m2 = lib.pose_regressor.MODELS['PoseRegressor'].construct_model(hp)       # resets the module-level timers
    assert not any(t.enabled for t in m2.TIMERS)


def test_graph_replay_is_bit_identical(lib, dev):
    """fpc_net_set_graph: on a non-default stream the frame-invariant launches are captured once and replayed;
    same plan (static tilings), so logits and categorical outputs must equal the plain-launch run bit for bit —
    over several frames with different images.  On the null stream the plan silently keeps plain launches."""
    from fastposecnn_amd import synth
    m, hp = _model(lib, dev, "resnet18")
    hp.ENGINE_AUTOTUNE = False
    m = m.to(dev)
    xs = [torch.stack([synth.make_image(i, 96, 128)]).to(dev) for i in range(3)]
    side = torch.cuda.Stream(device=dev)

    def run(graph):
        hp.ENGINE_GRAPH = graph
        m._drop_engines()
        outs = []
        with torch.no_grad(), torch.cuda.stream(side):
            for x in xs + xs:
                lg = m.pure_model_forward(x)
                cat = m.class_compression(lg)
                outs.append({k: v.clone() for k, v in lg.items()} | {"cat_" + k: v.clone() for k, v in cat.items()})
        side.synchronize()
        return outs

    plain, replay = run(False), run(True)
    for a, b in zip(plain, replay):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    with torch.no_grad():                      # null stream: not capturable, must still work
        m._drop_engines()
        lg = m.pure_model_forward(xs[0])
    torch.cuda.synchronize()
    assert torch.equal(lg["mask"], plain[0]["mask"])


@pytest.mark.parametrize("split", [True, False], ids=["split-precision", "plain-f32-products"])
def test_engine_split_precision_meets_the_f32_bar(lib, dev, split):
    """HPARAM.ENGINE_SPLIT_PRECISION (default on): autotuning may pick the bf16 x 3 form of the direct AND the Winograd
    convolutions (every f32 operand split exactly into three bf16 pieces, six partial products, f32 accumulation).  The
    whole network must stay within the same 1e-4 bar against the float64 CPU reference as with plain f32 products."""
    from fastposecnn_amd import synth
    m, hp = _model(lib, dev, "resnet18")
    hp.ENGINE_SPLIT_PRECISION = split
    x = torch.stack([synth.make_image(i, 64, 96) for i in range(2)])
    import copy
    ref_m = copy.deepcopy(m).double()
    ref_m.HPARAM = copy.copy(hp); ref_m.HPARAM.USE_NATIVE_ENGINE = False
    with torch.no_grad():
        ref = ref_m.pure_model_forward(x.double())
    m = m.to(dev)
    with torch.no_grad():
        out = m(x.to(dev))
    assert m._engines
    plans = next(iter(m._engines.values())).conv_plans()
    assert split or not any(p[2] == -5 for p in plans)          # -5 = the split-precision Winograd form
    for k in ("mask", "quaternion", "scales", "xy", "z"):
        got = out["logits"][k].cpu().double()
        err = (got - ref[k]).abs().max().item()
        assert err <= 1e-4 * max(1.0, ref[k].abs().max().item()), (k, err)


def test_split_precision_error_is_at_the_plain_f32_level(lib, dev):
    """Not just inside the 1e-4 bar: against float64 the network with split-precision products is as accurate as the one
    with plain f32 matrix products (each within 2x of the other), on a 96 x 128 input through every layer."""
    import copy
    from fastposecnn_amd import synth
    x = torch.stack([synth.make_image(i, 96, 128) for i in range(2)])
    errs = {}
    for split in (True, False):
        m, hp = _model(lib, dev, "resnet18")
        hp.ENGINE_SPLIT_PRECISION = split
        ref_m = copy.deepcopy(m).double()
        ref_m.HPARAM = copy.copy(hp); ref_m.HPARAM.USE_NATIVE_ENGINE = False
        with torch.no_grad():
            ref = ref_m.pure_model_forward(x.double())
            out = m.to(dev)(x.to(dev))
        errs[split] = max((out["logits"][k].cpu().double() - ref[k]).abs().max().item() / max(1.0, ref[k].abs().max().item())
                          for k in ("mask", "quaternion", "scales", "xy", "z"))
    assert errs[True] <= 2.0 * errs[False] + 1e-7 and errs[False] <= 2.0 * errs[True] + 1e-7, errs
    assert errs[True] <= 2e-5, errs


def test_fp16_pieces_operand_range(lib, dev):
    """What wino_h2.hip's header states about operand ranges, measured: activations of scale 1 and 1e3 keep the 2e-5 bar of the other
    forms (weights are rescaled on the device whatever their scale: 1e-4 .. 1e2 here); activations of scale 1e-2 — every second piece
    subnormal — stay within 1e-5 of the output's scale; a value beyond fp16's range saturates: the output stays finite."""
    B, Cin, H, W, Cout = 1, 64, 24, 40, 128
    for form in (-8, -9):                                         # all four piece products / three of them (wino_h3.hip): the same statements
        g = torch.Generator().manual_seed(5)
        for a_scale, w_scale, bar in ((1.0, 1.0, 2e-5), (1e3, 1e-4, 2e-5), (1.0, 1e2, 2e-5), (1e-2, 1.0, 1e-5)):
            x = torch.randn((B, Cin, H, W), generator=g) * a_scale
            w = torch.randn((Cout, Cin, 3, 3), generator=g) * (w_scale / (Cin * 9) ** 0.5)
            out, _, plan = _conv2d(dev, x, w, 1, 1, nsplit=form)
            ref = _ref_conv(x, w, 1, 1)
            err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
            assert err <= bar, (form, a_scale, w_scale, err)
        x = torch.randn((B, Cin, H, W), generator=g)
        x[0, 3, 5, 7] = 1.0e6                                        # beyond 2 x 65504: saturates
        w = torch.randn((Cout, Cin, 3, 3), generator=g) / (Cin * 9) ** 0.5
        out, _, _ = _conv2d(dev, x, w, 1, 1, nsplit=form)
        assert torch.isfinite(out).all()
        far = torch.ones_like(out, dtype=torch.bool)
        far[:, :, 3:8, 5:10] = False                                  # outputs the huge value does not reach are unaffected
        ref = _ref_conv(x, w, 1, 1)
        assert ((out.double() - ref).abs()[far]).max().item() <= 2e-5 * ref[far].abs().max().item()


def test_every_winograd_site_on_fp16_pieces_meets_the_float64_bars(lib, dev):
    """The fp16 x 2 Winograd form (wino_h2.hip) FORCED onto every 3x3 / stride-1 site of the network (ResNet34: 29 encoder sites + 7 grouped
    decoder sites = 28 convolutions) — not just where the autotuner happens to pick it: the logits against the float64 CPU module path stay inside
    north_star's 1e-4 of each tensor's scale and within 4x of the all-bf16x3 error (two fp16 pieces carry 22 significant bits of an
    operand of ordinary scale; the bf16 x 3 pieces 24).  HPARAM.ENGINE_SPLIT_F16 = False keeps the form out of the plans."""
    import copy
    from fastposecnn_amd import synth
    x = torch.stack([synth.make_image(i, 160, 224) for i in range(2)])
    m, hp = _model(lib, dev, "resnet34")
    ref_m = copy.deepcopy(m).double()
    ref_m.HPARAM = copy.copy(hp); ref_m.HPARAM.USE_NATIVE_ENGINE = False
    with torch.no_grad():
        ref = ref_m.pure_model_forward(x.double())
    del ref_m
    m = m.to(dev)
    errs = {}
    for form in (9, 8, 7):
        with torch.no_grad():
            m(x.to(dev))
            eng = next(iter(m._engines.values()))
            n = eng.force_winograd(form)
            assert n >= 30, n            # 29 encoder sites + the 7 decoder sites (one grouped launch for the four decoders each)
            assert sum(1 for p in eng.conv_plans() if p[2] == -form) >= n
            out = m(x.to(dev))
        errs[form] = max((out["logits"][k].cpu().double() - ref[k]).abs().max().item() / max(1.0, ref[k].abs().max().item())
                         for k in ("mask", "quaternion", "scales", "xy", "z"))
    assert errs[9] <= 1e-4 and errs[8] <= 1e-4 and errs[7] <= 1e-4, errs
    assert errs[8] <= 2e-5 and errs[8] <= 4.0 * errs[7] + 1e-6, errs
    # three of the four piece products (wino_h3.hip): the dropped one is <= 2^-22 of the term, like the two every two-piece form drops
    assert errs[9] <= 2e-5 and errs[9] <= 6.0 * errs[7] + 1e-6, errs
    m2, hp2 = _model(lib, dev, "resnet34")
    hp2.ENGINE_SPLIT_F16 = False
    m2 = m2.to(dev)
    with torch.no_grad():
        m2(x.to(dev))
    assert not any(p[2] == -8 for p in next(iter(m2._engines.values())).conv_plans())


def test_two_pass_merge_head_equals_the_one_pass_form_to_rounding(lib, dev, monkeypatch):
    """The engine applies the head in two passes (merge_split.hip: head of the three upsampled branches' sum at
    their own resolution, then head of the p2 branch + bias + the x2 upsample of that) instead of k_merge_head's one pass over
    the gathered 128-channel taps.  Same sum, another association: the logits agree to ~1e-6 of their scale, both forms meet
    the float64 bar, and a ragged map (H2 W2 not a multiple of the 4 x 32-pixel workgroup) is covered."""
    import copy
    from fastposecnn_amd import synth
    for (h, w) in ((96, 128), (160, 96)):
        x = torch.stack([synth.make_image(i, h, w) for i in range(3)])
        outs = {}
        for split in ("0", "1"):
            monkeypatch.setenv("FPC_MERGE_SPLIT", split)
            m, hp = _model(lib, dev, "resnet18")
            hp.ENGINE_GRAPH = False
            if split == "0":
                ref_m = copy.deepcopy(m).double()
                ref_m.HPARAM = copy.copy(hp); ref_m.HPARAM.USE_NATIVE_ENGINE = False
                with torch.no_grad():
                    ref = ref_m.pure_model_forward(x.double())
            with torch.no_grad():
                outs[split] = {k: v.cpu().double() for k, v in m.to(dev)(x.to(dev))["logits"].items()}
        for k in ("mask", "quaternion", "scales", "xy", "z"):
            scale = max(1.0, ref[k].abs().max().item())
            assert (outs["0"][k] - outs["1"][k]).abs().max().item() <= 2e-6 * scale, k
            for split in ("0", "1"):
                assert (outs[split][k] - ref[k]).abs().max().item() <= 1e-4 * scale, (k, split)


def test_frame_streamer_coalescing_returns_each_frames_own_result(lib, dev):
    """FrameStreamer(coalesce=2): consecutive single frames run as one batch-2 engine launch + one batched post-network
    enqueue; every ticket still yields ITS frame's forward() dict — logits / categorical equal to the frame's slice of the
    batched forward, instances split by sample id (frame-local sample ids), an odd frame at the end flushed on its own."""
    from fastposecnn_amd import config, synth
    from fastposecnn_amd.streaming import FrameStreamer
    import aggregation_layer as al
    hp = config.INFERENCE()
    hp.RUNTIME_TIMING = False
    hp.HV_NUM_OF_HYPOTHESES = 64
    torch.manual_seed(0)
    m = lib.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(None, hp).to(dev).eval()
    H, W = 96, 128
    xs = [synth.make_image(i, H, W)[None].to(dev) for i in range(3)]
    cats = []
    for i in range(3):
        c, _ = synth.make_vote_frame(i, K=2 + i, H=H, W=W, rmin=8, rmax=20)
        cats.append({k: v.to(dev) for k, v in c.items()})          # [1, ...] tensors: one frame each
    st = FrameStreamer(m, net_streams=2, coalesce=2)
    tickets = [st.submit(xs[i], categorical_override=cats[i]) for i in range(3)]
    st.flush()
    outs = [st.collect(t) for t in tickets]
    single = FrameStreamer(m, net_streams=1)
    for i in range(3):
        ref = single.collect(single.submit(xs[i], categorical_override=cats[i]))
        o = outs[i]
        assert tuple(o["logits"]["mask"].shape) == (1, 7, H, W) and tuple(o["categorical"]["mask"].shape) == (1, H, W)
        for k in ref["logits"]:
            err = (o["logits"][k] - ref["logits"][k]).abs().max().item()
            assert err <= 1e-4 * max(1.0, ref["logits"][k].abs().max().item()), (i, k, err)      # batch-2 plans may tile differently
        a, r = o["aggregated"], ref["aggregated"]
        assert a["class_ids"].tolist() == r["class_ids"].tolist() and len(a["class_ids"]) == 2 + i
        assert a["sample_ids"].tolist() == [0] * (2 + i)
        assert torch.equal(a["instance_masks"], r["instance_masks"])
        assert (a["quaternion"] - r["quaternion"]).abs().max().item() <= 1e-5
        assert tuple(a["xy"].shape) == (2 + i, 2) and tuple(a["RT"].shape) == (2 + i, 4, 4)
