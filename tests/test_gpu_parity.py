"""GPU parity: libfpc_hip.so (through the reference-shaped Python API, i.e. through the C ABI)
against the CPU oracle on identical inputs, against the committed golden vectors, and at full
640x480 size through size-independent properties.

Bars (BASELINE.json north_star): integer / index outputs bit-exact (class ids, labels, instance
order, pixel counts, inlier counts, winning hypothesis); floating point within 1e-4.
"""
import numpy as np
import pytest
import torch

from conftest import assert_pose, assert_rel, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def lib(dev):
    import fastposecnn_amd.lib as L          # puts the drop-in modules on sys.path
    from fastposecnn_amd import _native
    _native.lib()                             # raises if libfpc_hip.so is missing: no silent fallback
    return L


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ----------------------------------------------------------------------------- B1 kernels

def test_b1_kernels_bit_exact(lib, oracle, dev):
    import ransac_voting_gpu_layer.ransac_voting as ext
    rng = np.random.default_rng(0)
    for tn, vn, hn in [(500, 1, 64), (1777, 3, 130), (5, 1, 7)]:
        coords = np.stack([rng.integers(0, 640, tn), rng.integers(0, 480, tn)], 1).astype(np.float32)
        ang = rng.uniform(0, 2 * np.pi, (tn, vn))
        direct = np.stack([np.cos(ang), np.sin(ang)], -1).astype(np.float32)
        direct[::17] = 0                                       # zero votes (norm1 skip)
        direct[1::2, 0] = direct[0, 0]                         # many exactly parallel pairs
        idxs = rng.integers(0, tn, (hn, vn, 2)).astype(np.int32)
        idxs[0, :, 1] = idxs[0, :, 0]                          # identical pair -> degenerate
        hyp = ext.generate_hypothesis(T(direct, dev), T(coords, dev), T(idxs, dev))
        want = oracle.generate_hypothesis(direct, coords, idxs)
        assert np.array_equal(hyp.cpu().numpy(), want)
        inl = torch.zeros((hn, vn, tn), dtype=torch.uint8, device=dev)
        ext.voting_for_hypothesis(T(direct, dev), T(coords, dev), hyp, inl, 0.999)
        winl = np.zeros((hn, vn, tn), np.uint8)
        oracle.voting_for_hypothesis(direct, coords, want, winl, 0.999)
        assert np.array_equal(inl.cpu().numpy(), winl)
        inl.fill_(7)                                           # only ever writes 1
        ext.voting_for_hypothesis(T(direct, dev), T(coords, dev), hyp, inl, 0.999)
        assert set(torch.unique(inl).tolist()) <= {1, 7}


def test_b1_hand_derived_kernel_vectors(lib, dev):
    """tests/golden/kernel_kat.json (hand-derived from RV/src/ransac_voting_kernel.cu:22-48,100-125, no oracle
    involved): the B1 kernels, and the same decisions through the fused path's two-cone classification."""
    import json, os
    import ransac_voting_gpu_layer.ransac_voting as ext
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kernel_kat.json")) as f:
        kat = json.load(f)
    for c in kat["generate_hypothesis"]:
        hyp = ext.generate_hypothesis(T(np.asarray(c["direct"], np.float32)[:, None, :], dev),
                                      T(np.asarray(c["coords"], np.float32), dev),
                                      T(np.asarray([c["pair"]], np.int32)[:, None, :], dev))
        assert np.array_equal(hyp.cpu().numpy()[0, 0], np.asarray(c["expect"], np.float32)), c["name"]
    for c in kat["voting_for_hypothesis"]:
        inl = torch.zeros((1, 1, 1), dtype=torch.uint8, device=dev)
        ext.voting_for_hypothesis(T(np.asarray([[c["vote"]]], np.float32), dev), T(np.asarray([c["c"]], np.float32), dev),
                                  T(np.asarray([[c["h"]]], np.float32), dev), inl, float(np.float32(c["thresh"])))
        assert int(inl.item()) == c["inlier"], c["name"]


def test_b1_input_checks(lib, dev):
    import ransac_voting_gpu_layer.ransac_voting as ext
    d = torch.zeros((4, 1, 2), device=dev); c = torch.zeros((4, 2), device=dev)
    i = torch.zeros((3, 1, 2), dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError):
        ext.generate_hypothesis(d.cpu(), c, i)                 # CHECK_CUDA
    with pytest.raises(RuntimeError):
        ext.generate_hypothesis(torch.zeros((4, 1, 4), device=dev)[:, :, ::2], c, i)   # CHECK_CONTIGUOUS
    with pytest.raises(NotImplementedError):
        ext.generate_hypothesis_vanishing_point(d, c, i)


# ----------------------------------------------------------------------------- fused v3

def _run_v3(lib, dev, mask, vertex, hn, **kw):
    import ransac_voting_gpu_layer.ransac_voting_gpu as rvg
    out, dbg = rvg.ransac_voting_layer_v3(T(mask, dev), vertex, hn, return_debug=True, **kw)
    torch.cuda.synchronize()
    return out.cpu().numpy(), [{k: v.cpu().numpy() for k, v in d.items()} for d in dbg]


def _assert_v3_equal(out, dbg, want, wdbg):
    for k in ("tn", "win_idx", "win_count", "inlier_count"):
        assert np.array_equal(dbg[0][k], wdbg[0][k]), k
    assert np.array_equal(dbg[0]["hyp"], wdbg[0]["hyp"])       # same divisions, bit for bit
    assert np.array_equal(dbg[0]["counts"], wdbg[0]["counts"]) # every one of the hn x tn decisions agrees
    np.testing.assert_allclose(out, want, atol=1e-4, rtol=0)


def test_v3_golden_small(lib, oracle, dev):
    g = load_golden("vote_small.npz")
    xy = T(g["xy"], dev)
    vertex = xy.permute(0, 2, 3, 1).unsqueeze(3)               # strided view, as hough_voting.py:51
    out, dbg = _run_v3(lib, dev, g["mask"], vertex, int(g["hn"]), idxs=T(g["idxs"], dev))
    want, wdbg = oracle.ransac_voting_layer_v3(g["mask"], g["xy"].transpose(0, 2, 3, 1)[:, :, :, None, :],
                                               int(g["hn"]), idxs=g["idxs"], return_debug=True)
    _assert_v3_equal(out, dbg, want, wdbg)
    # the reference's own driver (fp32 torch.matmul normal equations there, fp64 here: measured 4e-7 relative)
    assert_rel(out, g["expected"], what="centre vs reference driver")
    assert np.array_equal(out[3], np.zeros((1, 2))) and np.array_equal(out[4], np.zeros((1, 2)))
    assert abs(out[6, 0, 1] - 2.0 / 3.0) < 1e-6                # rank-1 normal equations -> pinverse


def test_v3_golden_thinning(lib, oracle, dev):
    g = load_golden("vote_thin.npz")
    vertex = T(g["xy"], dev).permute(0, 2, 3, 1).unsqueeze(3)
    out, dbg = _run_v3(lib, dev, g["mask"], vertex, int(g["hn"]), idxs=T(g["idxs"], dev), keep=T(g["keep"], dev),
                       max_num=int(g["max_num"]))
    want, wdbg = oracle.ransac_voting_layer_v3(g["mask"], g["xy"].transpose(0, 2, 3, 1)[:, :, :, None, :],
                                               int(g["hn"]), idxs=g["idxs"], keep=g["keep"],
                                               max_num=int(g["max_num"]), return_debug=True)
    _assert_v3_equal(out, dbg, want, wdbg)
    assert_rel(out, g["expected"], what="centre vs reference driver")


def test_v3_golden_fullres(lib, oracle, dev):
    g = load_golden("vote_fullres.npz")
    H, W = int(g["H"]), int(g["W"])
    mask = np.zeros((2, H * W), np.float32); xy = np.zeros((2, 2, H * W), np.float32)
    for i in range(2):
        mask[i, g[f"pix{i}"]] = 1
        xy[i, 0, g[f"pix{i}"]] = g[f"dir{i}"][:, 0]; xy[i, 1, g[f"pix{i}"]] = g[f"dir{i}"][:, 1]
    mask = mask.reshape(2, H, W); xy = xy.reshape(2, 2, H, W)
    vertex = T(xy, dev).permute(0, 2, 3, 1).unsqueeze(3)
    out, dbg = _run_v3(lib, dev, mask, vertex, int(g["hn"]), idxs=T(g["idxs"], dev))
    want, wdbg = oracle.ransac_voting_layer_v3(mask, xy.transpose(0, 2, 3, 1)[:, :, :, None, :], int(g["hn"]),
                                               idxs=g["idxs"], return_debug=True)
    _assert_v3_equal(out, dbg, want, wdbg)
    assert_rel(out, g["expected"], what="centre vs reference driver")


def test_v3_builtin_sampler_matches_oracle_stream(lib, oracle, dev):
    """No injected idxs: the HIP sampler and the oracle share include/fpc_rng.h, so equal seeds
    give equal samples, and the built-in > max_num thinning keeps the same pixels."""
    g = load_golden("vote_small.npz")
    vertex = T(g["xy"], dev).permute(0, 2, 3, 1).unsqueeze(3)
    for kw in (dict(seed=11), dict(seed=12, max_num=100)):
        out, dbg = _run_v3(lib, dev, g["mask"], vertex, 200, **kw)
        want, wdbg = oracle.ransac_voting_layer_v3(g["mask"], g["xy"].transpose(0, 2, 3, 1)[:, :, :, None, :], 200,
                                                   return_debug=True, **kw)
        _assert_v3_equal(out, dbg, want, wdbg)
    assert (dbg[0]["tn"][:2] < (g["mask"][:2] != 0).sum((1, 2))).all()       # thinned


def test_v3_edge_shapes(lib, dev):
    import ransac_voting_gpu_layer.ransac_voting_gpu as rvg
    out = rvg.ransac_voting_layer_v3(torch.zeros((0, 8, 8), device=dev), torch.zeros((0, 8, 8, 1, 2), device=dev), 16)
    assert tuple(out.shape) == (0, 1, 2)
    out = rvg.ransac_voting_layer_v3(torch.zeros((2, 9, 7), device=dev), torch.zeros((2, 9, 7, 1, 2), device=dev), 16)
    assert torch.equal(out.cpu(), torch.zeros((2, 1, 2)))
    with pytest.raises(RuntimeError):
        rvg.ransac_voting_layer_v3(torch.zeros((1, 8, 8)), torch.zeros((1, 8, 8, 1, 2)), 16)   # CPU tensors: no fallback
    # vn = 2 keypoints, odd plane size (scalar tail of the 4-pixel chunks), bool mask
    H, W = 13, 11
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    v = np.zeros((1, H, W, 2, 2), np.float32)
    for k, (cx, cy) in enumerate([(4.5, 6.25), (8.0, 3.0)]):
        d = np.stack([cx - xx, cy - yy], -1); d /= np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-9)
        v[0, :, :, k, :] = d
    out = rvg.ransac_voting_layer_v3(torch.ones((1, H, W), dtype=torch.bool, device=dev), T(v, dev), 64, seed=1)
    np.testing.assert_allclose(out.cpu().numpy()[0], [[4.5, 6.25], [8.0, 3.0]], atol=1e-3)


# ----------------------------------------------------------------------------- class compression

def test_class_compress_golden_and_oracle(lib, oracle, dev):
    import gpu_tensor_funcs as gtf
    g = load_golden("class_compress.npz")
    C = int(g["num_classes"])
    logits_np = {k[3:]: v for k, v in g.items() if k.startswith("in_")}
    logits = {k: T(v, dev) for k, v in logits_np.items()}
    cat = gtf.class_compression_fused(C, logits)
    want = oracle.class_compress(logits_np, C)
    assert cat["mask"].dtype == torch.int64
    assert np.array_equal(cat["mask"].cpu().numpy(), g["out_mask"])         # ids bit-exact vs the reference
    assert np.array_equal(cat["mask"].cpu().numpy(), want["mask"])
    for k in ("scales", "z"):
        assert np.array_equal(cat[k].cpu().numpy(), g["out_" + k])          # pure selection: exact
    for k in ("quaternion", "xy"):
        assert cat[k].shape == tuple(g["out_" + k].shape)
        np.testing.assert_allclose(cat[k].cpu().numpy(), g["out_" + k], atol=1e-6, rtol=1e-6)
        np.testing.assert_allclose(cat[k].cpu().numpy(), want[k], atol=1e-6, rtol=1e-6)
    # reference signature with a caller-supplied mask
    cat2 = gtf.class_compress(C, T(g["in2_mask"], dev), logits)
    assert "mask" not in cat2
    for k in ("quaternion", "scales", "xy", "z"):
        np.testing.assert_allclose(cat2[k].cpu().numpy(), g["out2_" + k], atol=1e-6, rtol=1e-6)


def test_class_compress_fullsize_properties(lib, dev):
    import gpu_tensor_funcs as gtf
    torch.manual_seed(3)
    B, C, H, W = 2, 7, 480, 640
    logits = {"mask": torch.randn(B, C, H, W, device=dev), "quaternion": torch.randn(B, 24, H, W, device=dev),
              "scales": torch.randn(B, 18, H, W, device=dev), "xy": torch.randn(B, 12, H, W, device=dev),
              "z": torch.randn(B, 6, H, W, device=dev)}
    cat = gtf.class_compression_fused(C, logits)
    ref_mask = torch.argmax(torch.nn.LogSoftmax(dim=1)(logits["mask"]), dim=1)
    assert (cat["mask"] != ref_mask).sum().item() <= 2          # only rounding-level ties may differ
    fg = cat["mask"] != 0
    qn = cat["quaternion"].norm(dim=1)
    assert torch.allclose(qn[fg], torch.ones_like(qn[fg]), atol=1e-5) and (qn[~fg] == 0).all()
    sel = torch.gather(logits["z"], 1, (cat["mask"] - 1).clamp(min=0).unsqueeze(1)).squeeze(1) * fg
    assert torch.equal(cat["z"], sel)
    # idempotence of the compression given its own mask
    again = gtf.class_compress(C, cat["mask"], logits)
    for k in ("quaternion", "scales", "xy", "z"):
        assert torch.equal(again[k], cat[k])


# ----------------------------------------------------------------------------- connected components

# (40, 240, 320): 3000 blocks of 1024 pixels — the separate census scan (beyond 1024 blocks) with many tiny components;
# (1, 1024, 1024) = exactly 1024 blocks (the in-kernel scan's last entry); (5, 33, 31): 1023 pixels per image, blocks
# straddle images and the last block is partial; (2, 16, 2048): segments of 64 pixels inside one long row
@pytest.mark.parametrize("shape,p", [((3, 37, 53), 0.55), ((2, 64, 64), 0.6), ((1, 480, 640), 0.58), ((4, 40, 56), 0.3),
                                     ((40, 240, 320), 0.45), ((1, 1024, 1024), 0.5), ((5, 33, 31), 0.5), ((2, 16, 2048), 0.62),
                                     ((1, 1, 1), 1.0), ((7, 3, 5), 0.0)])
def test_cc_label_matches_oracle(lib, oracle, dev, shape, p):
    import aggregation_layer as al
    rng = np.random.default_rng(shape[1])
    fg = rng.random(shape) < p
    if p > 0.0:
        fg[0, :, 0] = True; fg[-1, -1, :] = True               # long vertical / horizontal runs
    layer = al.AggregationLayer(None, 7)
    labels, N = layer.batchwise_break_segmentation_mask(T(fg, dev))
    want, M = oracle.cc_label(fg)
    assert N == M and labels.dtype == torch.int32
    assert np.array_equal(labels.cpu().numpy(), want)


def test_cc_label_structures(lib, oracle, dev):
    import aggregation_layer as al
    layer = al.AggregationLayer(None, 7)
    H, W = 96, 128
    fg = np.zeros((3, H, W), bool)
    fg[0] = True                                               # one component covering the frame
    yy, xx = np.mgrid[0:H, 0:W]
    fg[1] = ((xx // 3 + yy // 3) % 2 == 0)                     # checkerboard of 3x3 cells: diagonal contacts only
    fg[2] = ((xx % 8 < 4) | (yy == H - 1))                     # comb: teeth joined by the LAST row (late merges)
    labels, N = layer.batchwise_break_segmentation_mask(T(fg, dev))
    want, M = oracle.cc_label(fg)
    assert N == M and np.array_equal(labels.cpu().numpy(), want)
    labels, N = layer.batchwise_break_segmentation_mask(torch.zeros((2, 8, 8), dtype=torch.bool, device=dev))
    assert N == 0 and int(labels.abs().sum()) == 0


def _pack_fg_words(fg):
    """bool [B,H,W] -> u64 words [B, ceil(HW / 4096) * 64] as the kernels lay them out (bit j of word w = pixel 64 w + j)."""
    B = fg.shape[0]
    flat = fg.reshape(B, -1)
    words = -(-flat.shape[1] // 4096) * 64
    pad = np.zeros((B, words * 64), bool)
    pad[:, :flat.shape[1]] = flat
    return np.packbits(pad.reshape(B, words, 64), axis=2, bitorder="little").view(np.uint64).reshape(B, words)


@pytest.mark.parametrize("shape,kind", [((1, 480, 640), "blobs"), ((3, 480, 640), "blobs+specks"), ((2, 64, 64), "noise"),
                                        ((1, 480, 640), "noise"), ((2, 96, 128), "structures"), ((33, 120, 192), "blobs+specks")])
def test_cc_label_from_bit_words(lib, oracle, dev, shape, kind):
    """fpc_cc_label_bits (round 4): whole-image run labelling from the foreground bit words, two launches.  Labels and
    their order bit-exact against the oracle (= scipy's numbering), root_pix = each component's first pixel; "noise" at
    640 x 480 has ~75 000 runs: the parent array leaves LDS for global memory."""
    import aggregation_layer as al
    B, H, W = shape
    rng = np.random.default_rng(B * 1000 + H)
    yy, xx = np.mgrid[0:H, 0:W]
    fg = np.zeros(shape, bool)
    if kind.startswith("blobs"):
        for b in range(B):
            for k in range(6):
                cy, cx, ry, rx = rng.integers(0, H), rng.integers(0, W), rng.integers(H // 16 + 2, H // 4), rng.integers(W // 16 + 2, W // 4)
                fg[b] |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        if "specks" in kind:
            fg ^= rng.random(shape) < 0.003
    elif kind == "noise":
        fg = rng.random(shape) < 0.58
    else:
        fg[0] = ((xx // 3 + yy // 3) % 2 == 0)
        fg[1] = ((xx % 8 < 4) | (yy == H - 1))
    from fastposecnn_amd import _native as nat
    assert nat.lib().fpc_cc_bits_supported(B, H, W) == 1
    cm = T(fg.astype(np.int64) * 3, dev)
    al.attach_fg_bits(cm)
    bits = al.fg_bits_of(cm)
    assert bits is not None and np.array_equal(bits.cpu().numpy().view(np.uint64), _pack_fg_words(fg))
    layer = al.AggregationLayer(None, 7)
    labels, N = layer.batchwise_break_segmentation_mask(cm)
    want, M = oracle.cc_label(fg)
    assert N == M and np.array_equal(labels.cpu().numpy(), want)
    root_pix = labels._fpc_root_pix[0].cpu().numpy()
    flat = want.reshape(-1)
    n_chk = min(M, root_pix.shape[0])
    first = np.full(M + 1, -1, np.int64)
    idx = np.nonzero(flat)[0]
    first[flat[idx][::-1]] = idx[::-1]                          # first (lowest) linear index of every label
    assert np.array_equal(root_pix[:n_chk], first[1:n_chk + 1])
    # the i64 entry takes the same path after converting the mask itself
    labels2, N2 = layer.batchwise_break_segmentation_mask(T(fg, dev))
    assert N2 == M and torch.equal(labels2, labels)


@pytest.mark.parametrize("shape,p", [((2, 64, 64), 0.0), ((1, 1, 64), 1.0), ((1, 1, 128), 0.5), ((300, 16, 64), 0.35), ((3, 480, 640), 1.0),
                                     ((2, 2, 4096), 0.5)])
def test_cc_label_bit_word_path_edge_shapes(lib, oracle, dev, shape, p):
    """The whole-image labelling on its edge cases: no foreground at all, a single row, more images than one sweep of the
    per-image component counts (300 > 256 threads), frames that are one component, one long row of many words."""
    import aggregation_layer as al
    from fastposecnn_amd import _native as nat
    assert nat.lib().fpc_cc_bits_supported(*shape) == 1
    rng = np.random.default_rng(shape[0] + shape[2])
    fg = rng.random(shape) < p if 0.0 < p < 1.0 else np.full(shape, p == 1.0)
    cm = al.attach_fg_bits(T(fg.astype(np.int64), dev))
    labels, N = al.AggregationLayer(None, 7).batchwise_break_segmentation_mask(cm)
    want, M = oracle.cc_label(fg)
    assert N == M and np.array_equal(labels.cpu().numpy(), want)


def test_class_compression_writes_the_foreground_bit_words(lib, dev):
    import aggregation_layer as al
    import gpu_tensor_funcs as gtf
    g = torch.Generator().manual_seed(5)
    B, C, H, W = 2, 7, 64, 128
    logits = {"mask": torch.randn(B, C, H, W, generator=g), "quaternion": torch.randn(B, 24, H, W, generator=g),
              "scales": torch.randn(B, 18, H, W, generator=g), "xy": torch.randn(B, 12, H, W, generator=g),
              "z": torch.randn(B, 6, H, W, generator=g)}
    cat = gtf.class_compression_fused(C, {k: v.to(dev) for k, v in logits.items()})
    bits = al.fg_bits_of(cat["mask"])
    assert bits is not None
    assert np.array_equal(bits.cpu().numpy().view(np.uint64), _pack_fg_words(cat["mask"].cpu().numpy() != 0))
    cat["mask"][0, 0, 0] = 1                                     # an in-place write retires the words
    assert al.fg_bits_of(cat["mask"]) is None


# ----------------------------------------------------------------------------- aggregation / pose

def test_aggregate_golden(lib, oracle, dev):
    import aggregation_layer as al
    g = load_golden("aggregate.npz")
    cat = {k[3:]: T(v, dev) for k, v in g.items() if k.startswith("in_")}
    agg = al.AggregationLayer(None, 7).forward(cat)
    assert np.array_equal(agg["class_ids"].cpu().numpy(), g["out_class_ids"].astype(np.int64))
    assert np.array_equal(agg["sample_ids"].cpu().numpy(), g["out_sample_ids"])
    assert np.array_equal(agg["instance_masks"].cpu().numpy(), g["out_instance_masks"])
    assert np.array_equal(agg["xy"].cpu().numpy(), g["out_xy"])
    for k in ("quaternion", "scales", "z"):
        assert tuple(agg[k].shape) == g["out_" + k].shape
        np.testing.assert_allclose(agg[k].cpu().numpy(), g["out_" + k], atol=1e-5, rtol=1e-5)
    e = load_golden("aggregate_empty.npz")
    agg0 = al.AggregationLayer(None, 7).forward({k: torch.zeros_like(v) for k, v in cat.items()})
    for k in ("class_ids", "sample_ids", "instance_masks", "quaternion", "scales", "xy", "z"):
        assert tuple(agg0[k].shape) == e["out_" + k].shape, k


def test_pose_rt_golden(lib, oracle, dev):
    import gpu_tensor_funcs as gtf
    g = load_golden("pose_rt.npz")
    R, Tt, RT = gtf.batchwise_get_RT(T(g["q"], dev), T(g["xy"], dev), T(g["z"], dev), T(g["Kinv"], dev))
    wR, wT, wRT = oracle.pose_rt(g["q"], g["xy"], g["z"], g["Kinv"])
    np.testing.assert_allclose(R.cpu().numpy(), wR, atol=1e-6); np.testing.assert_allclose(Tt.cpu().numpy(), wT, atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(RT.cpu().numpy(), wRT, atol=1e-5, rtol=1e-6)
    np.testing.assert_allclose(R.cpu().numpy(), g["R"], atol=1e-5)
    np.testing.assert_allclose(Tt.cpu().numpy(), g["T"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(RT.cpu().numpy(), g["RT"], atol=1e-4, rtol=1e-5)


def test_pipeline_golden(lib, oracle, dev):
    """logits -> class compression -> aggregation -> hough voting -> RT against the reference run."""
    import gpu_tensor_funcs as gtf
    import aggregation_layer as al
    import ransac_voting_gpu_layer.ransac_voting_gpu as rvg
    g = load_golden("pipeline.npz")
    C, hn = int(g["num_classes"]), int(g["hn"])
    logits = {k[7:]: T(v, dev) for k, v in g.items() if k.startswith("logits_")}
    cat = gtf.class_compression_fused(C, logits)
    assert np.array_equal(cat["mask"].cpu().numpy(), g["cat_mask"])
    agg = al.AggregationLayer(None, C).forward(cat)
    assert np.array_equal(agg["class_ids"].cpu().numpy(), g["agg_class_ids"].astype(np.int64))
    assert np.array_equal(agg["sample_ids"].cpu().numpy(), g["agg_sample_ids"])
    assert np.array_equal(agg["instance_masks"].cpu().numpy(), g["agg_instance_masks"])
    vertex = agg["xy"].permute(0, 2, 3, 1).unsqueeze(3)
    hyp = rvg.ransac_voting_layer_v3(agg["instance_masks"], vertex, hn, idxs=T(g["idxs"], dev))
    assert_rel(hyp.cpu().numpy(), g["agg_hypothesis"], what="pipeline centre vs reference driver")
    agg.update({"xy": hyp.squeeze(1)})
    agg = gtf.samplewise_get_RT(agg, T(g["Kinv"], dev))
    # north_star's bar per element: rotation entries absolute 1e-4, translations relative 1e-4 (z in millimetres, ~700)
    cpu = {k: agg[k].cpu().numpy() for k in ("R", "T", "RT", "quaternion", "scales", "z")}
    assert_pose(cpu["R"], cpu["T"], cpu["RT"], g["agg_R"], g["agg_T"], g["agg_RT"], what="pipeline golden")
    np.testing.assert_allclose(cpu["quaternion"], g["agg_quaternion"], atol=1e-4, rtol=0)
    assert_rel(cpu["scales"], g["agg_scales"], what="scales")
    assert_rel(cpu["z"], g["agg_z"], what="z")


# ----------------------------------------------------------------------------- full size

def test_fullsize_vote_bench_frame(lib, oracle, dev):
    """The 640x480 bench fixture (6 instances, hn = 1000): HIP == oracle on every integer output,
    fused counts == column sums of the B1 inlier matrix, and the centres are recovered."""
    import aggregation_layer as al
    import ransac_voting_gpu_layer.ransac_voting_gpu as rvg
    import ransac_voting_gpu_layer.ransac_voting as ext
    from fastposecnn_amd import synth
    cat_cpu, centres = synth.make_vote_frame(0)
    cat = {k: v.to(dev) for k, v in cat_cpu.items()}
    agg = al.AggregationLayer(None, 7).forward(cat)
    want_agg = oracle.aggregate({k: v.numpy() for k, v in cat_cpu.items()})
    assert np.array_equal(agg["class_ids"].cpu().numpy(), want_agg["class_ids"])
    assert np.array_equal(agg["instance_masks"].cpu().numpy(), want_agg["instance_masks"])
    assert np.array_equal(agg["xy"].cpu().numpy(), want_agg["xy"])
    for k in ("quaternion", "scales", "z"):
        np.testing.assert_allclose(agg[k].cpu().numpy(), want_agg[k], atol=1e-5, rtol=1e-5)
    hn = 1000
    vertex = agg["xy"].permute(0, 2, 3, 1).unsqueeze(3)
    out, dbg = rvg.ransac_voting_layer_v3(agg["instance_masks"], vertex, hn, seed=2026, return_debug=True)
    want, wdbg = oracle.ransac_voting_layer_v3(want_agg["instance_masks"],
                                               want_agg["xy"].transpose(0, 2, 3, 1)[:, :, :, None, :], hn, seed=2026,
                                               return_debug=True)
    d = {k: v.cpu().numpy() for k, v in dbg[0].items()}
    for k in ("tn", "win_idx", "win_count", "inlier_count", "counts"):
        assert np.array_equal(d[k], wdbg[0][k]), k
    assert np.array_equal(d["hyp"], wdbg[0]["hyp"])
    np.testing.assert_allclose(out.cpu().numpy(), want, atol=1e-4, rtol=0)
    # size-independent cross-check through the B1 path for the largest instance
    i = int(np.argmax(d["tn"]))
    m = agg["instance_masks"][i].bool()
    coords = torch.nonzero(m).float()[:, [1, 0]].contiguous()
    direct = vertex[i].masked_select(m[:, :, None, None]).view(-1, 1, 2).contiguous()
    hyp = dbg[0]["hyp"][i].view(hn, 1, 2).contiguous()
    inl = torch.zeros((hn, 1, coords.shape[0]), dtype=torch.uint8, device=dev)
    ext.voting_for_hypothesis(direct, coords, hyp, inl, 0.999)
    assert torch.equal(inl.sum(2).view(-1).int(), dbg[0]["counts"][i])
    order = np.argsort([c[2] for c in centres])                 # centres listed by class 1..6
    got = {int(c): xy for c, xy in zip(agg["class_ids"].cpu().numpy(), out.cpu().numpy()[:, 0])}
    for j in order:
        cx, cy, cls, n = centres[j]
        assert abs(got[cls][0] - cx) < 0.5 and abs(got[cls][1] - cy) < 0.5


def test_config3_post_network_batch32_vs_oracle(lib, oracle, dev):
    """BASELINE.json configs[2] post-network side: the 32-frame vote-bench batch (192 instances, per-image seeds 0..31)
    through the model's deferred path (device-side instance count, capacity-sized buffers) against the oracle: every
    integer output bit-exact, centres / RT within tolerance."""
    from fastposecnn_amd import config, synth
    hp = config.INFERENCE()
    hp.RUNTIME_TIMING = False
    hp.HV_NUM_OF_HYPOTHESES = 128
    model = lib.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(None, hp).to(dev).eval()
    cat_cpu, _ = synth.make_vote_batch(range(32))
    cat = {k: v.to(dev) for k, v in cat_cpu.items()}
    agg = model.post_network_finish(model.post_network_enqueue(cat, seed=77))
    torch.cuda.synchronize()
    oracle.set_threads(0)
    try:
        want = oracle.aggregate({k: v.numpy() for k, v in cat_cpu.items()})
        wxy = oracle.ransac_voting_layer_v3(want["instance_masks"], want["xy"].transpose(0, 2, 3, 1)[:, :, :, None, :], 128, seed=77)
    finally:
        oracle.set_threads(1)
    n = want["class_ids"].shape[0]
    assert n == 192 and agg["class_ids"].shape[0] == n
    assert np.array_equal(agg["class_ids"].cpu().numpy(), want["class_ids"])
    assert np.array_equal(agg["sample_ids"].cpu().numpy(), want["sample_ids"])
    assert np.array_equal(agg["instance_masks"].cpu().numpy(), want["instance_masks"])
    assert_rel(agg["xy"].cpu().numpy(), wxy[:, 0], what="centres")
    R, T, RT = oracle.pose_rt(want["quaternion"], wxy[:, 0], want["z"], model.inv_intrinsics.cpu().numpy())
    assert_pose(agg["R"].cpu().numpy(), agg["T"].cpu().numpy(), agg["RT"].cpu().numpy(), R, T, RT, what="batch 32")
    # the deferred path appends the RT assembly to the vote's last kernel (fpc_ransac_voting_v3_pose): the same bits as the separate
    # launch (gtf.batchwise_get_RT -> fpc_pose_rt) on the same operands
    import gpu_tensor_funcs as gtf
    R2, T2, RT2 = gtf.batchwise_get_RT(agg["quaternion"], agg["xy"], agg["z"], model.inv_intrinsics)
    assert torch.equal(R2, agg["R"]) and torch.equal(T2, agg["T"]) and torch.equal(RT2, agg["RT"])


def test_config3_vote_batch32_hn1000_every_count_vs_oracle(lib, oracle, dev):
    """BASELINE.json configs[2]'s vote at its actual setting (hn = 1000, 192 instances, one of them above max_num): the
    192 000 exact inlier counts, the winners and their inlier sets equal the oracle's, bit for bit."""
    from fastposecnn_amd import synth
    import aggregation_layer as al
    import ransac_voting_gpu_layer.ransac_voting_gpu as rvg
    cat_cpu, _ = synth.make_vote_batch(range(32))
    cat = {k: v.to(dev) for k, v in cat_cpu.items()}
    agg = al.AggregationLayer(None, 7).forward(cat)
    vertex = agg["xy"].permute(0, 2, 3, 1).unsqueeze(3)
    out, dbg = rvg.ransac_voting_layer_v3(agg["instance_masks"], vertex, 1000, seed=31, return_debug=True)
    torch.cuda.synchronize()
    oracle.set_threads(0)
    try:
        want = oracle.aggregate({k: v.numpy() for k, v in cat_cpu.items()})
        wxy, wdbg = oracle.ransac_voting_layer_v3(want["instance_masks"], want["xy"].transpose(0, 2, 3, 1)[:, :, :, None, :], 1000,
                                                  seed=31, return_debug=True)
    finally:
        oracle.set_threads(1)
    d = {k: v.cpu().numpy() for k, v in dbg[0].items()}
    assert d["counts"].shape == (192, 1000) and (wdbg[0]["tn"] < (want["instance_masks"] != 0).sum((1, 2))).any()   # one is thinned
    for k in ("tn", "win_idx", "win_count", "inlier_count", "counts"):
        assert np.array_equal(d[k], wdbg[0][k]), k
    assert np.array_equal(d["hyp"], wdbg[0]["hyp"], equal_nan=True)
    assert_rel(out.cpu().numpy(), wxy, what="centres")


# ----------------------------------------------------------------------------- vote filter soundness

@pytest.mark.parametrize("case", ["perfect", "noise", "scaled", "parallel_mix"])
@pytest.mark.parametrize("thresh", [0.5, 0.999, 0.99999])
def test_v3_filter_never_changes_the_answer(lib, oracle, dev, case, thresh):
    """The two cones of k_vote_count decide almost every pair without the reference's sqrt / divide; every count
    (hence the winner, lowest index on ties) must still be that of the exhaustive vote, also on tie-heavy and
    ill-conditioned inputs."""
    rng = np.random.default_rng(hash((case, thresh)) % 2 ** 32)
    H, W, n, hn = 72, 88, 3, 160
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    mask = np.zeros((n, H, W), np.float32); xy = np.zeros((n, 2, H, W), np.float32)
    for i, (cx, cy) in enumerate([(30.25, 20.5), (60.0, 50.0), (44.4, 36.6)]):
        m = ((xx - cx) ** 2 + (yy - cy) ** 2) <= (10 + 4 * i) ** 2
        d = np.stack([cx - xx, cy - yy]); nrm = np.maximum(np.sqrt((d ** 2).sum(0)), 1e-9)
        v = d / nrm
        if case == "noise":                       # pure noise: low counts, many ties, far hypotheses
            a = rng.uniform(0, 2 * np.pi, (H, W)); v = np.stack([np.cos(a), np.sin(a)])
        elif case == "scaled":                    # un-normalised votes over 6 decades, some exactly zero
            v = v * (10.0 ** rng.uniform(-3, 3, (H, W))); v[:, rng.random((H, W)) < 0.1] = 0
        elif case == "parallel_mix":              # mostly one direction (+-): near-parallel pairs, huge |h|
            a = rng.normal(0.3, 1e-4, (H, W)) + np.pi * (rng.random((H, W)) < 0.5)
            v = np.stack([np.cos(a), np.sin(a)])
        mask[i] = m; xy[i] = (v * m).astype(np.float32)
    vertex = T(xy, dev).permute(0, 2, 3, 1).unsqueeze(3)
    out, dbg = _run_v3(lib, dev, mask, vertex, hn, seed=5, inlier_thresh=thresh)
    want, wdbg = oracle.ransac_voting_layer_v3(mask, xy.transpose(0, 2, 3, 1)[:, :, :, None, :], hn, seed=5,
                                               inlier_thresh=thresh, return_debug=True)
    for k in ("tn", "win_idx", "win_count", "inlier_count", "counts"):
        assert np.array_equal(dbg[0][k], wdbg[0][k]), k
    assert np.array_equal(dbg[0]["hyp"], wdbg[0]["hyp"], equal_nan=True)
    np.testing.assert_allclose(out, want, atol=1e-4, rtol=1e-6)


def test_v3_exact_mode_for_nonpositive_threshold(lib, oracle, dev):
    g = load_golden("vote_small.npz")
    vertex = T(g["xy"], dev).permute(0, 2, 3, 1).unsqueeze(3)
    for th in (0.0, -0.5):
        out, dbg = _run_v3(lib, dev, g["mask"], vertex, 64, seed=3, inlier_thresh=th)
        want, wdbg = oracle.ransac_voting_layer_v3(g["mask"], g["xy"].transpose(0, 2, 3, 1)[:, :, :, None, :], 64,
                                                   seed=3, inlier_thresh=th, return_debug=True)
        _assert_v3_equal(out, dbg, want, wdbg)


# ----------------------------------------------------------------------------- progressive count (exact pruning)

@pytest.fixture
def prune_forced(lib):
    """The progressive count whenever the count rows are not an output (by default only large batches take it)."""
    import ransac_voting_gpu_layer.ransac_voting_gpu as rvg
    rvg.set_vote_prune(1, (5, 10))
    yield rvg
    rvg.set_vote_prune(0, (5, 10))


def _winner_equal(a, da, b, db, what):
    for k in ("tn", "win_idx", "win_count", "inlier_count"):
        assert torch.equal(da[0][k], db[0][k]), (what, k)
    assert torch.equal(da[0]["hyp"], db[0]["hyp"]) or np.array_equal(da[0]["hyp"].cpu().numpy(), db[0]["hyp"].cpu().numpy(), equal_nan=True)
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy(), equal_nan=True), what      # same inlier set, same fp64 sums: bit-identical


@pytest.mark.parametrize("sched", [(5, 10), (8,), (3, 6, 11), (1, 15)])
def test_progressive_count_bench_frames_equal_exhaustive_and_oracle(lib, oracle, dev, prune_forced, sched):
    """Four frames of the vote-bench fixture (24 instances, hn = 1000): the winner, its count, the inlier set and the
    refined centre of the progressive count equal the exhaustive count's bit for bit, for several pass schedules, and the
    oracle's; the per-instance record shows that hypotheses were in fact dropped."""
    import aggregation_layer as al
    from fastposecnn_amd import synth
    rvg = prune_forced
    rvg.set_vote_prune(1, sched)
    cat_cpu, _ = synth.make_vote_batch(range(4))
    agg = al.AggregationLayer(None, 7).forward({k: v.to(dev) for k, v in cat_cpu.items()})
    vertex = agg["xy"].permute(0, 2, 3, 1).unsqueeze(3)
    masks = agg["instance_masks"]
    n, hn = masks.shape[0], 1000
    a, da = rvg.ransac_voting_layer_v3(masks, vertex, hn, seed=77, return_debug="winner")
    info = rvg.vote_prune_info(n, 480, 640, hn, dev).cpu().numpy()
    b, db = rvg.ransac_voting_layer_v3(masks, vertex, hn, seed=77, return_debug=True)            # count rows: exhaustive
    torch.cuda.synchronize()
    _winner_equal(a, da, b, db, sched)
    counts = db[0]["counts"].cpu().numpy()
    assert np.array_equal(da[0]["win_count"].cpu().numpy(), counts.max(1))
    assert np.array_equal(da[0]["win_idx"].cpu().numpy(), counts.argmax(1))                       # first maximum
    # the record: alive set shrank, the leader's exact count is a count row entry and <= the winner's
    assert (info[:, 2] <= hn).all() and (info[:, 2] >= 1).all() and (sched == (1, 15) or (info[:, 2] < hn).all())
    lead, L = info[:, 4], info[:, 5]
    assert np.array_equal(counts[np.arange(n), lead], L) and (L <= counts.max(1)).all()
    if sched == (5, 10):
        want = oracle.aggregate({k: v.numpy() for k, v in cat_cpu.items()})
        oracle.set_threads(0)
        try:
            wxy, wdbg = oracle.ransac_voting_layer_v3(want["instance_masks"], want["xy"].transpose(0, 2, 3, 1)[:, :, :, None, :],
                                                      hn, seed=77, return_debug=True)
        finally:
            oracle.set_threads(1)
        for k in ("tn", "win_idx", "win_count", "inlier_count"):
            assert np.array_equal(da[0][k].cpu().numpy(), wdbg[0][k]), k
        assert_rel(a.cpu().numpy(), wxy, what="centres")


@pytest.mark.parametrize("case", ["ties", "near_ties", "noise", "thinned", "tiny_and_empty"])
def test_progressive_count_adversarial(lib, oracle, dev, prune_forced, case):
    """Inputs built against the pruning rule: exact ties between many hypotheses (the LOWEST index must win, also when
    it is not the leader of an early pass), winners that lead only in the last pass, pure-noise votes with low counts,
    an instance above max_num (the bound counts kept pixels only), instances of fewer pixels than a count unit and
    empty ones.  Oracle and exhaustive count agree with the progressive count on every integer output."""
    rvg = prune_forced
    rng = np.random.default_rng({"ties": 1, "near_ties": 2, "noise": 3, "thinned": 4, "tiny_and_empty": 5}[case])
    H, W, hn = 96, 128, 320
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    specs = [(40.3, 30.6, 26), (90.0, 60.0, 30), (64.5, 48.5, 44)]
    if case == "tiny_and_empty":
        specs = [(20.0, 20.0, 3), (60.0, 50.0, 0), (100.2, 70.7, 20), (30.0, 80.0, 1)]
    n = len(specs)
    mask = np.zeros((n, H, W), np.float32); xy = np.zeros((n, 2, H, W), np.float32)
    idxs = None
    kw = {}
    for i, (cx, cy, r) in enumerate(specs):
        m = ((xx - cx) ** 2 + (yy - cy) ** 2) <= r * r if r > 0 else np.zeros((H, W), bool)
        d = np.stack([cx - xx, cy - yy]); v = d / np.maximum(np.sqrt((d ** 2).sum(0)), 1e-9)
        ang = rng.normal(0, 0.03, (H, W))
        v = np.stack([np.cos(ang) * v[0] - np.sin(ang) * v[1], np.sin(ang) * v[0] + np.cos(ang) * v[1]])
        if case == "noise":
            a = rng.uniform(0, 2 * np.pi, (H, W)); v = np.stack([np.cos(a), np.sin(a)])
        if case == "near_ties":
            # the lower half of the blob votes for a second centre 1.5 px away: two families of hypotheses whose counts
            # cross between the passes (units are visited in a scattered order)
            d2 = np.stack([cx + 1.5 - xx, cy - yy]); v2 = d2 / np.maximum(np.sqrt((d2 ** 2).sum(0)), 1e-9)
            v = np.where((yy > cy)[None], v2, v)
        mask[i] = m; xy[i] = (v * m).astype(np.float32)
    if case in ("ties", "near_ties"):
        # injected pairs: blocks of identical pairs -> identical hypotheses -> exactly equal counts at different indices
        tn = (mask != 0).sum((1, 2))
        idxs = np.zeros((n, hn, 1, 2), np.int32)
        for i in range(n):
            base = rng.integers(0, tn[i], (hn // 8, 2))
            idxs[i, :, 0, :] = np.repeat(base, 8, axis=0)[rng.permutation(hn)]
        kw["idxs"] = idxs
    if case == "thinned":
        kw["max_num"] = 1500
    vertex = T(xy, dev).permute(0, 2, 3, 1).unsqueeze(3)
    ikw = dict(kw)
    if idxs is not None:
        ikw["idxs"] = T(idxs, dev)
    a, da = rvg.ransac_voting_layer_v3(T(mask, dev), vertex, hn, seed=9, return_debug="winner", **ikw)
    b, db = rvg.ransac_voting_layer_v3(T(mask, dev), vertex, hn, seed=9, return_debug=True, **ikw)
    torch.cuda.synchronize()
    _winner_equal(a, da, b, db, case)
    want, wdbg = oracle.ransac_voting_layer_v3(mask, xy.transpose(0, 2, 3, 1)[:, :, :, None, :], hn, seed=9, return_debug=True, **kw)
    for k in ("tn", "win_idx", "win_count", "inlier_count"):
        assert np.array_equal(da[0][k].cpu().numpy(), wdbg[0][k]), k
    np.testing.assert_allclose(a.cpu().numpy(), want, atol=1e-4, rtol=1e-6)
    if case == "ties":
        counts = wdbg[0]["counts"]
        assert all((counts[i] == counts[i].max()).sum() >= 8 for i in range(n))      # the fixture does produce ties at the top


def test_progressive_count_is_off_unless_asked_for(lib, dev, prune_forced):
    """fpc_vote_set_prune(0) (the default): the record of a call shows the full alive set; mode 1: a shrunken one."""
    import aggregation_layer as al
    from fastposecnn_amd import synth
    rvg = prune_forced
    cat_cpu, _ = synth.make_vote_batch(range(2))
    agg = al.AggregationLayer(None, 7).forward({k: v.to(dev) for k, v in cat_cpu.items()})
    vertex = agg["xy"].permute(0, 2, 3, 1).unsqueeze(3)
    n, hn = agg["instance_masks"].shape[0], 1000
    for mode, pruned in ((0, False), (1, True)):
        rvg.set_vote_prune(mode)
        rvg.ransac_voting_layer_v3(agg["instance_masks"], vertex, hn, seed=1)
        info = rvg.vote_prune_info(n, 480, 640, hn, dev).cpu().numpy()
        assert ((info[:, 2] < hn).all()) == pruned and (info[:, 2] == hn).all() == (not pruned)


# ----------------------------------------------------------------------------- deferred post-network path

def test_deferred_post_network_equals_staged(lib, oracle, dev):
    """Model.agg_hough_and_generate_RT on capacity-sized buffers with the instance count kept on the
    device (one host read at the end) returns exactly what the stage-by-stage path returns."""
    from fastposecnn_amd import config, synth
    hp = config.INFERENCE()
    hp.RUNTIME_TIMING = False
    hp.HV_NUM_OF_HYPOTHESES = 200
    model = lib.pose_regressor.MODELS['PoseRegressor'].load_from_ckpt(None, hp).to(dev).eval()
    for frames, cap in (([0, 1], 32), ([2], 2), ([3], 32)):            # cap 2 < 6 instances: re-run at exact size
        hp.MAX_INSTANCES = cap
        cat_cpu, _ = synth.make_vote_batch(frames)
        if frames == [3]:
            cat_cpu = {k: torch.zeros_like(v) for k, v in cat_cpu.items()}      # no instance at all
        cat = {k: v.to(dev) for k, v in cat_cpu.items()}
        torch.manual_seed(5)
        fused = model.agg_hough_and_generate_RT(cat)
        torch.manual_seed(5)
        staged = model.perform_RT_calculation(model.hough_voting(model.aggregate(cat)))
        assert set(fused) == set(staged)
        for k in staged:
            assert fused[k].shape == staged[k].shape and fused[k].dtype == staged[k].dtype, k
            assert torch.equal(fused[k], staged[k]), k
        assert fused["class_ids"].shape[0] == (0 if frames == [3] else 6 * len(frames))


# ----------------------------------------------------------------------------- randomised whole-path parity

@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_scenes_post_network_vs_oracle(lib, oracle, dev, seed):
    """Random small scenes (ragged sizes, touching multi-class blobs, specks below min_num, empty images,
    a blob above max_num) through compress -> CC -> aggregate -> vote -> RT: integer outputs bit-exact
    against the oracle, floating point within 1e-4."""
    import gpu_tensor_funcs as gtf
    import aggregation_layer as al
    import ransac_voting_gpu_layer.ransac_voting_gpu as rvg
    rng = np.random.default_rng(seed)
    B = int(rng.integers(1, 4))
    H = int(rng.integers(17, 70)); W = int(rng.integers(19, 90))
    C = 7
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    mask_logits = rng.normal(0, 0.1, (B, C, H, W)).astype(np.float32)
    mask_logits[:, 0] += 2.0                                               # background wins by default
    xy = rng.normal(0, 1, (B, 12, H, W)).astype(np.float32)
    for b in range(B):
        if seed == 3 and b == 0:
            continue                                                       # one empty image
        for k in range(int(rng.integers(1, 5))):
            cx, cy = rng.uniform(4, W - 4), rng.uniform(4, H - 4)
            r = rng.uniform(1.0, min(H, W) / 3)
            cls = int(rng.integers(1, C))
            blob = ((xx - cx) ** 2 + (yy - cy) ** 2) <= r * r
            mask_logits[b, cls][blob] += 6.0 + k                           # later blobs win overlaps: touching classes
            d = np.stack([cx - xx, cy - yy]); d /= np.maximum(np.sqrt((d ** 2).sum(0)), 1e-6)
            ang = rng.normal(0, 0.02, (H, W)).astype(np.float32)
            vx = np.cos(ang) * d[0] - np.sin(ang) * d[1]; vy = np.sin(ang) * d[0] + np.cos(ang) * d[1]
            xy[b, 2 * (cls - 1)][blob] = vx[blob] * 3.0                    # un-normalised: compression normalises
            xy[b, 2 * (cls - 1) + 1][blob] = vy[blob] * 3.0
    logits_np = {"mask": mask_logits, "quaternion": rng.normal(0, 1, (B, 24, H, W)).astype(np.float32),
                 "scales": rng.normal(0, 1, (B, 18, H, W)).astype(np.float32), "xy": xy,
                 "z": rng.normal(6, 0.2, (B, 6, H, W)).astype(np.float32)}
    cat = gtf.class_compression_fused(C, {k: T(v, dev) for k, v in logits_np.items()})
    wcat = oracle.class_compress(logits_np, C)
    assert np.array_equal(cat["mask"].cpu().numpy(), wcat["mask"])
    layer = al.AggregationLayer(None, C)
    agg = layer.forward(cat)
    want = oracle.aggregate(wcat)
    n = want["class_ids"].shape[0]
    assert agg["class_ids"].shape[0] == n
    if n == 0:
        return
    assert np.array_equal(agg["class_ids"].cpu().numpy(), want["class_ids"])
    assert np.array_equal(agg["sample_ids"].cpu().numpy(), want["sample_ids"])
    assert np.array_equal(agg["instance_masks"].cpu().numpy(), want["instance_masks"])
    for k in ("quaternion", "scales", "z"):
        np.testing.assert_allclose(agg[k].cpu().numpy(), want[k], atol=1e-5, rtol=1e-5)
    max_num = 60 if seed == 2 else 30000                                   # force the thinning path
    vertex = agg["xy"].permute(0, 2, 3, 1).unsqueeze(3)
    out, dbg = rvg.ransac_voting_layer_v3(agg["instance_masks"], vertex, 96, seed=seed + 5, max_num=max_num,
                                          return_debug=True)
    wxy, wdbg = oracle.ransac_voting_layer_v3(want["instance_masks"], want["xy"].transpose(0, 2, 3, 1)[:, :, :, None, :],
                                              96, seed=seed + 5, max_num=max_num, return_debug=True)
    d = {k: v.cpu().numpy() for k, v in dbg[0].items()}
    for k in ("tn", "win_idx", "win_count", "inlier_count", "counts"):
        assert np.array_equal(d[k], wdbg[0][k]), k
    assert np.array_equal(d["hyp"], wdbg[0]["hyp"], equal_nan=True)
    np.testing.assert_allclose(out.cpu().numpy(), wxy, atol=1e-4, rtol=1e-6)
    agg.update({"xy": out.squeeze(1)})
    kinv = np.linalg.inv(np.array([[577.5, 0, 319.5], [0, 577.5, 239.5], [0, 0, 1]], np.float32)).astype(np.float32)
    agg = gtf.samplewise_get_RT(agg, T(kinv, dev))
    R, Tt, RT = oracle.pose_rt(want["quaternion"], wxy[:, 0], want["z"], kinv)
    assert_pose(agg["R"].cpu().numpy(), agg["T"].cpu().numpy(), agg["RT"].cpu().numpy(), R, Tt, RT, what="adversarial frame")


# ----------------------------------------------------------------------------- matching

def _dicts(g, tag):
    return {k[len(tag) + 1:]: g[k] for k in list(g.keys()) if k.startswith(tag + "_")}


def test_mask_iou_golden_and_oracle(lib, oracle, dev):
    """gtf.batchwise_get_2d_iou through the C ABI: the reference's own output (golden) and the oracle on seeded
    stacks — ragged sizes (pixel count not a multiple of 64 / 256), f32 / bool / uint8 / int64 masks, NaN and
    -0.0 elements, empty masks (0/0 = NaN), n1 or n2 = 0.  Bit-exact: integer counts and one f32 division."""
    gtf = lib.gtf
    g = load_golden("matching.npz")
    got = gtf.batchwise_get_2d_iou(T(g["gts_instance_masks"], dev), T(g["preds_instance_masks"], dev))
    assert np.array_equal(got.cpu().numpy(), g["iou_all"], equal_nan=True)
    got2 = gtf.batchwise_get_2d_iou(T(g["gts2_instance_masks"], dev), T(g["preds2_instance_masks"], dev))
    assert np.array_equal(got2.cpu().numpy(), g["iou2"], equal_nan=True)
    rng = np.random.default_rng(5)
    for (n1, n2, H, W) in [(3, 5, 17, 23), (1, 1, 1, 1), (4, 2, 16, 16), (6, 7, 48, 64), (2, 9, 31, 64)]:
        a = (rng.random((n1, H, W)) < 0.4).astype(np.float32) * rng.normal(size=(n1, H, W)).astype(np.float32)
        b = (rng.random((n2, H, W)) < 0.3).astype(np.float32)
        a[0].flat[::7] = np.nan            # NaN counts as set
        b[-1][:] = 0; b[-1].flat[::5] = -0.0   # -0.0 does not
        want = oracle.mask_iou(a, b)
        assert np.array_equal(gtf.batchwise_get_2d_iou(T(a, dev), T(b, dev)).cpu().numpy(), want, equal_nan=True)
        ab, bb = (a != 0) | np.isnan(a), b != 0
        wantb = oracle.mask_iou(ab.astype(np.float32), bb.astype(np.float32))
        for cast in (lambda x: T(x, dev), lambda x: T(x.astype(np.uint8), dev), lambda x: T(x.astype(np.int64), dev)):
            assert np.array_equal(gtf.batchwise_get_2d_iou(cast(ab), cast(bb)).cpu().numpy(), wantb, equal_nan=True)
        assert np.array_equal(gtf.batchwise_get_2d_iou(T(a, dev), T(bb, dev)).cpu().numpy(), want, equal_nan=True)   # mixed dtypes
    e = gtf.batchwise_get_2d_iou(torch.zeros((0, 8, 8), device=dev), torch.zeros((3, 8, 8), device=dev))
    assert e.shape == (0, 3)
    e = gtf.batchwise_get_2d_iou(torch.zeros((2, 8, 8), device=dev), torch.zeros((0, 8, 8), device=dev))
    assert e.shape == (2, 0)
    with pytest.raises(RuntimeError):
        gtf.batchwise_get_2d_iou(torch.zeros((2, 8, 8)), torch.zeros((2, 8, 8)))        # CPU tensors: no fallback


def test_mask_iou_full_size_properties(lib, dev):
    """640x480, 24 x 24 masks (the size the oracle would need ~10 s for): IoU(a, a) = 1, symmetry,
    disjoint -> 0, nested -> |inner| / |outer|, and the optional count outputs against torch sums."""
    from fastposecnn_amd import _native as nat
    H, W, n = 480, 640, 24
    g = torch.Generator(device="cpu").manual_seed(3)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    ms = []
    for i in range(n):
        cx, cy, r = torch.randint(60, 580, (1,), generator=g), torch.randint(60, 420, (1,), generator=g), torch.randint(20, 120, (1,), generator=g)
        ms.append((((xx - cx) ** 2 + (yy - cy) ** 2) <= r * r).float())
    m = torch.stack(ms).to(dev)
    iou = lib.gtf.batchwise_get_2d_iou(m, m)
    assert torch.equal(torch.diagonal(iou), torch.ones(n, device=dev))
    assert torch.equal(iou, iou.t())
    mb = m.bool()
    inter = (mb[:, None] & mb[None]).flatten(2).sum(2)
    uni = (mb[:, None] | mb[None]).flatten(2).sum(2)
    assert torch.equal(iou, inter.float() / uni.float())
    L = nat.lib()
    o_i = torch.empty((n, n), dtype=torch.int32, device=dev); o_u = torch.empty_like(o_i); o = torch.empty((n, n), device=dev)
    ws = torch.empty(L.fpc_mask_iou_workspace_bytes(n, n, H * W), dtype=torch.uint8, device=dev)
    nat.check(L.fpc_mask_iou(nat.ptr(m), n, nat.ptr(m), n, H * W, 4, nat.ptr(o), nat.ptr(o_i), nat.ptr(o_u), nat.ptr(ws),
                             ws.numel(), nat.stream()), "fpc_mask_iou")
    assert torch.equal(o_i.long(), inter) and torch.equal(o_u.long(), uni) and torch.equal(o, iou)
    inner = torch.zeros((1, H, W), device=dev); inner[0, 100:200, 100:300] = 1
    outer = torch.zeros((1, H, W), device=dev); outer[0, 50:250, 50:350] = 1
    far = torch.zeros((1, H, W), device=dev); far[0, 300:400, 400:600] = 1
    assert lib.gtf.batchwise_get_2d_iou(inner, outer).item() == np.float32(100 * 200) / np.float32(200 * 300)
    assert lib.gtf.batchwise_get_2d_iou(inner, far).item() == 0.0


def test_find_matches_golden_and_oracle(lib, oracle, dev):
    """mg.batchwise_find_matches: the reference's outputs (golden, incl. the NaN row and the None cases) and
    the oracle on randomized AggData pairs — every key bit for bit (pure selection, no arithmetic)."""
    import matching as mg
    g = load_golden("matching.npz")
    td = lambda d: {k: T(v, dev) for k, v in d.items()}
    gts, preds, g2, p2 = (_dicts(g, t) for t in ("gts", "preds", "gts2", "preds2"))
    for (p, t, tag) in ((preds, gts, "out"), (p2, g2, "out2")):
        want = _dicts(g, tag)
        got = mg.batchwise_find_matches(td(p), td(t))
        assert sorted(got) == sorted(want)
        for k in want:
            assert got[k].dtype == torch.from_numpy(want[k]).dtype, k
            assert np.array_equal(got[k].cpu().numpy(), want[k]), k
    assert mg.batchwise_find_matches({k: v[:0] for k, v in td(preds).items()}, td(gts)) is None
    assert mg.batchwise_find_matches({k: v[5:6] for k, v in td(preds).items()}, td(gts)) is None
    assert mg.batchwise_find_matches(None, td(gts)) is None and mg.batchwise_find_matches(td(preds), {}) is None
    assert mg.batchwise_find_matches(td(preds), {k: v[:0] for k, v in td(gts).items()}) is None
    rng = np.random.default_rng(11)
    H, W = 60, 80
    yy, xx = np.mgrid[0:H, 0:W]
    for trial in range(6):
        def scene(n, seed):
            r = np.random.default_rng(seed)
            masks = np.stack([(((xx - r.integers(5, W - 5)) ** 2 + (yy - r.integers(5, H - 5)) ** 2) <= r.integers(3, 15) ** 2)
                              .astype(np.float32) for _ in range(n)]) if n else np.zeros((0, H, W), np.float32)
            return {"class_ids": r.integers(1, 4, n).astype(np.int64), "sample_ids": r.integers(0, 3, n).astype(np.int64),
                    "symmetric_ids": r.integers(0, 3, n).astype(np.int64), "instance_masks": masks,
                    "quaternion": r.normal(size=(n, 4)).astype(np.float32), "scales": r.random((n, 3)).astype(np.float32),
                    "xy": r.random((n, 2)).astype(np.float32), "z": r.random((n, 1)).astype(np.float32),
                    "R": r.normal(size=(n, 3, 3)).astype(np.float32), "T": r.normal(size=(n, 3)).astype(np.float32),
                    "RT": r.normal(size=(n, 4, 4)).astype(np.float32)}
        t_, p_ = scene(int(rng.integers(1, 12)), 100 + trial), scene(int(rng.integers(1, 12)), 200 + trial)
        want = oracle.find_matches(p_, t_)
        got = mg.batchwise_find_matches(td(p_), td(t_))
        assert (want is None) == (got is None)
        if want is not None:
            assert sorted(got) == sorted(want)
            for k in want:
                assert np.array_equal(got[k].cpu().numpy(), want[k]), (trial, k)


def test_pack_pose_records_native_equals_host_logic(lib, dev):
    """fpc_pack_pose_records (one launch) against parallel.pack_pose_records' torch path on CPU tensors (the path the
    world-size-2 gloo test covers): bit-identical buffer, incl. n = 0 and n = capacity; unpack round trip."""
    from fastposecnn_amd import parallel
    rng = np.random.default_rng(2)
    for n, cap in ((5, 8), (0, 4), (4, 4)):
        agg = {"sample_ids": torch.from_numpy(rng.integers(0, 3, n)), "class_ids": torch.from_numpy(rng.integers(1, 7, n)),
               "quaternion": torch.from_numpy(rng.normal(size=(n, 4)).astype(np.float32)),
               "scales": torch.from_numpy(rng.random((n, 3)).astype(np.float32)), "xy": torch.from_numpy(rng.random((n, 2)).astype(np.float32)),
               "z": torch.from_numpy(rng.random((n, 1)).astype(np.float32)), "R": torch.from_numpy(rng.normal(size=(n, 3, 3)).astype(np.float32)),
               "T": torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32)), "RT": torch.from_numpy(rng.normal(size=(n, 4, 4)).astype(np.float32))}
        want = parallel.pack_pose_records(agg, sample_offset=32, capacity=cap)
        got = parallel.pack_pose_records({k: v.to(dev) for k, v in agg.items()}, sample_offset=32, capacity=cap)
        assert got.is_cuda and torch.equal(got.cpu().view(torch.int32), want.view(torch.int32))
        out = parallel.unpack_pose_records(got[None].cpu())
        assert torch.equal(out["sample_ids"], agg["sample_ids"] + 32) and torch.equal(out["RT"], agg["RT"])
    with pytest.raises(RuntimeError):
        parallel.pack_pose_records({k: v.to(dev) for k, v in agg.items()}, 0, capacity=2)


def _pack_bits(masks, H, W):
    """[n,H,W] masks -> the bit words fpc_aggregate_bits defines (numpy, little-endian bit order, zero padded to chunks)."""
    from fastposecnn_amd import _native as nat
    nw = nat.lib().fpc_mask_bits_words(H, W)
    n = masks.shape[0]
    flat = np.zeros((n, nw * 64), dtype=np.uint8)
    flat[:, :H * W] = (masks.reshape(n, -1) != 0)
    return np.packbits(flat, axis=1, bitorder="little").view(np.uint64).astype(np.int64).reshape(n, nw)


@pytest.mark.parametrize("shape", [(480, 640), (100, 150), (67, 93)], ids=str)
def test_mask_bits_from_aggregation_and_vote_without_the_f32_planes(lib, oracle, dev, shape):
    """The aggregation layer's bit words are exactly (mask != 0), and the vote that reads them instead of the f32 planes
    returns the same integers and the same centres bit for bit (incl. H W that is not a multiple of 64 or 4096)."""
    import aggregation_layer as al
    import ransac_voting_gpu_layer.ransac_voting_gpu as rvg
    from fastposecnn_amd import synth
    H, W = shape
    if shape == (480, 640):
        cat_cpu, _ = synth.make_vote_frame(0)
    else:
        cat_cpu, _ = synth.make_vote_batch(range(2), H=H, W=W, rmin=8.0, rmax=min(H, W) / 4.0, K=4)
    cat = {k: v.to(dev) for k, v in cat_cpu.items()}
    agg = al.AggregationLayer(None, 7).forward(cat)
    masks = agg["instance_masks"]
    bits = al.mask_bits_of(masks)
    assert bits is not None and masks.shape[0] >= 2
    assert np.array_equal(bits.cpu().numpy(), _pack_bits(masks.cpu().numpy(), H, W))
    vertex = agg["xy"].permute(0, 2, 3, 1).unsqueeze(3)
    hn = 256
    a, da = rvg.ransac_voting_layer_v3(masks, vertex, hn, seed=11, return_debug=True)
    b, db = rvg.ransac_voting_layer_v3(masks, vertex, hn, seed=11, return_debug=True, mask_bits=bits)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    for k in da[0]:
        assert torch.equal(da[0][k], db[0][k]), k
    # a tensor that is not the aggregation's own output carries no bits
    assert al.mask_bits_of(masks[:1]) is None and al.mask_bits_of(masks.clone()) is None
    masks.mul_(1.0)
    assert al.mask_bits_of(masks) is None            # written in place since: the words may be stale
    with pytest.raises(RuntimeError):
        rvg.ransac_voting_layer_v3(masks, vertex, hn, mask_bits=bits[:, :-1].contiguous())
