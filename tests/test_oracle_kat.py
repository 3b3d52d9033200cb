"""Analytic known answers for the two restated kernels (the reference ships no test for them;
the pattern is the radial-field idea of F/lib/hough_voting.py:583-618)."""
import numpy as np


def _radial(cx, cy, pts):
    d = np.array([cx, cy], np.float32)[None] - pts
    return (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)


def test_generate_hypothesis_intersection_and_degenerate(oracle):
    # exactly representable geometry: votes along the axes meet at (4, 3)
    coords = np.array([[0, 3], [4, 0], [8, 3], [1, 1]], np.float32)
    direct = np.array([[1, 0], [0, 1], [-1, 0], [1, 0]], np.float32)[:, None, :]
    idxs = np.array([[0, 1], [1, 2], [0, 2], [0, 3], [1, 1]], np.int32)[:, None, :]
    hyp = oracle.generate_hypothesis(direct, coords, idxs)
    assert hyp.shape == (5, 1, 2)
    assert np.array_equal(hyp[0, 0], [4, 3]) and np.array_equal(hyp[1, 0], [4, 3])
    # anti-parallel, parallel and identical pairs: |det| < 1e-6 -> left at the zero initialisation
    assert np.array_equal(hyp[2:, 0], np.zeros((3, 2)))


def test_vote_strict_threshold_sign_and_skips(oracle):
    coords = np.array([[0, 0], [10, 0], [5, 5], [3, 0], [4, 0]], np.float32)
    direct = np.array([[1, 0], [1, 0], [0, 0], [1, 0], [0.6, 0.8]], np.float32)[:, None, :]
    hyp = np.array([[[5, 0]], [[3, 0]]], np.float32)
    inl = np.zeros((2, 1, 5), np.uint8)
    oracle.voting_for_hypothesis(direct, coords, hyp, inl, 0.999)
    # h=(5,0): pixel 0 points at it; pixel 1 points AWAY (cos = -1, signed test); pixel 2 has a zero
    # vote (norm1 < 1e-6 skip); pixel 3 points at it; pixel 4: cos = 0.6
    assert inl[0, 0].tolist() == [1, 0, 0, 1, 0]
    # h=(3,0): pixel 3 sits ON the hypothesis (norm2 < 1e-6 skip) -> not an inlier
    assert inl[1, 0].tolist() == [1, 0, 0, 0, 0]
    # only ever writes 1: pre-set entries survive
    inl[:] = 7
    oracle.voting_for_hypothesis(direct, coords, hyp, inl, 0.999)
    assert set(np.unique(inl).tolist()) == {1, 7}
    # strict '>' : cos == thresh exactly is NOT an inlier
    one = np.zeros((1, 1, 1), np.uint8)
    oracle.voting_for_hypothesis(np.array([[[1, 0]]], np.float32), np.array([[0, 0]], np.float32),
                                 np.array([[[2, 0]]], np.float32), one, 1.0)
    assert one[0, 0, 0] == 0


def test_radial_field_recovers_exact_centre(oracle):
    # SURVEY.md section 4 [probe]: a perfect radial field returns the centre exactly
    H, W = 48, 64
    cx, cy = 40.25, 27.5
    mask = np.zeros((1, H, W), np.float32); mask[0, 18:38, 30:50] = 1
    yy, xx = np.mgrid[0:H, 0:W]
    pts = np.stack([xx.reshape(-1), yy.reshape(-1)], 1).astype(np.float32)
    v = _radial(cx, cy, pts).reshape(H, W, 2) * mask[0][:, :, None]
    out, dbg = oracle.ransac_voting_layer_v3(mask, v[None, :, :, None, :], 128, seed=3, return_debug=True)
    np.testing.assert_allclose(out[0, 0], [cx, cy], atol=1e-4)
    assert dbg[0]["tn"][0] == 400 and dbg[0]["win_count"][0] >= 395
    # same seed -> same samples -> same answer; different seed -> still the centre
    out2 = oracle.ransac_voting_layer_v3(mask, v[None, :, :, None, :], 128, seed=3)
    assert np.array_equal(out, out2)
    out3 = oracle.ransac_voting_layer_v3(mask, v[None, :, :, None, :], 128, seed=4)
    np.testing.assert_allclose(out3[0, 0], [cx, cy], atol=1e-4)


def test_first_maximal_hypothesis_wins(oracle):
    # two identical hypotheses (same pair twice): torch.max returns the first maximal index
    H, W = 16, 16
    mask = np.zeros((1, H, W), np.float32); mask[0, 4:12, 4:12] = 1
    yy, xx = np.mgrid[0:H, 0:W]
    pts = np.stack([xx.reshape(-1), yy.reshape(-1)], 1).astype(np.float32)
    v = _radial(8.5, 7.5, pts).reshape(H, W, 2) * mask[0][:, :, None]
    idxs = np.array([[0, 0], [3, 60], [3, 60], [5, 40]], np.int32).reshape(1, 4, 1, 2)
    _, dbg = oracle.ransac_voting_layer_v3(mask, v[None, :, :, None, :], 4, idxs=idxs, return_debug=True)
    c = dbg[0]["counts"][0]
    assert c[1] == c[2] == c.max() and dbg[0]["win_idx"][0] == int(np.argmax(c))


def test_cc_label_matches_scipy(oracle):
    import scipy.ndimage
    rng = np.random.default_rng(5)
    fg = rng.random((3, 37, 53)) < 0.55
    s = np.zeros((3, 3, 3), bool); s[1, 1, :] = True; s[1, :, 1] = True     # aggregation_layer.py:43-59
    want, n = scipy.ndimage.label(fg, structure=s)
    got, m = oracle.cc_label(fg)
    assert m == n and np.array_equal(got, want.astype(np.int32))


def _kernel_kat():
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kernel_kat.json")) as f:
        return json.load(f)


def test_hand_derived_kernel_vectors(oracle):
    """tests/golden/kernel_kat.json: every skip branch of .cu:22-48 / :100-125 with the expected output derived by
    hand (exact fp32 arithmetic on powers of two), incl. the float32(1e-6)-vs-double-literal case."""
    kat = _kernel_kat()
    for c in kat["generate_hypothesis"]:
        direct = np.asarray(c["direct"], np.float32)[:, None, :]
        coords = np.asarray(c["coords"], np.float32)
        idxs = np.asarray([c["pair"]], np.int32)[:, None, :]
        hyp = oracle.generate_hypothesis(direct, coords, idxs)
        assert np.array_equal(hyp[0, 0], np.asarray(c["expect"], np.float32)), (c["name"], hyp[0, 0])
    for c in kat["voting_for_hypothesis"]:
        inl = np.zeros((1, 1, 1), np.uint8)
        oracle.voting_for_hypothesis(np.asarray([[c["vote"]]], np.float32), np.asarray([c["c"]], np.float32),
                                     np.asarray([[c["h"]]], np.float32), inl, np.float32(c["thresh"]))
        assert int(inl[0, 0, 0]) == c["inlier"], c["name"]
