"""The CPU oracle (oracle/fpc_oracle.c) against golden vectors produced by the reference's own
Python (oracle/gen_golden.py).  Tolerances: index / id / mask outputs bit-exact; floating point
within 1e-4 as BASELINE.json's north_star states (tighter where the arithmetic allows)."""
import numpy as np
import pytest

from conftest import assert_pose, assert_rel, load_golden


def test_vote_small_matches_reference_driver(oracle):
    g = load_golden("vote_small.npz")
    vertex = g["xy"].transpose(0, 2, 3, 1)[:, :, :, None, :]       # strided view like hough_voting.py:51
    out, dbg = oracle.ransac_voting_layer_v3(g["mask"], vertex, int(g["hn"]), idxs=g["idxs"], return_debug=True)
    assert out.shape == g["expected"].shape == (7, 1, 2)
    # fp32 torch sums in the reference vs fp64 sums in the oracle: measured 4e-7 relative; the bar is north_star's 1e-4 per element
    assert_rel(out, g["expected"], what="centre vs reference driver")
    assert np.array_equal(out[3], np.zeros((1, 2)))                # < min_num pixels -> zeros
    assert np.array_equal(out[4], np.zeros((1, 2)))                # parallel votes, no inlier -> pinv(0) = 0
    assert dbg[0]["win_idx"][4] == -1 and dbg[0]["inlier_count"][4] == 0
    assert out[6, 0, 0] == 0 and abs(out[6, 0, 1] - 2.0 / 3.0) < 1e-6   # rank-1 normal equations -> pinverse


def test_vote_thinning_with_recorded_selection(oracle):
    g = load_golden("vote_thin.npz")
    vertex = g["xy"].transpose(0, 2, 3, 1)[:, :, :, None, :]
    out, dbg = oracle.ransac_voting_layer_v3(g["mask"], vertex, int(g["hn"]), idxs=g["idxs"], keep=g["keep"],
                                             max_num=int(g["max_num"]), return_debug=True)
    assert_rel(out, g["expected"], what="centre vs reference driver")
    fg = (g["mask"] != 0).reshape(len(out), -1).sum(1)
    kept = (g["keep"] * (g["mask"] != 0)).reshape(len(out), -1).sum(1)
    for i in range(len(out)):
        if fg[i] > int(g["max_num"]):
            assert dbg[0]["tn"][i] == kept[i] < fg[i]
        elif fg[i] >= 5:
            assert dbg[0]["tn"][i] == fg[i]                        # selection is ignored below max_num


def test_vote_fullres(oracle):
    g = load_golden("vote_fullres.npz")
    H, W = int(g["H"]), int(g["W"])
    mask = np.zeros((2, H * W), np.float32); xy = np.zeros((2, 2, H * W), np.float32)
    for i in range(2):
        mask[i, g[f"pix{i}"]] = 1
        xy[i, 0, g[f"pix{i}"]] = g[f"dir{i}"][:, 0]; xy[i, 1, g[f"pix{i}"]] = g[f"dir{i}"][:, 1]
    vertex = xy.reshape(2, 2, H, W).transpose(0, 2, 3, 1)[:, :, :, None, :]
    out = oracle.ransac_voting_layer_v3(mask.reshape(2, H, W), vertex, int(g["hn"]), idxs=g["idxs"])
    assert_rel(out, g["expected"], what="centre vs reference driver")
    np.testing.assert_allclose(out[:, 0], g["centers"][:, :2], atol=0.5)     # and it is the true centre


def test_vote_empty_batch(oracle):
    out = oracle.ransac_voting_layer_v3(np.zeros((0, 8, 8), np.float32), np.zeros((0, 8, 8, 1, 2), np.float32), 16)
    assert out.shape == (0, 1, 2)


def test_class_compress(oracle):
    g = load_golden("class_compress.npz")
    logits = {k[3:]: v for k, v in g.items() if k.startswith("in_")}
    cat = oracle.class_compress(logits, int(g["num_classes"]))
    assert np.array_equal(cat["mask"], g["out_mask"])              # bit-exact ids, ties -> first index
    assert cat["mask"][0, 0, 0] == 0 and cat["mask"][0, 0, 1] == 0
    for k in ("quaternion", "scales", "xy", "z"):
        assert cat[k].shape == g["out_" + k].shape
        np.testing.assert_allclose(cat[k], g["out_" + k], atol=1e-6, rtol=1e-6)
    # selection is exact; only the normalisation has rounding freedom
    assert np.array_equal(cat["scales"], g["out_scales"]) and np.array_equal(cat["z"], g["out_z"])
    # gtf.class_compress with a caller-supplied mask
    cat2 = oracle.class_compress(logits, int(g["num_classes"]), cat_mask=g["in2_mask"])
    for k in ("quaternion", "scales", "xy", "z"):
        np.testing.assert_allclose(cat2[k], g["out2_" + k], atol=1e-6, rtol=1e-6)


def test_cc_label_and_aggregate(oracle):
    g = load_golden("aggregate.npz")
    cat = {k[3:]: v for k, v in g.items() if k.startswith("in_")}
    labels, N = oracle.cc_label(cat["mask"] != 0)
    assert N == int(g["N"]) == 7
    assert np.array_equal(labels, g["labels"])                     # scipy raster order, global numbering
    agg = oracle.aggregate(cat)
    assert np.array_equal(agg["class_ids"], g["out_class_ids"].astype(np.int64))
    assert np.array_equal(agg["sample_ids"], g["out_sample_ids"])
    assert np.array_equal(agg["instance_masks"], g["out_instance_masks"])
    assert np.array_equal(agg["xy"], g["out_xy"])
    for k in ("quaternion", "scales", "z"):
        assert agg[k].shape == g["out_" + k].shape
        np.testing.assert_allclose(agg[k], g["out_" + k], atol=1e-5, rtol=1e-5)


def test_aggregate_empty(oracle):
    g = load_golden("aggregate_empty.npz")
    cat = {"mask": np.zeros((3, 40, 56), np.int64), "quaternion": np.zeros((3, 4, 40, 56), np.float32),
           "scales": np.zeros((3, 3, 40, 56), np.float32), "xy": np.zeros((3, 2, 40, 56), np.float32),
           "z": np.zeros((3, 40, 56), np.float32)}
    agg = oracle.aggregate(cat)
    for k in ("class_ids", "sample_ids", "instance_masks", "quaternion", "scales", "xy", "z"):
        assert agg[k].shape == g["out_" + k].shape, k


def test_pose_rt(oracle):
    g = load_golden("pose_rt.npz")
    R, T, RT = oracle.pose_rt(g["q"], g["xy"], g["z"], g["Kinv"])
    assert_pose(R, T, RT, g["R"], g["T"], g["RT"], what="pose_rt golden")
    np.testing.assert_allclose(R, g["R"], atol=1e-5)
    np.testing.assert_allclose(np.diag(R[0]), [1, -1, -1], atol=1e-7)        # scalar-last, transposed


def test_pipeline_chain(oracle):
    g = load_golden("pipeline.npz")
    logits = {k[7:]: v for k, v in g.items() if k.startswith("logits_")}
    cat = oracle.class_compress(logits, int(g["num_classes"]))
    assert np.array_equal(cat["mask"], g["cat_mask"])
    agg = oracle.aggregate(cat)
    assert np.array_equal(agg["class_ids"], g["agg_class_ids"].astype(np.int64))
    assert np.array_equal(agg["sample_ids"], g["agg_sample_ids"])
    assert np.array_equal(agg["instance_masks"], g["agg_instance_masks"])
    np.testing.assert_allclose(agg["xy"], g["agg_xy_mask"], atol=1e-6)
    vertex = agg["xy"].transpose(0, 2, 3, 1)[:, :, :, None, :]
    xy = oracle.ransac_voting_layer_v3(agg["instance_masks"], vertex, int(g["hn"]), idxs=g["idxs"])
    assert_rel(xy, g["agg_hypothesis"], what="pipeline centre")
    assert_rel(xy[:, 0], g["agg_xy"], what="pipeline xy")
    R, T, RT = oracle.pose_rt(agg["quaternion"], xy[:, 0], agg["z"], g["Kinv"])
    assert_pose(R, T, RT, g["agg_R"], g["agg_T"], g["agg_RT"], what="pipeline")
    np.testing.assert_allclose(agg["quaternion"], g["agg_quaternion"], atol=1e-4, rtol=0)
    assert_rel(agg["scales"], g["agg_scales"], what="scales")
    assert_rel(agg["z"], g["agg_z"], what="z")


def _dicts(g, tag):
    return {k[len(tag) + 1:]: g[k] for k in list(g.keys()) if k.startswith(tag + "_")}


def test_matching_iou_and_find_matches(oracle):
    """gtf.batchwise_get_2d_iou and mg.batchwise_find_matches of the reference (golden) vs the
    oracle's restatement: IoU matrix bit for bit (integer counts, one f32 division, NaN for 0/0), matches identical."""
    g = load_golden("matching.npz")
    gts, preds, g2, p2 = (_dicts(g, t) for t in ("gts", "preds", "gts2", "preds2"))
    iou, inter, uni = oracle.mask_iou(gts["instance_masks"], preds["instance_masks"], return_counts=True)
    assert np.array_equal(iou, g["iou_all"], equal_nan=True)
    assert (inter <= uni).all() and inter.dtype == np.int64
    assert np.array_equal(oracle.mask_iou(g2["instance_masks"], p2["instance_masks"]), g["iou2"], equal_nan=True)
    assert np.isnan(g["iou2"][1, 0])                                   # empty ground truth vs empty prediction
    for (p, t, tag) in ((preds, gts, "out"), (p2, g2, "out2")):
        want = _dicts(g, tag)
        got = oracle.find_matches(p, t)
        assert sorted(got) == sorted(want)
        for k in want:
            assert np.array_equal(got[k], want[k]), k
    assert oracle.find_matches({k: v[:0] for k, v in preds.items()}, gts) is None       # no prediction
    assert oracle.find_matches({k: v[5:6] for k, v in preds.items()}, gts) is None      # foreign class only
    assert oracle.find_matches(None, gts) is None and oracle.find_matches(preds, {}) is None
