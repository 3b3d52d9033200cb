#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python.

Runs only in the build container (needs /root/reference); never on the GPU box.
What is reference code here and what is not:

* imported and executed unmodified from /root/reference/source_code/FastPoseCNN/lib:
  gpu_tensor_funcs (normalize, class_compress, samplewise_get_RT, batchwise_get_RT,
  quats_2_rotation_matrix), aggregation_layer.AggregationLayer (scipy CPU branch),
  hough_voting.HoughVotingLayer, ransac_voting_gpu.ransac_voting_layer_v3 / b_inv,
  pose_regressor.Model.class_compression; matching.batchwise_find_matches and
  gpu_tensor_funcs.batchwise_get_2d_iou.
* NOT available: the CUDA extension `ransac_voting` (no NVIDIA toolchain, no GPU).  Its two
  live kernels are stood in for by oracle/fpc_oracle.c's line-by-line restatement
  (fpco_generate_hypothesis / fpco_voting_for_hypothesis).  So the voting goldens pin the
  reference's DRIVER (compaction order, arg-max, ratio update, refinement, b_inv) around
  restated kernels; the kernels themselves are pinned only by analytic known answers
  (tests/test_oracle_kat.py).
* third-party modules absent from this image (cupy, cupyx, skimage, segmentation_models_pytorch,
  pytorch_lightning, catalyst, pytorch_toolbelt) are replaced by empty module objects: none of
  their code is on the CPU post-network path.
* torch.solve was removed from torch >= 1.13; b_inv's try-branch is restored with
  torch.linalg.solve (raises a RuntimeError subclass on singular input, as the original did).

The RANSAC pair indices (`idxs`, drawn by the reference with torch's global RNG) and the
> max_num thinning selection are RECORDED from the run and stored in the fixture so that any
implementation can be fed exactly the same samples.

Usage: python oracle/gen_golden.py   (writes tests/golden/*.npz)
"""
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/source_code/FastPoseCNN"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    """Import the reference's post-network modules on CPU. Returns a namespace."""
    import torch
    from oracle import oracle as orc

    os.environ["TOOLS_DIR"] = REF + "/tools"
    cpx = _stub("cupyx"); cpxs = _stub("cupyx.scipy"); cpxn = _stub("cupyx.scipy.ndimage")
    cpx.scipy = cpxs; cpxs.ndimage = cpxn
    _stub("cupy")
    sk = _stub("skimage"); sk.io = _stub("skimage.io")
    smp = _stub("segmentation_models_pytorch")
    pl = _stub("pytorch_lightning", LightningModule=torch.nn.Module)
    core = _stub("pytorch_lightning.core"); dec = _stub("pytorch_lightning.core.decorators", auto_move_data=lambda f: f)
    pl.core = core; core.decorators = dec
    plm = _stub("pytorch_lightning.metrics", Metric=object, functional=types.SimpleNamespace()); pl.metrics = plm
    cat = _stub("catalyst"); cc = _stub("catalyst.contrib"); ccn = _stub("catalyst.contrib.nn")
    cat.contrib = cc; cc.nn = ccn
    tb = _stub("pytorch_toolbelt"); tbl = _stub("pytorch_toolbelt.losses"); tb.losses = tbl
    del smp

    # the compiled extension: restated kernels + recording of what the driver passes in
    rec = {"calls": []}

    def generate_hypothesis(direct, coords, idxs):
        rec["calls"].append(dict(coords=coords.numpy().copy(), direct=direct.numpy().copy(),
                                 idxs=idxs.numpy().copy()))
        return torch.from_numpy(orc.generate_hypothesis(direct.numpy(), coords.numpy(), idxs.numpy()))

    def voting_for_hypothesis(direct, coords, hyp, inliers, thresh):
        assert inliers.dtype == torch.uint8 and inliers.is_contiguous()
        orc.voting_for_hypothesis(direct.numpy(), coords.numpy(), hyp.contiguous().numpy(), inliers.numpy(), thresh)

    pkg = _stub("ransac_voting_gpu_layer"); pkg.__path__ = [REF + "/lib/ransac_voting_gpu_layer"]
    ext = _stub("ransac_voting_gpu_layer.ransac_voting", generate_hypothesis=generate_hypothesis,
                voting_for_hypothesis=voting_for_hypothesis)
    pkg.ransac_voting = ext

    if not hasattr(torch, "_fpc_solve_patched"):
        torch.solve = lambda B, A: (torch.linalg.solve(A, B), None)
        torch._fpc_solve_patched = True

    sys.path.insert(0, REF + "/lib")
    import gpu_tensor_funcs as gtf
    import aggregation_layer as al
    import hough_voting as hv
    import ransac_voting_gpu_layer.ransac_voting_gpu as rvg
    try:
        import pose_regressor as pr
    except Exception as e:  # pragma: no cover - informative only
        print("pose_regressor import failed:", repr(e))
        pr = None
    return types.SimpleNamespace(torch=torch, gtf=gtf, al=al, hv=hv, rvg=rvg, pr=pr, rec=rec, orc=orc)


# ---------------------------------------------------------------------------
# synthetic scenes (numpy only; the test-suite regenerates nothing, it loads the arrays)

def radial_instance(H, W, cx, cy, rx, ry, rng, noise=0.03, outlier=0.05):
    """Elliptical instance mask and a vote field pointing at (cx, cy)."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    m = (((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2) <= 1.0
    dx, dy = cx - xx, cy - yy
    nrm = np.sqrt(dx * dx + dy * dy)
    nrm[nrm == 0] = 1
    ux, uy = dx / nrm, dy / nrm
    eps = rng.normal(0, noise, size=ux.shape).astype(np.float32)
    c, s = np.cos(eps), np.sin(eps)
    vx, vy = c * ux - s * uy, s * ux + c * uy
    o = rng.random(ux.shape) < outlier
    ang = rng.uniform(0, 2 * np.pi, size=ux.shape)
    vx = np.where(o, np.cos(ang), vx).astype(np.float32)
    vy = np.where(o, np.sin(ang), vy).astype(np.float32)
    return m, vx, vy


def _collapse_rounds(calls):
    """The reference's while-loop (RV/ransac_voting_gpu.py:557-581) calls generate_hypothesis once per
    round with the SAME idxs/coords/direct; keep one record per instance and assert the redundancy."""
    out = []
    for c in calls:
        if out and out[-1]["idxs"].shape == c["idxs"].shape and out[-1]["coords"].shape == c["coords"].shape \
                and np.array_equal(out[-1]["idxs"], c["idxs"]) and np.array_equal(out[-1]["coords"], c["coords"]):
            assert np.array_equal(out[-1]["direct"], c["direct"])
            out[-1]["rounds"] += 1
            continue
        c = dict(c); c["rounds"] = 1
        out.append(c)
    return out


def run_v3(ref, mask, vertex, hn, seed, **kw):
    """Run the reference driver, return (output, recorded idxs [n,hn,vn,2], keep planes)."""
    torch = ref.torch
    ref.rec["calls"].clear()
    torch.manual_seed(seed)
    n, H, W = mask.shape
    try:
        out = ref.rvg.ransac_voting_layer_v3(torch.from_numpy(mask), torch.from_numpy(vertex), hn, **kw)
        out = out.numpy()
    except ValueError:
        # torch >= 2 raises ValueError (not RuntimeError) for torch.cat([]); the reference's
        # intent (:602-605) is an empty (0,vn,2) result.
        out = np.zeros((0, vertex.shape[3], 2), np.float32)
    vn = vertex.shape[3]
    idxs = np.zeros((n, hn, vn, 2), np.int32)
    keep = np.ones((n, H, W), np.uint8)
    used = np.zeros(n, bool)
    min_num = kw.get("min_num", 5)
    calls = _collapse_rounds(ref.rec["calls"])
    ci = 0
    for bi in range(n):
        fg = int((mask[bi] != 0).sum())
        if fg < min_num:
            continue
        call = calls[ci]; ci += 1
        idxs[bi] = call["idxs"]
        kp = np.zeros((H, W), np.uint8)
        c = call["coords"].astype(np.int64)
        kp[c[:, 1], c[:, 0]] = 1
        keep[bi] = kp
        used[bi] = True
    assert ci == len(calls)
    return out, idxs, keep, used


def gen_vote_small(ref, rng):
    H, W, hn = 48, 64, 64
    insts = [(20.25, 18.5, 9, 11), (47.0, 30.75, 12, 8), (10.5, 40.0, 5, 4)]
    masks, vx, vy = [], [], []
    for (cx, cy, rx, ry) in insts:
        m, a, b = radial_instance(H, W, cx, cy, rx, ry, rng)
        masks.append(m.astype(np.float32)); vx.append(a * m); vy.append(b * m)
    # instance 3: four pixels only (< min_num) -> zeros row
    m = np.zeros((H, W), np.float32); m[5, 5:9] = 1
    masks.append(m); vx.append(m * 1.0); vy.append(m * 0.0)
    # instance 4: every vote parallel (1,0): all hypotheses degenerate, singular normal equations
    m = np.zeros((H, W), np.float32); m[20:30, 30:50] = 1
    masks.append(m); vx.append(m * 1.0); vy.append(m * 0.0)
    # instance 5: zero vote vectors on half of the pixels (norm1 < 1e-6 skip path)
    m, a, b = radial_instance(H, W, 32.0, 24.0, 10, 10, rng, noise=0.0, outlier=0.0)
    half = (np.arange(W)[None, :] % 2 == 0)
    masks.append(m.astype(np.float32)); vx.append(a * m * half); vy.append(b * m * half)
    # instance 6: every vote (-1,0) on rows 0..2: hypotheses degenerate to (0,0), which has inliers;
    # normal equations are rank 1 -> b_inv falls back to pinverse (RV/ransac_voting_gpu.py:513-515)
    m = np.zeros((H, W), np.float32); m[0:3, 30:50] = 1
    masks.append(m); vx.append(m * -1.0); vy.append(m * 0.0)
    mask = np.stack(masks); xy = np.stack([np.stack(vx), np.stack(vy)], axis=1)  # [n,2,H,W]
    vertex = np.ascontiguousarray(xy.transpose(0, 2, 3, 1)[:, :, :, None, :])    # [n,H,W,1,2]
    out, idxs, keep, used = run_v3(ref, mask, vertex, hn, seed=1234)
    np.savez_compressed(os.path.join(OUT, "vote_small.npz"), mask=mask, xy=xy, hn=hn, idxs=idxs,
                        expected=out, used=used, inlier_thresh=0.999, min_num=5, max_num=30000)
    print("vote_small", out.reshape(-1, 2))

    # thinning: same scene, max_num = 150 -> instances 0,1,4 are thinned with torch's uniform_()
    out, idxs, keep, used = run_v3(ref, mask, vertex, hn, seed=99, max_num=150)
    np.savez_compressed(os.path.join(OUT, "vote_thin.npz"), mask=mask, xy=xy, hn=hn, idxs=idxs, keep=keep,
                        expected=out, used=used, inlier_thresh=0.999, min_num=5, max_num=150)
    print("vote_thin", out.reshape(-1, 2), keep.reshape(len(mask), -1).sum(1))

    # empty batch
    out, _, _, _ = run_v3(ref, mask[:0], vertex[:0], hn, seed=1)
    assert out.shape == (0, 1, 2)


def gen_vote_fullres(ref, rng):
    """Two 640x480 instances, stored compactly (pixel indices + votes)."""
    H, W, hn = 480, 640, 128
    insts = [(201.3, 155.8, 60, 45), (455.6, 300.2, 38, 70)]
    masks, xys, pix, dirs = [], [], [], []
    for (cx, cy, rx, ry) in insts:
        m, a, b = radial_instance(H, W, cx, cy, rx, ry, rng)
        masks.append(m.astype(np.float32)); xys.append(np.stack([a * m, b * m]))
        p = np.flatnonzero(m.reshape(-1)).astype(np.int32)
        pix.append(p); dirs.append(np.stack([a.reshape(-1)[p], b.reshape(-1)[p]], 1).astype(np.float32))
    mask = np.stack(masks); xy = np.stack(xys)
    vertex = xy.transpose(0, 2, 3, 1)[:, :, :, None, :]          # strided view, as the reference passes it
    out, idxs, keep, used = run_v3(ref, mask, vertex, hn, seed=777)
    np.savez_compressed(os.path.join(OUT, "vote_fullres.npz"), H=H, W=W, hn=hn, idxs=idxs, expected=out,
                        pix0=pix[0], dir0=dirs[0], pix1=pix[1], dir1=dirs[1], centers=np.array(insts, np.float32))
    print("vote_fullres", out.reshape(-1, 2))


def synth_logits(B, C, H, W, rng, scenes):
    """Full-res logits whose arg-max reproduces `scenes` (list per image of (class, cx, cy, rx, ry))."""
    ml = rng.normal(0, 0.3, size=(B, C, H, W)).astype(np.float32)
    ml[:, 0] += 3.0
    q = rng.normal(0, 1, size=(B, 4 * (C - 1), H, W)).astype(np.float32)
    s = rng.normal(0, 1, size=(B, 3 * (C - 1), H, W)).astype(np.float32)
    xy = rng.normal(0, 1, size=(B, 2 * (C - 1), H, W)).astype(np.float32)
    z = rng.normal(6.5, 0.05, size=(B, (C - 1), H, W)).astype(np.float32)
    for b, scene in enumerate(scenes):
        for (cls, cx, cy, rx, ry) in scene:
            m, vx, vy = radial_instance(H, W, cx, cy, rx, ry, rng)
            ml[b, cls][m] += 6.0
            g = cls - 1
            # un-normalised votes: class_compress re-normalises them
            sc = rng.uniform(0.5, 2.0, size=vx.shape).astype(np.float32)
            xy[b, 2 * g][m] = (vx * sc)[m]; xy[b, 2 * g + 1][m] = (vy * sc)[m]
            qq = rng.normal(0, 1, 4).astype(np.float32)
            for a in range(4):
                q[b, 4 * g + a][m] = qq[a] + rng.normal(0, 0.01, int(m.sum())).astype(np.float32)
    return {"mask": ml, "quaternion": q, "scales": s, "xy": xy, "z": z}


def gen_class_compress(ref, rng):
    torch = ref.torch
    B, C, H, W = 2, 7, 24, 32
    logits = {k: v for k, v in synth_logits(B, C, H, W, rng, [[(1, 8, 8, 5, 4), (3, 22, 15, 6, 6)], [(6, 16, 12, 9, 7)]]).items()}
    # force exact ties and a -0.0 to pin tie-breaking / sign behaviour
    logits["mask"][0, :, 0, 0] = 1.25
    logits["mask"][0, 2, 0, 1] = logits["mask"][0, 0, 0, 1] = 7.5
    t = {k: torch.from_numpy(v) for k, v in logits.items()}
    if ref.pr is not None:
        self_ = types.SimpleNamespace(classes=C)
        cat = ref.pr.Model.class_compression(self_, t)
    else:
        cat_mask = torch.argmax(torch.nn.LogSoftmax(dim=1)(t["mask"]), dim=1)
        cat = ref.gtf.class_compress(C, cat_mask, t); cat["mask"] = cat_mask
    # gtf.class_compress alone with a caller-supplied mask (its public signature)
    cm2 = torch.from_numpy(rng.integers(0, C, size=(B, H, W)).astype(np.int64))
    cat2 = ref.gtf.class_compress(C, cm2, t)
    np.savez_compressed(os.path.join(OUT, "class_compress.npz"), num_classes=C,
                        **{"in_" + k: v for k, v in logits.items()},
                        **{"out_" + k: v.numpy() for k, v in cat.items()},
                        in2_mask=cm2.numpy(), **{"out2_" + k: v.numpy() for k, v in cat2.items()})
    print("class_compress classes:", np.bincount(cat["mask"].numpy().reshape(-1), minlength=C))


def blobs_cat(B, H, W, rng):
    """Categorical data with hand-placed components (touching multi-class blob, specks, empty image)."""
    cm = np.zeros((B, H, W), np.int64)
    # image 0: two separate blobs + a blob made of two touching classes + diagonal-only contact
    cm[0, 4:14, 5:20] = 2
    cm[0, 20:34, 8:18] = 5
    cm[0, 20:34, 18:30] = 3          # touches class 5 -> one component, class id = min(3,5) = 3
    cm[0, 2:4, 40:42] = 1
    cm[0, 4:6, 42:44] = 1            # diagonal neighbour of the previous speck: separate component
    cm[0, 38, 50:53] = 4             # 3-pixel speck (< min_num for voting)
    # image 1: nothing
    # image 2: U-shape whose two arms merge late in raster order + one ring
    cm[2, 5:25, 10:14] = 6; cm[2, 5:25, 30:34] = 6; cm[2, 21:25, 10:34] = 6
    cm[2, 30:38, 40:52] = 1; cm[2, 32:36, 43:49] = 0
    q = rng.normal(0, 1, (B, 4, H, W)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    s = rng.uniform(0.1, 1, (B, 3, H, W)).astype(np.float32)
    xy = rng.normal(0, 1, (B, 2, H, W)).astype(np.float32); xy /= np.linalg.norm(xy, axis=1, keepdims=True)
    z = rng.normal(6.5, 0.2, (B, H, W)).astype(np.float32)
    fg = (cm != 0)
    q *= fg[:, None]; s *= fg[:, None]; xy *= fg[:, None]; z *= fg
    return {"mask": cm, "quaternion": q, "scales": s, "xy": xy, "z": z}


def gen_aggregate(ref, rng):
    torch = ref.torch
    B, H, W = 3, 40, 56
    cat = blobs_cat(B, H, W, rng)
    layer = ref.al.AggregationLayer(types.SimpleNamespace(HV_NUM_OF_HYPOTHESES=32), 7)
    t = {k: torch.from_numpy(v) for k, v in cat.items()}
    agg = layer.forward(t)
    labels, N = layer.batchwise_break_segmentation_mask(t["mask"] != 0)
    np.savez_compressed(os.path.join(OUT, "aggregate.npz"), **{"in_" + k: v for k, v in cat.items()},
                        labels=labels.numpy().astype(np.int32), N=N,
                        **{"out_" + k: v.numpy() for k, v in agg.items()})
    print("aggregate N =", N, "classes", agg["class_ids"].numpy(), "samples", agg["sample_ids"].numpy())

    # zero instances in the whole batch
    cat0 = {k: np.zeros_like(v) for k, v in cat.items()}
    agg0 = layer.forward({k: torch.from_numpy(v) for k, v in cat0.items()})
    np.savez_compressed(os.path.join(OUT, "aggregate_empty.npz"),
                        **{"out_" + k: v.numpy() for k, v in agg0.items()},
                        **{"dtype_" + k: str(v.dtype) for k, v in agg0.items()})
    print("aggregate_empty", {k: tuple(v.shape) for k, v in agg0.items()})


def gen_pose_rt(ref, rng):
    torch = ref.torch
    n = 9
    q = rng.normal(0, 1, (n, 4)).astype(np.float32)
    q[0] = [1, 0, 0, 0]; q[1] = [0, 0, 0, 1]
    q[2:] /= np.linalg.norm(q[2:], axis=1, keepdims=True)
    q[5] *= 3.0                                         # not unit: batchwise_get_RT renormalises
    xy = np.stack([rng.uniform(0, 640, n), rng.uniform(0, 480, n)], 1).astype(np.float32)
    z = np.exp(rng.normal(6.7, 0.3, (n, 1))).astype(np.float32)
    K = np.array([[577.5, 0, 319.5], [0., 577.5, 239.5], [0., 0., 1.]])      # F/tools/project.py:78
    Kinv = torch.inverse(torch.from_numpy(K).float())
    agg = {"quaternion": torch.from_numpy(q), "xy": torch.from_numpy(xy), "z": torch.from_numpy(z)}
    out = ref.gtf.samplewise_get_RT(agg, Kinv)
    np.savez_compressed(os.path.join(OUT, "pose_rt.npz"), q=q, xy=xy, z=z, K=K.astype(np.float32), Kinv=Kinv.numpy(),
                        R=out["R"].numpy(), T=out["T"].numpy(), RT=out["RT"].numpy())
    print("pose_rt q=(1,0,0,0) -> R diag", np.diag(out["R"].numpy()[0]))


def gen_pipeline(ref, rng):
    """logits -> class compression -> aggregation -> hough voting -> RT, B=2, 48x64."""
    torch = ref.torch
    B, C, H, W, hn = 2, 7, 48, 64, 96
    scenes = [[(2, 15.3, 14.2, 9, 8), (5, 45.8, 30.1, 11, 10)], [(1, 30.5, 22.25, 14, 12), (4, 8.0, 40.0, 4, 4), (6, 55, 8, 5, 5)]]
    logits = synth_logits(B, C, H, W, rng, scenes)
    t = {k: torch.from_numpy(v) for k, v in logits.items()}
    hp = types.SimpleNamespace(HV_NUM_OF_HYPOTHESES=hn)
    self_ = types.SimpleNamespace(classes=C)
    if ref.pr is not None:
        cat = ref.pr.Model.class_compression(self_, t)
    else:
        cm = torch.argmax(torch.nn.LogSoftmax(dim=1)(t["mask"]), dim=1)
        cat = ref.gtf.class_compress(C, cm, t); cat["mask"] = cm
    agg = ref.al.AggregationLayer(hp, C).forward(cat)
    n = agg["instance_masks"].shape[0]
    ref.rec["calls"].clear()
    torch.manual_seed(4242)
    masks_np = agg["instance_masks"].numpy().copy()
    agg = ref.hv.HoughVotingLayer(hp).forward(agg)
    idxs = np.zeros((n, hn, 1, 2), np.int32)
    calls = _collapse_rounds(ref.rec["calls"])
    ci = 0
    for bi in range(n):
        if (masks_np[bi] != 0).sum() < 5:
            continue
        idxs[bi] = calls[ci]["idxs"]; ci += 1
    assert ci == len(calls)
    K = np.array([[577.5, 0, 319.5], [0., 577.5, 239.5], [0., 0., 1.]])
    Kinv = torch.inverse(torch.from_numpy(K).float())
    agg = ref.gtf.samplewise_get_RT(agg, Kinv)
    np.savez_compressed(os.path.join(OUT, "pipeline.npz"), num_classes=C, hn=hn, idxs=idxs, Kinv=Kinv.numpy(),
                        **{"logits_" + k: v for k, v in logits.items()},
                        **{"cat_" + k: v.numpy() for k, v in cat.items()},
                        **{"agg_" + k: v.numpy() for k, v in agg.items()})
    print("pipeline n =", n, "xy", agg["xy"].numpy(), "classes", agg["class_ids"].numpy())


def gen_matching(ref, rng):
    """matching.batchwise_find_matches + gpu_tensor_funcs.batchwise_get_2d_iou on two AggData dicts:
    ground truth (7 instances, 3 classes, 2 samples) against predictions (8 instances; perturbed masks, one of a
    class the ground truth lacks, one empty mask, one ground-truth instance left without any overlapping
    prediction).  A second case pins the 0/0 = NaN row (empty gt mask vs empty prediction)."""
    torch = ref.torch
    import matching as mg
    H, W = 40, 56
    yy, xx = np.mgrid[0:H, 0:W]

    def disc(cx, cy, r):
        return (((xx - cx) ** 2 + (yy - cy) ** 2) <= r * r).astype(np.float32)

    def agg(masks, cls, sid, seed):
        r = np.random.default_rng(seed)
        n = len(masks)
        q = r.normal(size=(n, 4)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
        d = {"class_ids": np.asarray(cls, np.int64), "sample_ids": np.asarray(sid, np.int64),
             "instance_masks": np.stack(masks).astype(np.float32), "quaternion": q,
             "scales": r.uniform(0.1, 1, (n, 3)).astype(np.float32), "xy": r.uniform(0, 50, (n, 2)).astype(np.float32),
             "z": r.uniform(500, 1500, (n, 1)).astype(np.float32), "R": r.normal(size=(n, 3, 3)).astype(np.float32),
             "T": r.normal(size=(n, 3)).astype(np.float32), "RT": r.normal(size=(n, 4, 4)).astype(np.float32)}
        return d

    g_masks = [disc(10, 10, 6), disc(30, 12, 7), disc(45, 30, 6), disc(12, 30, 5), disc(28, 28, 6), disc(48, 8, 4), disc(20, 20, 3)]
    gts = agg(g_masks, [1, 1, 2, 2, 3, 3, 1], [0, 0, 0, 1, 1, 1, 1], 1)
    gts["symmetric_ids"] = np.asarray([0, 1, 0, 2, 0, 1, 0], np.int64)
    p_masks = [disc(11, 10, 6), disc(31, 13, 6), disc(44, 31, 7), disc(29, 27, 6), disc(13, 31, 5), disc(5, 35, 3),
               np.zeros((H, W), np.float32), disc(10, 11, 5)]
    preds = agg(p_masks, [1, 1, 2, 3, 2, 4, 3, 1], [0, 0, 0, 1, 1, 1, 1, 0], 2)
    t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    out = mg.batchwise_find_matches(t(preds), t(gts))
    iou_all = ref.gtf.batchwise_get_2d_iou(torch.from_numpy(gts["instance_masks"]), torch.from_numpy(preds["instance_masks"]))
    # NaN case: an empty ground-truth mask and an empty prediction of the same class
    g2 = agg([disc(10, 10, 6), np.zeros((H, W), np.float32), disc(40, 20, 5)], [1, 1, 2], [0, 0, 0], 3)
    g2["symmetric_ids"] = np.asarray([0, 0, 1], np.int64)
    p2 = agg([np.zeros((H, W), np.float32), disc(10, 11, 6), disc(41, 20, 5)], [1, 1, 2], [0, 0, 0], 4)
    out2 = mg.batchwise_find_matches(t(p2), t(g2))
    iou2 = ref.gtf.batchwise_get_2d_iou(torch.from_numpy(g2["instance_masks"]), torch.from_numpy(p2["instance_masks"]))
    # no prediction at all -> None; predictions of foreign classes only -> None
    none1 = mg.batchwise_find_matches({k: v[:0] for k, v in t(preds).items()}, t(gts))
    p3 = {k: v[5:6] for k, v in t(preds).items()}
    none2 = mg.batchwise_find_matches(p3, t(gts))
    assert none1 is None and none2 is None
    sv = {}
    for tag, d in (("gts", gts), ("preds", preds), ("gts2", g2), ("preds2", p2)):
        sv.update({f"{tag}_{k}": v for k, v in d.items()})
    sv.update({f"out_{k}": v.numpy() for k, v in out.items()})
    sv.update({f"out2_{k}": v.numpy() for k, v in out2.items()})
    np.savez_compressed(os.path.join(OUT, "matching.npz"), iou_all=iou_all.numpy(), iou2=iou2.numpy(), **sv)
    print("matching:", {k: tuple(v.shape) for k, v in out.items()}, "| nan case rows:", out2["class_ids"].tolist())


def gen_eval_and_losses(ref):
    """SURVEY 8f rank 2 + the matched losses of rank 4, from the reference's own functions on one synthetic set of matched
    pairs: gpu_tensor_funcs.get_quat_distance (raw / symmetric / mixed), get_3d_ious, from_Ts_get_offset_error,
    from_RTs_get_T_offset_errors, calculate_aps, calculate_complex_aps; loss.QLoss / XYLoss / ZLoss / ScalesLoss / RLoss /
    TLoss / Iou3dLoss / OffsetLoss / CE / CCE with their input gradients.  loss.Focal needs pytorch_toolbelt (absent) and
    is not pinned.  Own RNG: the older fixtures do not change when this one is added."""
    torch = ref.torch
    gtf = ref.gtf
    rng = np.random.default_rng(20261004)
    _stub("matplotlib"); sys.modules["matplotlib"].pyplot = _stub("matplotlib.pyplot")
    foc = _stub("pytorch_toolbelt.losses.focal", FocalLoss=object)
    sys.modules["pytorch_toolbelt.losses"].focal = foc
    import loss as ref_loss
    n = 11
    f32 = lambda a: np.asarray(a, np.float32)

    def poses(seed):
        r = np.random.default_rng(seed)
        q = r.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
        xy = r.uniform(50, 590, (n, 2)); z = r.uniform(600, 1500, (n, 1)); sc = r.uniform(0.08, 0.6, (n, 3))
        return f32(q), f32(xy), f32(z), f32(sc)

    q0, xy0, z0, s0 = poses(1)
    dq, dxy, dz, ds = poses(2)
    q1 = q0 + 0.25 * dq; q1 /= np.linalg.norm(q1, axis=1, keepdims=True)
    q1[3] = -q0[3]                                   # antipodal pair: zero distance through the |q0 + q1| branch
    xy1 = xy0 + f32(rng.normal(0, 6, (n, 2))); z1 = z0 * f32(rng.uniform(0.9, 1.1, (n, 1))); s1 = s0 * f32(rng.uniform(0.8, 1.25, (n, 3)))
    K = np.array([[577.5, 0, 319.5], [0., 577.5, 239.5], [0., 0., 1.]])
    inv_k = torch.inverse(torch.from_numpy(K).float())
    t = torch.from_numpy
    R0, T0, RT0 = gtf.batchwise_get_RT(t(q0), t(xy0), t(z0), inv_k)
    R1, T1, RT1 = gtf.batchwise_get_RT(t(f32(q1)), t(xy1), t(z1), inv_k)
    sym = np.asarray([0, 1, 0, 2, 0, 1, 0, 0, 1, 0, 2], np.int64)
    cls = np.asarray([1, 1, 2, 2, 3, 3, 1, 4, 4, 5, 6], np.int64)
    out = {"q0": q0, "q1": f32(q1), "xy0": xy0, "xy1": xy1, "z0": z0, "z1": z1, "s0": s0, "s1": s1, "sym": sym, "cls": cls,
           "R0": R0.numpy(), "T0": T0.numpy(), "RT0": RT0.numpy(), "R1": R1.numpy(), "T1": T1.numpy(), "RT1": RT1.numpy()}
    tq0, tq1 = t(q0), t(f32(q1))
    out["deg_mixed"] = gtf.get_quat_distance(tq0, tq1, t(sym)).numpy()
    out["deg_raw"] = gtf.get_quat_distance(tq0, tq1).numpy()
    out["deg_sym"] = gtf.get_symmetric_quat_distance(tq0, tq1).numpy()
    out["deg_all_sym"] = gtf.get_quat_distance(tq0, tq1, torch.ones(n, dtype=torch.int64)).numpy()
    out["deg_none_sym"] = gtf.get_quat_distance(tq0, tq1, torch.zeros(n, dtype=torch.int64)).numpy()
    out["iou3d"] = gtf.get_3d_ious(RT0, RT1, t(s0), t(s1)).numpy()
    out["iou3d_self"] = gtf.get_3d_ious(RT0, RT0, t(s0), t(s0)).numpy()
    out["offset"] = gtf.from_Ts_get_offset_error(T0, T1).numpy()
    out["offset_rt"] = gtf.from_RTs_get_T_offset_errors(RT0, RT1).numpy()
    rot, _ = gtf.quat_symmetric_tf(tq1[:2], tq0[:2])
    out["sym_tf_first2"] = rot.numpy()
    # calculate_aps / calculate_complex_aps as evaluate.py:205-330 drives them
    raw = {"degree_error": {}, "3d_iou": {}, "offset_error": {}}
    deg_pp = torch.from_numpy(np.where(sym == 0, out["deg_raw"], out["deg_sym"]))
    for c in np.unique(cls):
        i = torch.from_numpy(np.where(cls == c)[0])
        raw["degree_error"][int(c)] = deg_pp[i]
        raw["3d_iou"][int(c)] = torch.from_numpy(out["iou3d"])[i] * 100
        raw["offset_error"][int(c)] = torch.from_numpy(out["offset"])[i]
    raw["degree_error"][1][0] = float("nan")        # NaNs are dropped per class
    thr = {"degree_error": torch.tensor([5., 10., 30., 60.]), "3d_iou": torch.tensor([1., 10., 25., 50.]),
           "offset_error": torch.tensor([5., 10., 50., 200.])}
    ops = {"degree_error": torch.less, "3d_iou": torch.greater, "offset_error": torch.less}
    aps = gtf.calculate_aps(raw, thr, ops)
    for k in aps:
        for c, v in aps[k].items():
            out[f"aps_{k}_{c}"] = v.numpy()
    raw2 = {k: {c: torch.nan_to_num(v.double(), nan=1e9) for c, v in d.items()} for k, d in raw.items()}
    cthr = {"degree_error+offset_error": torch.vstack((torch.tensor([5, 10, 60]), torch.tensor([5, 50, 200])))}
    caps = gtf.calculate_complex_aps(raw2, cthr, ops)
    for c, v in caps["degree_error+offset_error"].items():
        out[f"caps_{c}"] = v.numpy()
    for k, d in raw.items():
        for c, v in d.items():
            out[f"raw_{k}_{c}"] = v.numpy()

    # matched losses with gradients w.r.t. the prediction
    def matched(pred_req):
        m = {"instance_masks": torch.zeros((2, n, 2, 2)), "symmetric_ids": t(sym), "class_ids": t(cls)}
        leaves = {}
        for key, a, b in (("quaternion", q0, f32(q1)), ("xy", xy0, xy1), ("z", z0, z1), ("scales", s0, s1),
                          ("R", R0.numpy(), R1.numpy()), ("T", T0.numpy(), T1.numpy()), ("RT", RT0.numpy(), RT1.numpy())):
            p = t(np.ascontiguousarray(b)).clone().requires_grad_(pred_req)
            leaves[key] = p
            m[key] = torch.stack((t(np.ascontiguousarray(a)), p))
        return m, leaves

    for name, fn in (("QLoss", ref_loss.QLoss(key="quaternion")), ("XYLoss", ref_loss.XYLoss(key="xy")),
                     ("ZLoss", ref_loss.ZLoss(key="z")), ("ScalesLoss", ref_loss.ScalesLoss(key="scales")),
                     ("RLoss", ref_loss.RLoss(key="R")), ("TLoss", ref_loss.TLoss(key="T")),
                     ("Iou3dLoss", ref_loss.Iou3dLoss()), ("OffsetLoss", ref_loss.OffsetLoss())):
        m, leaves = matched(True)
        val = fn(m)
        out[f"loss_{name}"] = val.detach().numpy()
        if val.requires_grad:
            val.backward()
            key = {"QLoss": "quaternion", "XYLoss": "xy", "ZLoss": "z", "ScalesLoss": "scales", "RLoss": "R", "TLoss": "T",
                   "Iou3dLoss": "RT", "OffsetLoss": "RT"}[name]
            out[f"grad_{name}"] = leaves[key].grad.numpy()
        out[f"loss_{name}_none"] = fn(None).cpu().numpy()
    # pixel-wise mask losses
    B, C, H, W = 2, 7, 12, 16
    ml = t(f32(rng.normal(0, 2, (B, C, H, W)))).requires_grad_(True)
    gt_mask = t(rng.integers(0, C, (B, H, W)).astype(np.int64))
    out["mask_logits"] = ml.detach().numpy(); out["gt_mask"] = gt_mask.numpy()
    for name, fn in (("CE", ref_loss.CE()), ("CCE", ref_loss.CCE())):
        ml.grad = None
        val = fn({"logits": {"mask": ml}}, {"mask": gt_mask})
        val.backward()
        out[f"loss_{name}"] = val.detach().numpy(); out[f"grad_{name}"] = ml.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "eval_losses.npz"), **out)
    print("eval_losses:", {k: (v.shape, str(v.dtype)) for k, v in out.items() if k.startswith(("deg_", "iou", "off", "loss_"))})


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    if "--only-eval" in sys.argv:
        gen_eval_and_losses(ref)
        return
    rng = np.random.default_rng(20261003)
    gen_vote_small(ref, rng)
    gen_vote_fullres(ref, rng)
    gen_class_compress(ref, rng)
    gen_aggregate(ref, rng)
    gen_pose_rt(ref, rng)
    gen_pipeline(ref, rng)
    gen_matching(ref, rng)
    gen_eval_and_losses(ref)
    import torch, scipy
    with open(os.path.join(OUT, "PROVENANCE.txt"), "w") as f:
        f.write("generated by oracle/gen_golden.py from /root/reference (FastPoseCNN @ v0)\n"
                f"torch {torch.__version__}, numpy {np.__version__}, scipy {scipy.__version__}\n"
                "CUDA extension kernels stood in by oracle/fpc_oracle.c (see gen_golden.py docstring)\n")


if __name__ == "__main__":
    main()
