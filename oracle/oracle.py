"""numpy/ctypes front end of the CPU oracle (oracle/fpc_oracle.c).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under fastposecnn_amd/ may import this.

Every function mirrors one reference entry point (citations in fpc_oracle.c):
    generate_hypothesis / voting_for_hypothesis  RV/src/ransac_voting_kernel.cu:11-126
    ransac_voting_layer_v3                        RV/ransac_voting_gpu.py:518-607
    class_compress                                F/lib/pose_regressor.py:445-457, F/lib/gpu_tensor_funcs.py:52-99
    cc_label                                      F/lib/aggregation_layer.py:160-183
    aggregate                                     F/lib/aggregation_layer.py:61-158
    pose_rt                                       F/lib/gpu_tensor_funcs.py:204-253,306-326
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfpc_oracle.so")
_lib = None

c_f32p = ctypes.POINTER(ctypes.c_float)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile libfpc_oracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "fpc_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "fpc_rng.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libfpc_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()          # no-op when the library is newer than its sources
        _lib = ctypes.CDLL(_LIB_PATH)
        for name in ("fpco_generate_hypothesis", "fpco_voting_for_hypothesis", "fpco_ransac_voting_v3",
                     "fpco_class_compress", "fpco_cc_label", "fpco_aggregate", "fpco_pose_rt"):
            getattr(_lib, name).restype = ctypes.c_int
    return _lib


def set_threads(n):
    """Threads of the oracle's parallel loops (the vote's hn x tn decisions): 1 = scalar port, <= 0 = all cores."""
    f = lib().fpco_set_threads
    f.restype = ctypes.c_int
    return int(f(int(n)))


def get_threads():
    """The count the oracle's parallel loops run on now (its own setting: the process-wide OpenMP count is never touched)."""
    f = lib().fpco_get_threads
    f.restype = ctypes.c_int
    return int(f())


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with code {rc}")


def generate_hypothesis(direct, coords, idxs):
    direct, coords = _f32(direct), _f32(coords)
    idxs = np.ascontiguousarray(idxs, dtype=np.int32)
    tn, vn, _ = direct.shape
    hn = idxs.shape[0]
    hyp = np.zeros((hn, vn, 2), np.float32)
    _check(lib().fpco_generate_hypothesis(_p(direct, c_f32p), _p(coords, c_f32p), _p(idxs, c_i32p),
                                          _p(hyp, c_f32p), tn, vn, hn), "generate_hypothesis")
    return hyp


def voting_for_hypothesis(direct, coords, hyp, inliers, thresh):
    """In place on `inliers` (u8 [hn,vn,tn]), like the reference extension."""
    direct, coords, hyp = _f32(direct), _f32(coords), _f32(hyp)
    assert inliers.dtype == np.uint8 and inliers.flags.c_contiguous
    tn, vn, _ = direct.shape
    hn = hyp.shape[0]
    _check(lib().fpco_voting_for_hypothesis(_p(direct, c_f32p), _p(coords, c_f32p), _p(hyp, c_f32p),
                                            _p(inliers, c_u8p), tn, vn, hn, ctypes.c_float(thresh)),
           "voting_for_hypothesis")
    return inliers


def ransac_voting_layer_v3(mask, vertex, round_hyp_num, inlier_thresh=0.999, min_num=5, max_num=30000,
                           idxs=None, keep=None, seed=0, return_debug=False):
    """mask [n,H,W] (any dtype, != 0 is foreground), vertex [n,H,W,vn,2] (may be a
    strided view) -> [n,vn,2].  idxs: i32 [n,hn,vn,2] or None; keep: u8 [n,H,W] or None."""
    mask = _f32(mask)
    vertex = np.asarray(vertex, dtype=np.float32)
    n, H, W, vn, _ = vertex.shape
    hn = int(round_hyp_num)
    out = np.zeros((n, vn, 2), np.float32)
    dbg = []
    es = vertex.itemsize
    if keep is not None:
        keep = np.ascontiguousarray(keep, dtype=np.uint8)
    for vi in range(vn):
        v = vertex[:, :, :, vi, :]
        if n > 0 and any(s % es for s in v.strides):
            v = np.ascontiguousarray(v)
        sn, sh, sw, sc = (s // es for s in v.strides) if n > 0 else (0, 0, 0, 0)
        ii = None if idxs is None else np.ascontiguousarray(np.asarray(idxs)[:, :, vi, :], dtype=np.int32)
        xy = np.zeros((n, 2), np.float32)
        tn = np.zeros(n, np.int32); widx = np.zeros(n, np.int32); wcnt = np.zeros(n, np.int32)
        icnt = np.zeros(n, np.int32); hyp = np.zeros((n, hn, 2), np.float32); counts = np.zeros((n, hn), np.int32)
        base = ctypes.cast(ctypes.c_void_p(v.ctypes.data if n > 0 else 0), c_f32p)
        _check(lib().fpco_ransac_voting_v3(
            _p(mask, c_f32p), base, ctypes.c_int64(sn), ctypes.c_int64(sh), ctypes.c_int64(sw), ctypes.c_int64(sc),
            n, H, W, hn, _p(ii, c_i32p), _p(keep, c_u8p), ctypes.c_uint64(seed),
            ctypes.c_float(inlier_thresh), int(min_num), int(max_num),
            _p(xy, c_f32p), _p(tn, c_i32p), _p(widx, c_i32p), _p(wcnt, c_i32p), _p(icnt, c_i32p),
            _p(hyp, c_f32p), _p(counts, c_i32p)), "ransac_voting_v3")
        out[:, vi, :] = xy
        dbg.append(dict(tn=tn, win_idx=widx, win_count=wcnt, inlier_count=icnt, hyp=hyp, counts=counts))
    if return_debug:
        return out, dbg
    return out


def class_compress(logits, num_classes, cat_mask=None):
    """logits: dict mask [B,C,H,W], quaternion [B,4(C-1),H,W], scales [B,3(C-1),H,W],
    xy [B,2(C-1),H,W], z [B,C-1,H,W] -> categorical dict (mask i64 [B,H,W], quaternion
    [B,4,H,W], scales [B,3,H,W], xy [B,2,H,W], z [B,H,W])."""
    ml = _f32(logits["mask"]); q = _f32(logits["quaternion"]); s = _f32(logits["scales"])
    xy = _f32(logits["xy"]); z = _f32(logits["z"])
    B, C, H, W = ml.shape
    assert C == num_classes
    HW = H * W
    cm_in = None if cat_mask is None else np.ascontiguousarray(cat_mask, dtype=np.int64)
    cm = np.zeros((B, H, W), np.int64)
    oq = np.zeros((B, 4, H, W), np.float32); os_ = np.zeros((B, 3, H, W), np.float32)
    oxy = np.zeros((B, 2, H, W), np.float32); oz = np.zeros((B, H, W), np.float32)
    _check(lib().fpco_class_compress(_p(ml, c_f32p), _p(q, c_f32p), _p(s, c_f32p), _p(xy, c_f32p), _p(z, c_f32p),
                                     _p(cm_in, c_i64p), B, C, HW, _p(cm, c_i64p), _p(oq, c_f32p), _p(os_, c_f32p),
                                     _p(oxy, c_f32p), _p(oz, c_f32p)), "class_compress")
    return {"mask": cm, "quaternion": oq, "scales": os_, "xy": oxy, "z": oz}


def cc_label(fg):
    """fg: bool/u8 [B,H,W] -> (labels i32 [B,H,W], N)."""
    fg = np.ascontiguousarray(np.asarray(fg) != 0, dtype=np.uint8)
    B, H, W = fg.shape
    labels = np.zeros((B, H, W), np.int32)
    n = ctypes.c_int32(0)
    _check(lib().fpco_cc_label(_p(fg, c_u8p), B, H, W, _p(labels, c_i32p), ctypes.byref(n)), "cc_label")
    return labels, int(n.value)


def aggregate(cat):
    """cat: categorical dict (see class_compress) -> AggData dict of numpy arrays."""
    cm = np.ascontiguousarray(cat["mask"], dtype=np.int64)
    B, H, W = cm.shape
    labels, N = cc_label(cm != 0)
    q = _f32(cat["quaternion"]); s = _f32(cat["scales"]); xy = _f32(cat["xy"]); z = _f32(cat["z"])
    cls = np.zeros(N, np.int64); sid = np.zeros(N, np.int64)
    im = np.zeros((N, H, W), np.float32)
    oq = np.zeros((N, 4), np.float32); os_ = np.zeros((N, 3), np.float32); oz = np.zeros((N, 1), np.float32)
    oxy = np.zeros((N, 2, H, W), np.float32)
    _check(lib().fpco_aggregate(_p(labels, c_i32p), _p(cm, c_i64p), _p(q, c_f32p), _p(s, c_f32p), _p(xy, c_f32p),
                                _p(z, c_f32p), B, H, W, N, _p(cls, c_i64p), _p(sid, c_i64p), _p(im, c_f32p),
                                _p(oq, c_f32p), _p(os_, c_f32p), _p(oz, c_f32p), _p(oxy, c_f32p)), "aggregate")
    return {"class_ids": cls, "sample_ids": sid, "instance_masks": im, "quaternion": oq, "scales": os_,
            "z": oz, "xy": oxy, "labels": labels}


def pose_rt(q, xy, z, inv_intrinsics):
    q, xy, z, k = _f32(q), _f32(xy), _f32(z).reshape(-1), _f32(inv_intrinsics)
    n = q.shape[0]
    R = np.zeros((n, 3, 3), np.float32); T = np.zeros((n, 3), np.float32); RT = np.zeros((n, 4, 4), np.float32)
    _check(lib().fpco_pose_rt(_p(q, c_f32p), _p(xy, c_f32p), _p(z, c_f32p), _p(k, c_f32p), n,
                              _p(R, c_f32p), _p(T, c_f32p), _p(RT, c_f32p)), "pose_rt")
    return R, T, RT


def mask_iou(m1, m2, return_counts=False):
    """gpu_tensor_funcs.py:386-409 batchwise_get_2d_iou: m1 [n1,H,W], m2 [n2,H,W] (non-zero = set) -> f32 [n1,n2]."""
    a = _f32(np.asarray(m1, dtype=np.float32)); b = _f32(np.asarray(m2, dtype=np.float32))
    n1, n2 = a.shape[0], b.shape[0]
    hw = int(np.prod(a.shape[1:])) if n1 else int(np.prod(b.shape[1:]))
    iou = np.zeros((n1, n2), np.float32); inter = np.zeros((n1, n2), np.int64); uni = np.zeros((n1, n2), np.int64)
    c_i64p_ = ctypes.POINTER(ctypes.c_int64)
    f = lib().fpco_mask_iou
    f.restype = ctypes.c_int
    f.argtypes = [c_f32p, ctypes.c_int, c_f32p, ctypes.c_int, ctypes.c_int64, c_f32p, c_i64p_, c_i64p_]
    _check(f(_p(a, c_f32p), n1, _p(b, c_f32p), n2, hw, _p(iou, c_f32p), _p(inter, c_i64p_), _p(uni, c_i64p_)), "mask_iou")
    return (iou, inter, uni) if return_counts else iou


KEYS_TO_STACK = ['instance_masks', 'quaternion', 'R', 'scales', 'xy', 'z', 'T', 'RT']     # matching.py:29-35


def find_matches(preds, gts):
    """matching.py:226-325 batchwise_find_matches on numpy dicts: per ground-truth class (ascending,
    torch.unique), IoU of that class's gt masks against that class's predicted masks (ANY sample of the
    batch — the reference does not compare sample ids), row arg-max (first maximum; a NaN in the row wins,
    as torch.max propagates it), rows whose maximum is not > 0 dropped; matched gt / pred tensors stacked
    as [2, m, ...].  Returns None where the reference does."""
    if not preds or not gts:
        return None
    if preds['class_ids'].shape[0] == 0:
        return None
    out = {'sample_ids': [], 'class_ids': [], 'symmetric_ids': []}
    for cid in np.unique(gts['class_ids']):
        gi = np.where(gts['class_ids'] == cid)[0]
        pi = np.where(preds['class_ids'] == cid)[0]
        if gi.size == 0 or pi.size == 0:
            continue
        iou = mask_iou(gts['instance_masks'][gi], preds['instance_masks'][pi])
        max_pred = np.zeros(gi.size, np.int64); max_v = np.zeros(gi.size, np.float32)
        for r in range(gi.size):
            row = iou[r]
            nan = np.isnan(row)
            k = int(np.argmax(nan)) if nan.any() else int(np.argmax(row))     # torch.max: first NaN, else first maximum
            max_pred[r] = k; max_v[r] = row[k]
        valid = max_v > 0
        if not valid.any():
            continue
        g_sel = gi[valid]; p_sel = pi[max_pred[valid]]
        out['sample_ids'].append(gts['sample_ids'][g_sel])
        out['symmetric_ids'].append(gts['symmetric_ids'][g_sel])
        out['class_ids'].append(np.full(g_sel.size, cid, dtype=gts['class_ids'].dtype))
        for k in gts.keys():
            if k in KEYS_TO_STACK:
                out.setdefault(k, []).append(np.stack((gts[k][g_sel], preds[k][p_sel])))
    for k in list(out.keys()):
        if len(out[k]) == 0:
            return None
        out[k] = np.concatenate(out[k], axis=0 if k in ('sample_ids', 'class_ids', 'symmetric_ids') else 1)
    return out
