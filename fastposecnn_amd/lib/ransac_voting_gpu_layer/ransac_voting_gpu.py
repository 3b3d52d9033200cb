"""Drop-in for RV/ransac_voting_gpu.py's live surface: `ransac_voting_layer_v3` (:518-607)
and `b_inv` (:503-516).  The other 13 variants of the reference are never called from
FastPoseCNN and are not provided.

`ransac_voting_layer_v3` keeps the reference signature.  The whole per-instance Python loop —
mask compaction, pair sampling, hypothesis generation, voting, arg-max, winner re-vote and the
2x2 normal-equation solve — runs as one enqueue of HIP kernels for the entire batch
(fastposecnn_amd/csrc/ransac.hip); no device->host synchronisation happens here.
`confidence` / `max_iter` are accepted and ignored: the reference re-votes the SAME samples
every round (idxs is drawn once, :552) so its result does not depend on them.

Keyword-only extensions (not in the reference): `idxs` (i32 [b,hn,vn,2]) injects the RANSAC
pairs, `keep` (u8/bool [b,h,w]) injects the > max_num thinning selection, `seed` fixes the
built-in counter-based sampler (include/fpc_rng.h; default: drawn from torch's CPU generator,
so torch.manual_seed() makes runs repeatable), `return_debug` also returns per-instance
diagnostics (tn, win_idx, win_count, inlier_count, hyp, counts; `return_debug="winner"` leaves out the count rows, which
lets the progressive count run: `set_vote_prune`), `n_dev` (device i32[1]) limits the work
to the first n_dev instances of a capacity-sized batch without a host read (rows past it are left
uninitialised), `mask_bits` (i64 [b, fpc_mask_bits_words(h,w)]: the foreground as bit words, as the aggregation layer
writes them) lets the kernels skip the f32 mask planes — same result, a third of the scan's bytes.
"""
import torch

from fastposecnn_amd import _native as nat


def set_vote_prune(mode=0, cum16=None):
    """The progressive count of the vote (include/fpc.h: fpc_vote_set_prune).  mode 0: never (default), 1: whenever the
    count rows are not asked for.  cum16: e.g. (5, 10) = three passes over 5/16, 5/16, 6/16 of the units."""
    import ctypes
    if cum16 is None:
        nat.check(nat.lib().fpc_vote_set_prune(int(mode), 0, None), "fpc_vote_set_prune")
    else:
        arr = (ctypes.c_int32 * len(cum16))(*[int(c) for c in cum16])
        nat.check(nat.lib().fpc_vote_set_prune(int(mode), len(cum16) + 1, arr), "fpc_vote_set_prune")


def vote_prune_info(b, h, w, hn, dev):
    """i32 [b,8] of the LAST vote call with these sizes on the current stream (include/fpc.h: fpc_vote_prune_info)."""
    L = nat.lib()
    out = torch.empty((b, 8), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        ws = nat.workspace("ransac", dev, L.fpc_ransac_workspace_bytes(b, h, w, hn))
        nat.check(L.fpc_vote_prune_info(nat.ptr(ws), ws.numel(), b, h, w, hn, nat.ptr(out), nat.stream()), "fpc_vote_prune_info")
    return out


def b_inv(b_mat):
    """Batched inverse with pseudo-inverse fallback on singular input (reference :503-516)."""
    try:
        return torch.linalg.inv(b_mat)
    except RuntimeError:  # singular
        return torch.pinverse(b_mat)


def ransac_voting_layer_v3(mask, vertex, round_hyp_num, inlier_thresh=0.999, confidence=0.99, max_iter=20,
                           min_num=5, max_num=30000, *, idxs=None, keep=None, seed=None, return_debug=False, n_dev=None,
                           refine_out=None, mask_bits=None, pose=None):
    """
    :param mask:      [b,h,w]   foreground where != 0
    :param vertex:    [b,h,w,vn,2]  (any strides; the permuted view of hough_voting.py:51 is read in place)
    :param round_hyp_num: hypotheses per instance
    :param pose: (not in the reference) dict(q=[b,4], z=[b] or [b,1], kinv=[3,3], R=[b,3,3], T=[b,3], RT=[b,4,4]) of contiguous f32
                 tensors on the masks' device, vn == 1: the RT assembly of gtf.batchwise_get_RT is appended to the vote's last kernel
                 (fpc_ransac_voting_v3_pose) — R / T / RT are written for the instances the vote processes
    :return: [b,vn,2]  (x = column, y = row)
    """
    nat.require_gpu(mask, vertex, what="ransac_voting_layer_v3")
    b, h, w, vn, two = vertex.shape
    if two != 2 or tuple(mask.shape) != (b, h, w):
        raise RuntimeError("ransac_voting_layer_v3: mask [b,h,w] / vertex [b,h,w,vn,2] shape mismatch")
    hn = int(round_hyp_num)
    dev = mask.device
    # every processed row is written by the kernels (instances below min_num get zeros)
    out = torch.empty((b, vn, 2), dtype=torch.float32, device=dev)
    dbg = []
    if b == 0:
        # the reference's torch.cat([]) guard (:602-605)
        return (out, dbg) if return_debug else out
    if mask.dtype != torch.float32 or not mask.is_contiguous():
        mask = mask.to(torch.float32).contiguous()
    if mask_bits is not None:
        if mask_bits.dtype != torch.int64 or not mask_bits.is_contiguous() or mask_bits.device != dev or \
                tuple(mask_bits.shape) != (b, nat.lib().fpc_mask_bits_words(h, w)):
            raise RuntimeError("ransac_voting_layer_v3: mask_bits must be contiguous i64 [b, fpc_mask_bits_words(h, w)] on the masks' device")
    if vertex.dtype != torch.float32:
        vertex = vertex.float()
    if keep is not None:
        keep = keep.to(torch.uint8).contiguous()
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    L = nat.lib()
    with torch.cuda.device(dev):
        ws = nat.workspace("ransac", dev, L.fpc_ransac_workspace_bytes(b, h, w, hn))
        for vi in range(vn):
            v = vertex[:, :, :, vi, :]
            sn, sh, sw, sc = v.stride()
            ii = None
            if idxs is not None:
                ii = idxs[:, :, vi, :].to(device=dev, dtype=torch.int32).contiguous()
                if tuple(ii.shape) != (b, hn, 2):
                    raise RuntimeError("ransac_voting_layer_v3: idxs must be [b,hn,vn,2]")
            refine = None
            if refine_out is not None:      # f64 [b,vn,8]: winner, normal equations, inlier count (training backward)
                refine = refine_out[:, vi, :] if vn == 1 else torch.empty((b, 8), dtype=torch.float64, device=dev)
            xy = out[:, vi, :] if vn == 1 else torch.empty((b, 2), dtype=torch.float32, device=dev)
            d = None
            if return_debug:
                d = dict(tn=torch.empty(b, dtype=torch.int32, device=dev),
                         win_idx=torch.empty(b, dtype=torch.int32, device=dev),
                         win_count=torch.empty(b, dtype=torch.int32, device=dev),
                         inlier_count=torch.empty(b, dtype=torch.int32, device=dev),
                         hyp=torch.empty((b, hn, 2), dtype=torch.float32, device=dev))
                if return_debug != "winner":     # the count rows of EVERY hypothesis: the exhaustive count runs (include/fpc.h)
                    d["counts"] = torch.empty((b, hn), dtype=torch.int32, device=dev)
            pp = [None] * 6
            if pose is not None:
                if vn != 1:
                    raise RuntimeError("ransac_voting_layer_v3: pose needs vn == 1")
                for k_, shp in (("q", (b, 4)), ("z", (b,)), ("kinv", (3, 3)), ("R", (b, 3, 3)), ("T", (b, 3)), ("RT", (b, 4, 4))):
                    t_ = pose[k_]
                    if t_.dtype != torch.float32 or not t_.is_contiguous() or t_.device != dev or t_.numel() != int(torch.Size(shp).numel()):
                        raise RuntimeError("ransac_voting_layer_v3: pose[%r] must be a contiguous f32 tensor of %s on the masks' device" % (k_, shp))
                pp = [nat.ptr(pose[k_]) for k_ in ("q", "z", "kinv", "R", "T", "RT")]
            nat.check(L.fpc_ransac_voting_v3_pose(
                nat.ptr(mask), nat.ptr(mask_bits), v.data_ptr(), sn, sh, sw, sc, b, nat.ptr(n_dev), h, w, hn, nat.ptr(ii), nat.ptr(keep),
                (seed + vi) & (2 ** 64 - 1), float(inlier_thresh), int(min_num), int(max_num), nat.ptr(xy),
                nat.ptr(d["tn"]) if d else None, nat.ptr(d["win_idx"]) if d else None,
                nat.ptr(d["win_count"]) if d else None, nat.ptr(d["inlier_count"]) if d else None,
                nat.ptr(d["hyp"]) if d else None, nat.ptr(d.get("counts")) if d else None, nat.ptr(refine),
                pp[0], pp[1], pp[2], pp[3], pp[4], pp[5],
                nat.ptr(ws), ws.numel(), nat.stream()), "fpc_ransac_voting_v3_pose")
            if vn != 1:
                out[:, vi, :] = xy
                if refine is not None:
                    refine_out[:, vi, :] = refine
            if d:
                dbg.append(d)
    return (out, dbg) if return_debug else out
