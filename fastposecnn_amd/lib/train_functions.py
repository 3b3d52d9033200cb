"""Differentiable post-network path for training: the inference kernels as the forward, hand-written
HIP kernels as the backward (csrc/train.hip).

The reference trains through torch autograd over its Python forward (F/lib/pose_regressor.py:168-263): gradients reach the
heads through the class gathers (F/lib/gpu_tensor_funcs.py:52-99), the masked means of the aggregation layer
(F/lib/aggregation_layer.py:119-156) and, for the vote, ONLY through the final least squares over the winner's inliers
(RV/ransac_voting_gpu.py:583-599) — sampling, voting and arg-max are non-differentiable selectors.  The same derivative is
computed here without the [N,A,H,W] expansions: `ClassCompressFn` and `PostNetworkFn` are `torch.autograd.Function`s whose
backward is one kernel each; `VoteRefineFn` gives a direct `ransac_voting_layer_v3` caller the vote term alone.
"""
import torch

from fastposecnn_amd import _native as nat

import gpu_tensor_funcs as gtf
import ransac_voting_gpu_layer.ransac_voting_gpu as rvg


def _g(t, like_shape, dev):
    """A contiguous f32 gradient or None (autograd passes None for unused outputs)."""
    if t is None:
        return None
    t = t.to(torch.float32).contiguous()
    assert tuple(t.shape) == tuple(like_shape) and t.device == dev
    return t


class ClassCompressFn(torch.autograd.Function):
    """Model.class_compression (arg-max + gather + normalise, F/lib/pose_regressor.py:445-457) with a native backward."""

    @staticmethod
    def forward(ctx, num_classes, mask_logits, quat, scales, xy, z):
        logits = {"mask": mask_logits, "quaternion": quat, "scales": scales, "xy": xy, "z": z}
        out, cat_mask = gtf._class_compress_hip(num_classes, None, logits)
        ctx.C = num_classes
        ctx.shapes = (quat.shape, scales.shape, xy.shape, z.shape)
        ctx.save_for_backward(cat_mask, gtf._f32c(quat), gtf._f32c(xy))
        ctx.mark_non_differentiable(cat_mask)
        return out["quaternion"], out["scales"], out["xy"], out["z"], cat_mask

    @staticmethod
    def backward(ctx, go_q, go_s, go_xy, go_z, _go_mask):
        cat_mask, quat, xy = ctx.saved_tensors
        B, H, W = cat_mask.shape
        dev = cat_mask.device
        go_q, go_s = _g(go_q, (B, 4, H, W), dev), _g(go_s, (B, 3, H, W), dev)
        go_xy, go_z = _g(go_xy, (B, 2, H, W), dev), _g(go_z, (B, H, W), dev)
        gq, gs, gxy, gz = (torch.empty(s, dtype=torch.float32, device=dev) for s in ctx.shapes)
        with torch.cuda.device(dev):
            nat.check(nat.lib().fpc_class_compress_backward(
                nat.ptr(cat_mask), nat.ptr(quat), nat.ptr(xy), nat.ptr(go_q), nat.ptr(go_s), nat.ptr(go_xy), nat.ptr(go_z),
                B, ctx.C, H * W, nat.ptr(gq), nat.ptr(gs), nat.ptr(gxy), nat.ptr(gz), nat.stream()),
                "fpc_class_compress_backward")
        return None, None, gq, gs, gxy, gz


def _solve2_sym(a00, a01, a11, g):
    """lam = b_inv(A) g for the symmetric 2x2 normal matrices (f64), with csrc/ransac.hip:solve2_sym's singular rule."""
    tr, det = a00 + a11, a00 * a11 - a01 * a01
    regular = det > 1e-12 * tr * tr
    inv = torch.where(regular, 1.0 / torch.where(regular, det, torch.ones_like(det)), torch.zeros_like(det))
    s = torch.where(tr > 0, 1.0 / torch.where(tr > 0, tr * tr, torch.ones_like(tr)), torch.zeros_like(tr))
    l0 = torch.where(regular, (a11 * g[:, 0] - a01 * g[:, 1]) * inv, (a00 * g[:, 0] + a01 * g[:, 1]) * s)
    l1 = torch.where(regular, (-a01 * g[:, 0] + a00 * g[:, 1]) * inv, (a01 * g[:, 0] + a11 * g[:, 1]) * s)
    ok = tr > 0
    return torch.where(ok, l0, torch.zeros_like(l0)), torch.where(ok, l1, torch.zeros_like(l1))


class PostNetworkFn(torch.autograd.Function):
    """aggregate -> hough voting on the categorical planes; differentiable in (quaternion, scales, xy, z).

    forward returns (class_ids, sample_ids, instance_masks, xy_mask, quaternion [N,4], scales [N,3], xy [N,2], z [N,1]);
    the first four are non-differentiable."""

    @staticmethod
    def forward(ctx, model, cat_mask, cq, cs, cxy, cz, seed):
        layer = model.aggregation_layer
        hp = model.HPARAM
        cm = cat_mask.to(torch.int64).contiguous()
        cat = {"quaternion": cq, "scales": cs, "xy": cxy, "z": cz}
        labels, N = layer.batchwise_break_segmentation_mask(cm)               # the one host read (N shapes the outputs)
        dev = cm.device
        stats = torch.empty((N, 2), dtype=torch.float32, device=dev)
        agg = layer._aggregate(cat, cm, labels, N, None, stats=stats)
        refine = torch.zeros((N, 1, 8), dtype=torch.float64, device=dev)
        xy_mask = agg["xy"]
        if N > 0:
            vertex = torch.unsqueeze(xy_mask.permute(0, 2, 3, 1), dim=3)
            voted = rvg.ransac_voting_layer_v3(mask=agg["instance_masks"], vertex=vertex,
                                               round_hyp_num=hp.HV_NUM_OF_HYPOTHESES, seed=seed, refine_out=refine)
            xy = torch.squeeze(voted, dim=1)
        else:
            xy = torch.empty((0, 2), dtype=torch.float32, device=dev)
        ctx.seed = seed
        ctx.vote_args = (0.999, 5, 30000)           # ransac_voting_layer_v3's defaults, as HoughVotingLayer calls it
        ctx.save_for_backward(labels, gtf._f32c(cxy), stats, refine, agg["quaternion"], agg["z"], xy)
        for t in (agg["class_ids"], agg["sample_ids"], agg["instance_masks"], xy_mask):
            ctx.mark_non_differentiable(t)
        return agg["class_ids"], agg["sample_ids"], agg["instance_masks"], xy_mask, agg["quaternion"], agg["scales"], xy, agg["z"]

    @staticmethod
    def backward(ctx, _g_cls, _g_smp, _g_msk, _g_xym, g_quat, g_scales, g_xy, g_z):
        labels, cxy, stats, refine, q_hat, z_out, xy = ctx.saved_tensors
        B, H, W = labels.shape
        N = stats.shape[0]
        dev = labels.device
        f64 = dict(dtype=torch.float64, device=dev)
        tab = torch.zeros((N, 16), **f64)
        if N > 0:
            cnt = stats[:, 0].double().clamp(min=1.0)
            nq = stats[:, 1].double()
            if g_quat is not None:          # mean -> normalise (F/lib/aggregation_layer.py:139-152)
                g, qh = g_quat.double(), q_hat.double()
                proj = (g - qh * (qh * g).sum(dim=1, keepdim=True)) / torch.where(nq != 0, nq, torch.ones_like(nq))[:, None]
                tab[:, 0:4] = torch.where((nq != 0)[:, None], proj, g) / cnt[:, None]
            if g_scales is not None:
                tab[:, 4:7] = g_scales.double() / cnt[:, None]
            if g_z is not None:             # exp(mean(log-depth plane)) (:146-148)
                tab[:, 7] = g_z.double().reshape(-1) * z_out.double().reshape(-1) / cnt
            if g_xy is not None:
                r = refine[:, 0, :]
                l0, l1 = _solve2_sym(r[:, 2], r[:, 3], r[:, 4], g_xy.double())
                tab[:, 8], tab[:, 9] = l0, l1
                tab[:, 10:12] = xy.double()
                tab[:, 12:14] = r[:, 0:2]
                tab[:, 14] = stats[:, 0].double()
                tab[:, 15] = (stats[:, 0] >= ctx.vote_args[1]).double()
        gq = torch.empty((B, 4, H, W), dtype=torch.float32, device=dev)
        gs = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
        gxy = torch.empty((B, 2, H, W), dtype=torch.float32, device=dev)
        gz = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        thresh, _min_num, max_num = ctx.vote_args
        with torch.cuda.device(dev):
            nat.check(nat.lib().fpc_post_network_backward(
                nat.ptr(labels), nat.ptr(cxy), B, H, W, N, None, nat.ptr(tab), thresh, max_num, ctx.seed & (2 ** 64 - 1), None,
                nat.ptr(gq), nat.ptr(gs), nat.ptr(gxy), nat.ptr(gz), nat.stream()), "fpc_post_network_backward")
        return None, None, gq, gs, gxy, gz, None


class VoteRefineFn(torch.autograd.Function):
    """ransac_voting_layer_v3 with the derivative of its final least squares (RV/ransac_voting_gpu.py:583-599) with
    respect to `vertex`; for vn = 1 (FastPoseCNN's only use)."""

    @staticmethod
    def forward(ctx, mask, vertex, round_hyp_num, inlier_thresh, min_num, max_num, seed):
        b, h, w, vn, _ = vertex.shape
        if vn != 1:
            raise RuntimeError("VoteRefineFn: vn must be 1")
        refine = torch.zeros((b, 1, 8), dtype=torch.float64, device=mask.device)
        maskf = mask.to(torch.float32).contiguous()
        out = rvg.ransac_voting_layer_v3(maskf, vertex, round_hyp_num, inlier_thresh, min_num=min_num, max_num=max_num,
                                         seed=seed, refine_out=refine)
        ctx.args = (float(inlier_thresh), int(min_num), int(max_num), int(seed))
        ctx.save_for_backward(maskf, vertex.detach(), refine, out)
        return out

    @staticmethod
    def backward(ctx, g_out):
        maskf, vertex, refine, out = ctx.saved_tensors
        thresh, min_num, max_num, seed = ctx.args
        b, h, w, _, _ = vertex.shape
        dev = maskf.device
        fg = (maskf != 0).flatten(1).sum(dim=1).double()
        r = refine[:, 0, :]
        l0, l1 = _solve2_sym(r[:, 2], r[:, 3], r[:, 4], g_out[:, 0, :].double())
        tab = torch.stack([l0, l1, out[:, 0, 0].double(), out[:, 0, 1].double(), r[:, 0], r[:, 1], fg, (fg >= min_num).double()],
                          dim=1).contiguous()
        v = vertex[:, :, :, 0, :].float()
        sn, sh, sw, sc = v.stride()
        g = torch.empty((b, 2, h, w), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            nat.check(nat.lib().fpc_vote_refine_backward(nat.ptr(maskf), v.data_ptr(), sn, sh, sw, sc, b, h, w, nat.ptr(tab),
                                                         thresh, max_num, seed & (2 ** 64 - 1), None, nat.ptr(g), nat.stream()),
                      "fpc_vote_refine_backward")
        return None, g.permute(0, 2, 3, 1).unsqueeze(3), None, None, None, None, None


def class_compression_train(num_classes, logits):
    q, s, xy, z, cat_mask = ClassCompressFn.apply(num_classes, logits["mask"], logits["quaternion"], logits["scales"],
                                                  logits["xy"], logits["z"])
    return {"quaternion": q, "scales": s, "xy": xy, "z": z, "mask": cat_mask}


def post_network_train(model, categorical, seed=None):
    """Model.agg_hough_and_generate_RT for a training step: same dict as the inference path, with autograd edges from
    quaternion / scales / xy / z (and R / T / RT through them) back to the categorical planes."""
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    cls, smp, masks, xy_mask, quat, scales, xy, z = PostNetworkFn.apply(
        model, categorical["mask"], categorical["quaternion"], categorical["scales"], categorical["xy"], categorical["z"], seed)
    if cls.shape[0] == 0:
        cls = cls.float()                       # the reference's float class ids for an empty batch
    hyp = xy.unsqueeze(1)
    agg = {"class_ids": cls, "instance_masks": masks, "sample_ids": smp, "quaternion": quat, "scales": scales, "xy": xy,
           "z": z, "hypothesis": hyp, "pruned_hypothesis": hyp, "xy_mask": xy_mask}
    if model.HPARAM.PERFORM_RT_CALCULATION:
        agg = gtf.samplewise_get_RT(agg, model._inv_k(quat.device))
    return agg
