"""Drop-in for the hot-path part of F/lib/gpu_tensor_funcs.py:
freeze (:29-32), normalize (:37-50), class_compress (:52-99), batchwise_get_RT (:204-235),
samplewise_get_RT (:237-253), quats_2_rotation_matrix (:306-326).
The evaluation maths of the reference file (:104-202, :258-304, :328-799) is outside the hot
path and is not provided.

GPU tensors go through libfpc_hip.so (fastposecnn_amd/csrc/class_compress.hip, pose.hip);
a missing library raises.  CPU tensors are only accepted by `normalize`, `freeze`,
`quats_2_rotation_matrix` and `class_compress` (plain torch ops) — that is BASELINE.json's
config 1, the reference's own CPU plumbing case with aggregation disabled; everything
downstream of class compression exists only as HIP kernels.
"""
import torch

from fastposecnn_amd import _native as nat


def freeze(dict_of_params):
    for param in dict_of_params.parameters():
        param.requires_grad = False


def normalize(data, dim):
    norm_data = data.norm(dim=dim, keepdim=True)
    safe_norm_data = torch.where(norm_data != 0, norm_data.float(), torch.ones_like(norm_data).float())
    return data / safe_norm_data


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _class_compress_hip(num_of_classes, cat_mask_in, logits):
    ml = logits.get("mask") if cat_mask_in is None else None
    q, s, xy, z = (_f32c(logits[k]) for k in ("quaternion", "scales", "xy", "z"))
    B, _, H, W = q.shape
    C, G, HW = num_of_classes, num_of_classes - 1, H * W
    if q.shape[1] != 4 * G or s.shape[1] != 3 * G or xy.shape[1] != 2 * G or z.shape[1] != G:
        raise RuntimeError("class_compress: channel counts do not match num_of_classes")
    dev = q.device
    if ml is not None:
        ml = _f32c(ml)
        if tuple(ml.shape) != (B, C, H, W):
            raise RuntimeError("class_compress: mask logits must be [B,C,H,W]")
    if cat_mask_in is not None:
        cat_mask_in = cat_mask_in.to(torch.int64).contiguous()
    cat_mask = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    oq = torch.empty((B, 4, H, W), dtype=torch.float32, device=dev)
    os_ = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
    oxy = torch.empty((B, 2, H, W), dtype=torch.float32, device=dev)
    oz = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().fpc_class_compress(nat.ptr(ml), nat.ptr(q), nat.ptr(s), nat.ptr(xy), nat.ptr(z),
                                               nat.ptr(cat_mask_in), B, C, HW, nat.ptr(cat_mask), nat.ptr(oq),
                                               nat.ptr(os_), nat.ptr(oxy), nat.ptr(oz), nat.stream()),
                  "fpc_class_compress")
    return {"quaternion": oq, "scales": os_, "xy": oxy, "z": oz}, cat_mask


def _class_compress_cpu(num_of_classes, cat_mask, logits):
    """Config-1 plumbing on CPU tensors: gather of the arg-max class's channel group."""
    out = {}
    fg = (cat_mask != 0).unsqueeze(1)
    grp = (cat_mask - 1).clamp(min=0)
    for key, v in logits.items():
        if key == "mask":
            continue
        b, ch, h, w = v.shape
        a = ch // (num_of_classes - 1)
        idx = (grp.unsqueeze(1) * a + torch.arange(a, device=v.device).view(1, a, 1, 1))
        sel = torch.gather(v, 1, idx) * fg
        if key == "z":
            sel = sel.squeeze(1)
        elif key in ("quaternion", "xy"):
            sel = normalize(sel, dim=1)
        out[key] = sel
    return out


def class_compress(num_of_classes, cat_mask, logits):
    """Reference signature (gpu_tensor_funcs.py:52-55): returns the categorical dict WITHOUT 'mask'."""
    if cat_mask.is_cuda:
        out, _ = _class_compress_hip(num_of_classes, cat_mask, logits)
        return out
    return _class_compress_cpu(num_of_classes, cat_mask, logits)


def class_compression_fused(num_of_classes, logits):
    """Model.class_compression (pose_regressor.py:445-457) in one kernel: arg-max of the
    log-softmax + compression.  Returns the categorical dict INCLUDING 'mask' (i64)."""
    ml = logits["mask"]
    if ml.is_cuda:
        out, cat_mask = _class_compress_hip(num_of_classes, None, logits)
    else:
        cat_mask = torch.argmax(torch.nn.LogSoftmax(dim=1)(ml), dim=1)
        out = _class_compress_cpu(num_of_classes, cat_mask, logits)
    out["mask"] = cat_mask
    return out


def quats_2_rotation_matrix(q):
    q1, q2, q3, q4 = q.unbind(dim=-1)
    a, b, c, d = q1 * q1, q2 * q2, q3 * q3, q4 * q4
    R = torch.stack([
        torch.stack([a - b - c + d, 2 * (q1 * q2 + q3 * q4), 2 * (q1 * q3 - q2 * q4)], dim=-1),
        torch.stack([2 * (q1 * q2 - q3 * q4), -a + b - c + d, 2 * (q2 * q3 + q1 * q4)], dim=-1),
        torch.stack([2 * (q1 * q3 + q2 * q4), 2 * (q2 * q3 - q1 * q4), -a - b + c + d], dim=-1),
    ], dim=-2)
    return torch.transpose(R, dim0=-2, dim1=-1)


def _batchwise_get_RT_autograd(q, xys, exp_zs, inv_intrinsics):
    """The same maths (reference :204-235) in torch ops on the [n, .] tensors, for a training step: R / T / RT keep their
    autograd edges to q, xy, z.  R is orthonormal for a unit quaternion, so inv(inv_RT) is written out: RT = [R^-T | -R^-T T]
    with inv_R = R^T."""
    n = q.shape[0]
    z = exp_zs.reshape(n, 1) / 1000
    T = (inv_intrinsics @ torch.cat([xys * z, z], dim=1).T).T
    norm = q.norm(dim=1, keepdim=True)
    qn = q / torch.where(norm > 0, norm, torch.ones_like(norm))
    R = quats_2_rotation_matrix(qn)
    inv_R = torch.inverse(R) if n else R
    top = torch.cat([inv_R, T.unsqueeze(-1)], dim=-1)
    bottom = torch.tensor([0, 0, 0, 1], device=q.device, dtype=q.dtype).expand((n, 1, 4))
    inv_RT = torch.cat([top, bottom], dim=1)
    RT = torch.inverse(inv_RT) if n else inv_RT
    return R, T, RT


def batchwise_get_RT(q, xys, exp_zs, inv_intrinsics):
    """q [n,4] scalar-last, xys [n,2], exp_zs [n,1], inv_intrinsics [3,3] -> R [n,3,3], T [n,3], RT [n,4,4]."""
    nat.require_gpu(q, xys, exp_zs, what="batchwise_get_RT")
    if torch.is_grad_enabled() and (q.requires_grad or xys.requires_grad or exp_zs.requires_grad):
        return _batchwise_get_RT_autograd(q, xys, exp_zs, inv_intrinsics.to(q.device))
    n = q.shape[0]
    dev = q.device
    R = torch.empty((n, 3, 3), dtype=torch.float32, device=dev)
    T = torch.empty((n, 3), dtype=torch.float32, device=dev)
    RT = torch.empty((n, 4, 4), dtype=torch.float32, device=dev)
    if n == 0:
        return R, T, RT
    q, xys = _f32c(q), _f32c(xys)
    z = _f32c(exp_zs).reshape(-1)
    k = _f32c(inv_intrinsics.to(dev))
    with torch.cuda.device(dev):
        nat.check(nat.lib().fpc_pose_rt(nat.ptr(q), nat.ptr(xys), nat.ptr(z), nat.ptr(k), n, nat.ptr(R), nat.ptr(T),
                                        nat.ptr(RT), nat.stream()), "fpc_pose_rt")
    return R, T, RT


def samplewise_get_RT(agg_data, inv_intrinsics):
    R_data, T_data, RT_data = batchwise_get_RT(agg_data['quaternion'], agg_data['xy'], agg_data['z'], inv_intrinsics)
    agg_data['R'] = R_data
    agg_data['T'] = T_data
    agg_data['RT'] = RT_data
    return agg_data


def batchwise_get_2d_iou(batch_masks1, batch_masks2):
    """Reference signature (gpu_tensor_funcs.py:386-409): masks [n1,H,W], [n2,H,W] (any dtype, non-zero = set)
    -> IoU f32 [n1,n2].  One native call (fpc_mask_iou): every mask is read once and packed into a bitset;
    the reference's [n1,n2,H,W] logical_and / logical_or expansions are never built.  GPU tensors only."""
    if not (batch_masks1.is_cuda and batch_masks2.is_cuda):
        raise RuntimeError("batchwise_get_2d_iou: the native path needs GPU tensors (there is no CPU fallback)")
    n1, n2 = batch_masks1.shape[0], batch_masks2.shape[0]
    dev = batch_masks1.device
    iou = torch.empty((n1, n2), dtype=torch.float32, device=dev)
    if n1 == 0 or n2 == 0:
        return iou
    if batch_masks1.shape[1:] != batch_masks2.shape[1:]:
        raise RuntimeError("batchwise_get_2d_iou: mask shapes differ")

    def prep(m):
        if m.dtype == torch.float32:
            return m.contiguous(), 4
        if m.dtype in (torch.bool, torch.uint8):
            return m.contiguous().view(torch.uint8), 1
        return (m != 0).contiguous().view(torch.uint8), 1

    a, ea = prep(batch_masks1)
    b, eb = prep(batch_masks2)
    if ea != eb:                      # mixed stacks: bring both to bytes
        a = (a != 0).view(torch.uint8) if ea == 4 else a
        b = (b != 0).view(torch.uint8) if eb == 4 else b
        ea = 1
    hw = a[0].numel()
    L = nat.lib()
    ws = torch.empty(L.fpc_mask_iou_workspace_bytes(n1, n2, hw), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        nat.check(L.fpc_mask_iou(nat.ptr(a), n1, nat.ptr(b), n2, hw, ea, nat.ptr(iou), None, None, nat.ptr(ws), ws.numel(),
                                 nat.stream()), "fpc_mask_iou")
    return iou
