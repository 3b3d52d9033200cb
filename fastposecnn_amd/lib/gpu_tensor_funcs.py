"""Drop-in for the hot-path part of F/lib/gpu_tensor_funcs.py:
freeze (:29-32), normalize (:37-50), class_compress (:52-99), batchwise_get_RT (:204-235),
samplewise_get_RT (:237-253), quats_2_rotation_matrix (:306-326).
The evaluation maths that follows the matching (:104-129, :177-202, :328-378, :411-476, :486-547, :563-609, :611-655,
:717-799) is at the end of this file; the rest of the reference file (single-quaternion
converters :258-304, complex APs, memory debugging) is never called from train / evaluate / inference and is not provided.

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
    # the foreground also as bit words for the connected-component labelling (they ride on the mask tensor, see
    # aggregation_layer.fg_bits_of; only where that labelling can use them)
    bits = None
    L = nat.lib()
    if L.fpc_cc_bits_supported(B, H, W):
        bits = torch.empty((B, L.fpc_mask_bits_words(H, W)), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        nat.check(L.fpc_class_compress_bits(nat.ptr(ml), nat.ptr(q), nat.ptr(s), nat.ptr(xy), nat.ptr(z),
                                            nat.ptr(cat_mask_in), B, C, HW, nat.ptr(cat_mask), nat.ptr(oq),
                                            nat.ptr(os_), nat.ptr(oxy), nat.ptr(oz), nat.ptr(bits), nat.stream()),
                  "fpc_class_compress_bits")
    if bits is not None:
        cat_mask._fpc_fg_bits = (bits, cat_mask._version)
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


# ---- evaluation maths right after the matching (reference :104-129, 177-202, 328-378, 411-799) -------
# GPU tensors that need no gradient go through ONE native launch for all pairs (csrc/eval.hip: fpc_pose_errors).  The
# torch-op forms below are the same arithmetic for tensors that carry autograd edges (QLoss / Iou3dLoss) and for CPU
# tensors (plumbing and the parity tests against the reference's goldens).

def cartesian_2_homogeneous_coord(cartesian_coord):
    ones = torch.ones((1, cartesian_coord.shape[1]), device=cartesian_coord.device, dtype=cartesian_coord.dtype)
    return torch.vstack([cartesian_coord, ones])


def homogeneous_2_cartesian_coord(homogeneous_coord):
    return homogeneous_coord[:-1, :] / homogeneous_coord[-1, :]


def transform_3d_camera_coords_to_3d_world_coords(cartesian_camera_coordinates_3d, RT):
    return homogeneous_2_cartesian_coord(torch.inverse(RT) @ cartesian_2_homogeneous_coord(cartesian_camera_coordinates_3d))


_UNIT_BOX = ((1, 1, 1), (1, 1, -1), (-1, 1, 1), (-1, 1, -1), (1, -1, 1), (1, -1, -1), (-1, -1, 1), (-1, -1, -1))


def get_3d_bbox(scale, shift=0):
    """[3] scales -> [3,8] box corners (reference :328-378; corner order matters to get_asymmetric_3d_iou)."""
    unit = torch.tensor(_UNIT_BOX, device=scale.device, dtype=scale.dtype) / 2
    return (unit * torch.unsqueeze(scale, 0) + shift).T


def quaternion_raw_multiply(a, b):
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)


def quaternion_multiply(a, b):
    return normalize(quaternion_raw_multiply(a, b), dim=-1)


_ROT_Q = {}


def _rotation_table(device):
    """The 360 one-degree rotations about the second imaginary axis, built in f32 exactly as the reference does (:764-781)."""
    key = str(device)
    if key not in _ROT_Q:
        degrees = torch.arange(0, 360).float()
        factor = torch.sin(torch.deg2rad(degrees) / 2)
        w = torch.cos(torch.deg2rad(degrees) / 2)
        _ROT_Q[key] = torch.vstack((w, 0 * factor, 1 * factor, 0 * factor)).T.contiguous().to(device)
    return _ROT_Q[key]


def quat_symmetric_tf(tf_q, ex_q):
    """tf_q rotated by every table entry (f64, normalised) and ex_q expanded to the same [n,360,4] shape (:752-799)."""
    rot = _rotation_table(tf_q.device).unsqueeze(0)
    n, r = tf_q.shape[0], rot.shape[1]
    e_tf_q = torch.unsqueeze(tf_q, dim=1).expand((n, r, 4))
    e_ex_q = torch.unsqueeze(ex_q, dim=1).expand((n, r, 4))
    return quaternion_multiply(e_tf_q.double(), rot.expand((n, r, 4)).double()), e_ex_q


def _native_pairs_ok(*tensors):
    return all(t.is_cuda for t in tensors) and not (torch.is_grad_enabled() and any(t.requires_grad for t in tensors))


def _pose_errors(q0=None, q1=None, symmetric_ids=None, RT1=None, RT2=None, s1=None, s2=None, T1=None, T2=None):
    ref = q0 if q0 is not None else (RT1 if RT1 is not None else T1)
    n, dev = ref.shape[0], ref.device
    deg = torch.empty(n, dtype=torch.float64, device=dev) if q0 is not None else None
    iou = torch.empty(n, dtype=torch.float32, device=dev) if RT1 is not None else None
    off = torch.empty(n, dtype=torch.float32, device=dev) if T1 is not None else None
    sym = symmetric_ids.to(torch.int64).contiguous() if symmetric_ids is not None else None
    rot = _rotation_table(dev) if sym is not None else None
    args = [None if t is None else _f32c(t) for t in (q0, q1)] + [sym, rot, 0 if rot is None else rot.shape[0]] + \
        [None if t is None else _f32c(t) for t in (RT1, RT2, s1, s2, T1, T2)]
    with torch.cuda.device(dev):
        nat.check(nat.lib().fpc_pose_errors(*[a if isinstance(a, int) else nat.ptr(a) for a in args], n, nat.ptr(deg), nat.ptr(iou),
                                            nat.ptr(off), nat.stream()), "fpc_pose_errors")
    return deg, iou, off


def get_raw_quat_distance(q0, q1):
    if q0.shape[0] == 0:
        return torch.tensor([float('nan')], device=q0.device)
    if q0.dim() == 2 and q0.dtype == torch.float32 and q1.dtype == torch.float32 and _native_pairs_ok(q0, q1):
        return _pose_errors(q0, q1)[0].float()
    ds = torch.stack(((q0 - q1).norm(dim=-1), (q0 + q1).norm(dim=-1)))
    return torch.rad2deg(torch.min(ds, dim=0).values)


def get_symmetric_quat_distance(q0, q1):
    if q0.shape[0] == 0:
        return torch.tensor([float('nan')], device=q0.device)
    if _native_pairs_ok(q0, q1):
        return _pose_errors(q0, q1, torch.ones(q0.shape[0], dtype=torch.int64, device=q0.device))[0]
    rot_e_q1, e_q0 = quat_symmetric_tf(q1, q0)
    return torch.min(get_raw_quat_distance(e_q0, rot_e_q1), dim=-1).values


def get_quat_distance(q0, q1, symmetric_ids=None):
    """Degree error per pair (:411-436).  With symmetric ids the result lists the non-symmetric pairs first, then the
    symmetric ones (the reference's concatenation), NaNs removed."""
    if symmetric_ids is None:
        return get_raw_quat_distance(q0, q1)
    non_sym_i = torch.where(symmetric_ids == 0)[0]
    sym_i = torch.where(symmetric_ids != 0)[0]
    if q0.shape[0] and _native_pairs_ok(q0, q1):
        deg = _pose_errors(q0, q1, symmetric_ids)[0]                       # one launch for both kinds
        nan = torch.tensor([float('nan')], device=q0.device)
        parts = (deg[non_sym_i].float() if non_sym_i.numel() else nan, deg[sym_i] if sym_i.numel() else nan)
        distances = torch.cat(parts, dim=0)
    else:
        distances = torch.cat((get_raw_quat_distance(q0[non_sym_i], q1[non_sym_i]),
                               get_symmetric_quat_distance(q0[sym_i], q1[sym_i])), dim=0)
    return distances[torch.isnan(distances) == False]      # noqa: E712


def get_asymmetric_3d_iou(RT_1, RT_2, scales_1, scales_2):
    """:499-526, including its reduction of the [3,8] corner matrix over dim 0 (per corner over x / y / z)."""
    bbox_3d_1 = transform_3d_camera_coords_to_3d_world_coords(get_3d_bbox(scales_1, 0), RT_1)
    bbox_3d_2 = transform_3d_camera_coords_to_3d_world_coords(get_3d_bbox(scales_2, 0), RT_2)
    b1max, b1min = torch.amax(bbox_3d_1, dim=0), torch.amin(bbox_3d_1, dim=0)
    b2max, b2min = torch.amax(bbox_3d_2, dim=0), torch.amin(bbox_3d_2, dim=0)
    overlap_min, overlap_max = torch.maximum(b1min, b2min), torch.minimum(b1max, b2max)
    if torch.amin(overlap_max - overlap_min) < 0:
        intersections = 0
    else:
        intersections = torch.prod(overlap_max - overlap_min)
    union = torch.prod(b1max - b1min) + torch.prod(b2max - b2min) - intersections
    return intersections / union


def get_3d_iou(RT_1, RT_2, scales_1, scales_2):
    return get_asymmetric_3d_iou(RT_1, RT_2, scales_1, scales_2)       # the reference's symmetry flag is off (:530-535)


def get_3d_ious(RTs_1, RTs_2, scales_1, scales_2):
    if RTs_1.shape[0] and _native_pairs_ok(RTs_1, RTs_2, scales_1, scales_2):
        return _pose_errors(RT1=RTs_1, RT2=RTs_2, s1=scales_1, s2=scales_2)[1]
    return torch.stack([get_3d_iou(RTs_1[i], RTs_2[i], scales_1[i], scales_2[i]) for i in range(RTs_1.shape[0])])


def from_Ts_get_offset_error(gt_Ts, pred_Ts):
    if gt_Ts.shape[0] and gt_Ts.dtype == torch.float32 and _native_pairs_ok(gt_Ts, pred_Ts):
        return _pose_errors(T1=gt_Ts, T2=pred_Ts)[2]
    return torch.linalg.norm(gt_Ts - pred_Ts, dim=1) * 10


def get_offset_error_from_centroid(center3d_1, center3d_2):
    return torch.sqrt(torch.sum(torch.pow(center3d_1 - center3d_2, 2)))


def from_RTs_get_T_offset_errors(gt_RTs, pred_RTs):
    """:567-609: world positions of the camera origin under inverse(RT); ONE distance over all pairs (the reference sums
    over every element), times 10."""
    origin = torch.tensor([[0, 0, 0]], device=gt_RTs.device, dtype=gt_RTs.dtype).T
    gts = torch.stack([transform_3d_camera_coords_to_3d_world_coords(origin, gt_RTs[i]).flatten() for i in range(gt_RTs.shape[0])])
    preds = torch.stack([transform_3d_camera_coords_to_3d_world_coords(origin, pred_RTs[i]).flatten() for i in range(gt_RTs.shape[0])])
    return get_offset_error_from_centroid(gts, preds) * 10


def calculate_aps(raw_data, metrics_threshold, metrics_operator):
    """:611-655: per metric and class the fraction of (non-NaN) samples that satisfy `operator(sample, threshold)` for
    every threshold; 'mean' over the classes."""
    aps = {}
    for data_key, data in raw_data.items():
        aps[data_key] = {}
        thresholds, operator = metrics_threshold[data_key], metrics_operator[data_key]
        for class_id, class_data in data.items():
            class_data = class_data[torch.isnan(class_data) == False]      # noqa: E712
            hits = operator(class_data.unsqueeze(0), thresholds.unsqueeze(1))
            aps[data_key][class_id] = torch.sum(hits, dim=1) / class_data.shape[0]
        aps[data_key]['mean'] = torch.mean(torch.stack(list(aps[data_key].values())).float(), dim=0)
    return aps


def calculate_complex_aps(raw_data, metrics_threshold, metrics_operator):
    """:657-713: for a key such as 'degree_error+offset_error' (every raw metric whose name occurs in it, stacked) the
    fraction of samples below ALL of its per-metric thresholds, per class and threshold column; 'mean' over the classes."""
    aps = {}
    for data_key, thresholds in metrics_threshold.items():
        aps[data_key] = {}
        data = {}
        for key in [k for k in raw_data.keys() if k in data_key]:
            for class_id, v in raw_data[key].items():
                data[class_id] = torch.stack((data[class_id], v)) if class_id in data else v
        for class_id, class_data in data.items():
            applied = torch.less(torch.unsqueeze(class_data, dim=1), torch.unsqueeze(thresholds, dim=-1))
            mixed = (torch.sum(applied, dim=0) == applied.shape[0]).bool()
            aps[data_key][class_id] = torch.sum(mixed, dim=1) / class_data.shape[1]
        aps[data_key]['mean'] = torch.mean(torch.stack(list(aps[data_key].values())).float(), dim=0)
    return aps
