"""Drop-in for the reference's `matching.batchwise_find_matches` (F/lib/matching.py:226-325), the step right
after the inference path in `evaluate.py:148` and in every training step (`pose_regressor.py:184`).

The reference loops over the ground-truth classes and, per class, gathers the class's masks (1.2 MB each at
640x480), expands them to [n1,n2,H,W] twice (logical_and / logical_or) and reduces.  Here ONE native call
(`gtf.batchwise_get_2d_iou` -> fpc_mask_iou) yields the IoU of every (ground truth, prediction) pair from
bitsets; IoU is pairwise, so a class's matrix is a sub-matrix of it.  The per-class arg-max / validity logic
runs on that small matrix on the host with torch's own `max` (first maximum, NaN propagates — the reference's
semantics), and the matched tensors are gathered once per key.  One host synchronisation per call (the
reference has several per class: torch.unique, torch.where, boolean indexing).
Same quirk as the reference: sample ids are NOT compared — a ground-truth instance can match a prediction of
another image of the batch that overlaps it in pixel coordinates.
"""
import torch

import gpu_tensor_funcs as gtf

KEYS_TO_STACK = [                      # matching.py:29-35
    'instance_masks',                  # Class
    'quaternion', 'R',                 # Rotation
    'scales',                          # Size
    'xy', 'z', 'T',                    # Translation
    'RT',                              # Transformation
]


def batchwise_find_matches(preds, gts):
    if not preds or not gts:                                  # :229-230
        return None
    if preds['class_ids'].shape[0] == 0:                      # :233-234
        return None
    if gts['class_ids'].shape[0] == 0:                        # no class to loop over: every list stays empty (:316-317)
        return None
    dev = gts['instance_masks'].device
    iou = gtf.batchwise_get_2d_iou(gts['instance_masks'], preds['instance_masks']).cpu()     # the one host sync
    g_cls = gts['class_ids'].cpu()
    p_cls = preds['class_ids'].cpu()
    g_idx, p_idx = [], []
    for class_id in torch.unique(g_cls):                      # ascending, as :252
        gi = torch.where(g_cls == class_id)[0]
        pi = torch.where(p_cls == class_id)[0]
        if gi.shape[0] == 0 or pi.shape[0] == 0:              # :263-264
            continue
        max_v, max_pred = torch.max(iou[gi][:, pi], dim=1)    # :276
        valid = max_v > 0                                     # :280 (NaN > 0 is False)
        if not bool(valid.any()):                             # :283-284
            continue
        g_idx.append(gi[valid])
        p_idx.append(pi[max_pred[valid]])
    if not g_idx:                                             # :316-317
        return None
    g_sel = torch.cat(g_idx).to(dev)
    p_sel = torch.cat(p_idx).to(dev)
    out = {
        'sample_ids': gts['sample_ids'][g_sel],               # :297-299
        'symmetric_ids': gts['symmetric_ids'][g_sel],
        'class_ids': gts['class_ids'][g_sel],                 # = the loop's class id, repeated
    }
    for key in gts.keys():                                    # stack_and_store_data, :41-59
        if key in KEYS_TO_STACK:
            out[key] = torch.stack((gts[key][g_sel], preds[key][p_sel]))
    return out
