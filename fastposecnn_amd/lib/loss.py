"""Drop-in for the loss classes `train.py` selects (F/lib/loss.py; criterion table F/train.py:159-187).

Pixel-wise mask losses take (pred, gt) = (forward()'s dict, the batch); matched losses take the dict of
`matching.batchwise_find_matches`: value [0] = ground truth, [1] = prediction.  Semantics follow the reference line by
line where it matters for the numbers: NaN (not zero) for "nothing matched", NaN-filtering before the mean, the symmetric
quaternion loss over 360 rotations about the object's axis in float64, XY / scales as a SUM of per-component means,
Z on log-depth.  `FocalLoss` restates pytorch_toolbelt.losses.FocalLoss (upstream, not in the reference tree; parity
unpinned) as the reference uses it: applied per class to LOG-SOFTMAX outputs as if they were logits.
"""
import functools

import torch
from torch import nn
from torch.nn.modules.loss import _Loss

import gpu_tensor_funcs as gtf


def _nan(gt_pred_matches=None):
    try:
        return torch.tensor(float('nan'), device=gt_pred_matches['instance_masks'].device).float()
    except Exception:
        return torch.tensor(float('nan'), device='cuda' if torch.cuda.is_available() else 'cpu').float()


# ---- mask losses (F/lib/loss.py:26-98) ---------------------------------------------------------------------------
class _MaskLossFn(torch.autograd.Function):
    """One of CE / CCE / Focal (which = 0 / 1 / 2) from the shared forward sums; its own backward launch, so the three loss
    objects stay independent autograd nodes (csrc/train.hip: k_mask_losses)."""

    @staticmethod
    def forward(ctx, logits, x, target, sums, which, ignore_cce, alpha, gamma):
        # x: the contiguous NCHW copy the forward sums were taken from (== logits when those are contiguous): the kernel
        # indexes plain NCHW, so the backward reads x, never a strided / channels_last `logits`
        ctx.save_for_backward(x, target, sums)
        ctx.args = (int(which), int(ignore_cce), float(alpha), float(gamma))
        return (sums[2 * which] / sums[2 * which + 1]).float()

    @staticmethod
    def backward(ctx, g):
        from fastposecnn_amd import _native as nat
        x, t, sums = ctx.saved_tensors
        which, ignore_cce, alpha, gamma = ctx.args
        w3 = torch.zeros(3, dtype=torch.float32, device=x.device)
        w3[which] = (g.double() / sums[2 * which + 1]).float()
        grad = torch.empty_like(x, memory_format=torch.contiguous_format)
        with torch.cuda.device(x.device):
            nat.check(nat.lib().fpc_mask_losses(nat.ptr(x), nat.ptr(t), x.shape[0], x.shape[1], x[0, 0].numel(), -100, ignore_cce,
                                                alpha, gamma, None, nat.ptr(w3), nat.ptr(grad), nat.stream()), "fpc_mask_losses")
        return grad, None, None, None, None, None, None, None


def _fused_mask_loss(which, logits, target, ignore_cce=-1, alpha=0.5, gamma=2):
    """CE / CCE / Focal (which = 0 / 1 / 2) for GPU f32 logits; the forward pass (all three sums) runs once per
    (logits, target, parameters) and is shared by the three loss objects of the criterion table, which are called one after
    the other on the same tensors.  None when the torch-op forms must run."""
    import os
    if not (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() >= 3 and logits.shape[1] <= 32
            and bool(int(os.environ.get("FPC_FUSED_MASK_LOSSES", "1")))):
        return None
    from fastposecnn_amd import _native as nat
    key = (int(ignore_cce), float(alpha), float(gamma))
    cached = getattr(logits, "_fpc_mask_losses", None)
    if cached is not None and cached[0] is target and cached[1] == key and cached[2] == logits._version:
        x, t, sums = cached[3]
    else:
        x = logits.detach()
        x = x if x.is_contiguous() else x.contiguous()
        t = target.to(torch.int64).contiguous()
        # (a target outside [0, C) other than the ignore indices contributes nothing here, where torch's own losses raise:
        # checking would cost a host synchronisation per training step; F/tools/dataset.py only produces class ids)
        sums = torch.zeros(6, dtype=torch.float64, device=x.device)
        with torch.cuda.device(x.device):
            nat.check(nat.lib().fpc_mask_losses(nat.ptr(x), nat.ptr(t), x.shape[0], x.shape[1], x[0, 0].numel(), -100, key[0], key[1],
                                                key[2], nat.ptr(sums), None, None, nat.stream()), "fpc_mask_losses")
        logits._fpc_mask_losses = (target, key, logits._version, (x, t, sums))
    return _MaskLossFn.apply(logits, x, t, sums, which, *key)


class CE(_Loss):

    def __init__(self, ignore_index=-1):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, pred, gt):
        fused = _fused_mask_loss(0, pred['logits']['mask'], gt['mask'])
        if fused is not None:
            return fused
        return nn.functional.cross_entropy(pred['logits']['mask'], gt['mask'])


class CCE(_Loss):

    def __init__(self, from_logits=True, ignore_index=-1):
        super().__init__()
        self.from_logits = from_logits
        self.ignore_index = ignore_index

    def forward(self, pred, gt):
        fused = _fused_mask_loss(1, pred['logits']['mask'], gt['mask'], ignore_cce=self.ignore_index)
        if fused is not None:
            return fused
        y = nn.functional.log_softmax(pred['logits']['mask'], dim=1)
        return nn.functional.nll_loss(y, gt['mask'], ignore_index=self.ignore_index)


def focal_loss_with_logits(output, target, gamma=2.0, alpha=0.25):
    """pytorch_toolbelt.losses.functional.focal_loss_with_logits, reduction='mean', not normalised, no reduced threshold."""
    target = target.type(output.type())
    logpt = nn.functional.binary_cross_entropy_with_logits(output, target, reduction='none')
    pt = torch.exp(-logpt)
    loss = (1.0 - pt).pow(gamma) * logpt
    if alpha is not None:
        loss = loss * (alpha * target + (1 - alpha) * (1 - target))
    return loss.mean()


class FocalLoss(_Loss):
    """pytorch_toolbelt.losses.FocalLoss (multi-class): one-vs-rest binary focal loss per class, summed."""

    def __init__(self, alpha=None, gamma=2, ignore_index=None):
        super().__init__()
        self.alpha, self.gamma, self.ignore_index = alpha, gamma, ignore_index

    def forward(self, label_input, label_target):
        loss = 0
        not_ignored = label_target != self.ignore_index if self.ignore_index is not None else None
        for cls in range(label_input.size(1)):
            cls_target = (label_target == cls).long()
            cls_input = label_input[:, cls, ...]
            if not_ignored is not None:
                cls_target = cls_target[not_ignored]
                cls_input = cls_input[not_ignored]
            loss = loss + focal_loss_with_logits(cls_input, cls_target, gamma=self.gamma, alpha=self.alpha)
        return loss


class Focal(_Loss):

    def __init__(self, key='mask', from_logits=True, alpha=0.5, gamma=2, ignore_index=-1):
        super().__init__()
        self.from_logits = from_logits
        self.ignore_index = ignore_index
        self.alpha = alpha
        self.gamma = gamma

    def forward(self, pred, gt):
        fused = _fused_mask_loss(2, pred['logits']['mask'], gt['mask'], ignore_cce=self.ignore_index, alpha=self.alpha, gamma=self.gamma)
        if fused is not None:
            return fused
        y = nn.functional.log_softmax(pred['logits']['mask'], dim=1)       # the reference feeds log-probabilities (:91-98)
        return FocalLoss(alpha=self.alpha, gamma=self.gamma, ignore_index=self.ignore_index)(y, gt['mask'])


# ---- pixel-wise regression loss (:103-149) -----------------------------------------------------------------------
class MaskedMSELoss(_Loss):

    def __init__(self, key):
        super().__init__()
        self.key = key

    def forward(self, pred, gt):
        cat_mask = pred['categorical']['mask']
        if torch.sum(torch.logical_and(cat_mask != 0, gt['mask'] != 0)) == 0:
            return torch.tensor(float('nan'), device=cat_mask.device).float()
        y_pred, y_gt = pred[self.key], gt[self.key]
        binary = cat_mask != 0
        if len(y_pred.shape) > len(binary.shape):
            binary = torch.unsqueeze(binary, dim=1)
        return nn.functional.mse_loss(y_pred * binary, y_gt)


# ---- matched losses (:240-541) -----------------------------------------------------------------------------------
def dec_empty_check(function):
    """NaN when there are no matches or the key is absent (:240-270)."""

    @functools.wraps(function)
    def wrapper(self, gt_pred_matches=None, **kwargs):
        if gt_pred_matches is not None and self.key in gt_pred_matches.keys():
            return function(self, gt_pred_matches, **kwargs)
        return _nan(gt_pred_matches)

    return wrapper


def _mean_without_nan(loss):
    return torch.mean(loss[torch.isnan(loss) == False])      # noqa: E712 (the reference's filter, :299)


def _component_loss(loss_type):
    try:
        return {'L1': nn.L1Loss, 'SmoothL1': nn.SmoothL1Loss, 'L2': nn.MSELoss}[loss_type]()
    except KeyError:
        raise NotImplementedError(f"{loss_type} is an invalid loss function!")


class QLoss(_Loss):

    def __init__(self, key=None, eps=0.1):
        super().__init__()
        self.eps = eps
        self.key = key if key else 'quaternion'

    @dec_empty_check
    def forward(self, gt_pred_matches):
        gt, pred = gt_pred_matches[self.key][0], gt_pred_matches[self.key][1]
        non_symmetric = torch.where(gt_pred_matches['symmetric_ids'] == 0)[0]
        symmetric = torch.where(gt_pred_matches['symmetric_ids'] != 0)[0]
        loss = torch.cat((self.get_loss(gt[non_symmetric], pred[non_symmetric]),
                          self.get_symmetric_loss(gt[symmetric], pred[symmetric])), dim=0)
        return _mean_without_nan(loss)

    def get_loss(self, gt, pred):
        return self.dot_product_to_loss((gt * pred).sum(dim=1))          # diag(gt @ pred.T), :305

    def get_symmetric_loss(self, gt, pred):
        if gt.shape[0] == 0:
            return torch.tensor([float('nan')], device=gt.device)
        rot_e_gt, e_pred = gtf.quat_symmetric_tf(gt, pred)               # [n,360,4] each
        dot_product = torch.einsum('bij,bij->bi', e_pred.double(), rot_e_gt.double())
        return torch.min(self.dot_product_to_loss(dot_product), dim=1).values

    def dot_product_to_loss(self, dot_product):
        error = 1 - torch.pow(dot_product, 2)
        return torch.log(error + self.eps) - torch.log(torch.tensor(self.eps, device=error.device))


class RLoss(_Loss):

    def __init__(self, key=None, eps=0.1):
        super().__init__()
        self.eps = eps
        self.key = key if key else 'R'

    @dec_empty_check
    def forward(self, gt_pred_matches):
        gt, pred = gt_pred_matches[self.key][0], gt_pred_matches[self.key][1]
        traced = torch.einsum('bii->b', torch.bmm(torch.transpose(gt, 1, 2), pred))
        return _mean_without_nan(torch.acos((traced - 1) / 2))


class TLoss(_Loss):

    def __init__(self, key=None, eps=0.1):
        super().__init__()
        self.eps = eps
        self.key = key if key else 'T'

    @dec_empty_check
    def forward(self, gt_pred_matches):
        gt, pred = gt_pred_matches[self.key][0], gt_pred_matches[self.key][1]
        return _mean_without_nan((gt - pred).norm(dim=1))


class _ComponentwiseLoss(_Loss):
    """Sum over the components of the mean loss of each (XYLoss :431-468, ScalesLoss :505-541)."""
    default_key = None

    def __init__(self, key=None, loss_type='L2', eps=0.1):
        super().__init__()
        self.eps = eps
        self.key = key if key else self.default_key
        self.loss_func = _component_loss(loss_type)

    @dec_empty_check
    def forward(self, gt_pred_matches):
        gt, pred = gt_pred_matches[self.key][0], gt_pred_matches[self.key][1]
        return torch.sum(torch.stack([self.loss_func(gt[:, i], pred[:, i]) for i in range(gt.shape[1])]))


class XYLoss(_ComponentwiseLoss):
    default_key = 'xy'


class ScalesLoss(_ComponentwiseLoss):
    default_key = 'scales'


class ZLoss(_Loss):

    def __init__(self, key=None, loss_type='L2', eps=0.1):
        super().__init__()
        self.eps = eps
        self.key = key if key else 'z'
        self.loss_func = _component_loss(loss_type)

    @dec_empty_check
    def forward(self, gt_pred_matches):
        return self.loss_func(torch.log(gt_pred_matches[self.key][0]), torch.log(gt_pred_matches[self.key][1]))


class _PoseLoss(_Loss):
    """Losses over the assembled poses (F/lib/loss.py:546-626): NaN without matches, NaN-filtered mean."""

    def __init__(self, eps=0.1):
        super().__init__()
        self.eps = eps

    def forward(self, gt_pred_matches):
        if gt_pred_matches is None or 'RT' not in gt_pred_matches.keys():
            return _nan(gt_pred_matches)
        return _mean_without_nan(self.pair_loss(gt_pred_matches))


class Iou3dLoss(_PoseLoss):

    def pair_loss(self, m):
        return 1 - gtf.get_3d_ious(m['RT'][0], m['RT'][1], m['scales'][0], m['scales'][1])


class OffsetLoss(_PoseLoss):

    def pair_loss(self, m):
        return gtf.from_RTs_get_T_offset_errors(m['RT'][0], m['RT'][1]) / 10


def head_training_criterion(xy_loss_type='L2', z_loss_type='L2', scales_loss_type='L2'):
    """The criterion table of F/train.py:159-187 (D = where the inputs come from, weight = the factor in the task sum)."""
    return {
        'mask': {
            'loss_ce': {'D': 'pixel-wise', 'F': CE(), 'weight': 5.0},
            'loss_cce': {'D': 'pixel-wise', 'F': CCE(), 'weight': 5.0},
            'loss_focal': {'D': 'pixel-wise', 'F': Focal(), 'weight': 5.0},
        },
        'quaternion': {'loss_quat': {'D': 'matched', 'F': QLoss(key='quaternion'), 'weight': 0.1}},
        'xy': {'loss_xy': {'D': 'matched', 'F': XYLoss(key='xy', loss_type=xy_loss_type), 'weight': 0.01}},
        'z': {'loss_z': {'D': 'matched', 'F': ZLoss(key='z', loss_type=z_loss_type), 'weight': 0.1}},
        'scales': {'loss_scales': {'D': 'matched', 'F': ScalesLoss(key='scales', loss_type=scales_loss_type), 'weight': 0.1}},
    }


def total_loss(criterion, outputs, batch, gt_pred_matches, perform_matching=True):
    """PoseRegressionTask.shared_step's loss arithmetic (F/lib/pose_regressor.py:188-217, 265-307): per task the weighted
    sum of its non-NaN losses (NaN when all are NaN); the total is the sum of the non-NaN task sums.
    Returns (total, {task: {loss_name: value, 'task_total_loss': value}})."""
    dev = outputs['logits']['mask'].device
    total = torch.tensor(0.0, device=dev)
    report = {}
    for task_name, entries in criterion.items():
        losses = {}
        for loss_name, attrs in entries.items():
            if attrs['D'] == 'pixel-wise':
                losses[loss_name] = attrs['F'](outputs, batch)
            elif attrs['D'] == 'matched' and perform_matching:
                losses[loss_name] = attrs['F'](gt_pred_matches)
        weighted = [v * entries[k]['weight'] for k, v in losses.items() if not bool(torch.isnan(v))]
        task_total = torch.sum(torch.stack(weighted)) if weighted else torch.tensor(float('nan'), device=dev)
        losses['task_total_loss'] = task_total
        report[task_name] = losses
        if not bool(torch.isnan(task_total)):
            total = total + task_total
    return total, report
