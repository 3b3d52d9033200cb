"""Drop-in for the pose metrics `train.py` selects (F/lib/metrics.py; table F/train.py:208-215): thin accumulators
over the evaluation maths of gpu_tensor_funcs (one native launch per call on GPU tensors, csrc/eval.hip).

The reference derives them from `pl.metrics.Metric` (pytorch_lightning 1.0: `add_state`, distributed reduction, callable
= update + compute); Lightning is not a dependency here, so `Metric` below is a minimal stand-in with the same calling
convention: `metric(gt_pred_matches)` updates the state and returns `compute()`.  State arithmetic follows the reference
line by line, including its running "mean" `(old + new) / 2` (not an arithmetic mean over rounds) and the percentages.
"""
import torch

import gpu_tensor_funcs as gtf


class Metric:
    """update / compute / __call__ / reset, as the reference uses pl.metrics.Metric."""

    def __init__(self, name):
        self.name = name
        self._defaults = {}

    def add_state(self, name, default, dist_reduce_fx=None):
        self._defaults[name] = (default, dist_reduce_fx)
        setattr(self, name, default.clone())

    def reset(self):
        for name, (default, _) in self._defaults.items():
            setattr(self, name, default.clone())

    def __call__(self, *args, **kwargs):
        self.update(*args, **kwargs)
        return self.compute()


def _has(gt_pred_matches, key):
    return gt_pred_matches is not None and key in gt_pred_matches.keys()


class _ThresholdAP(Metric):
    """correct / total over all updates, in percent (DegreeErrorMeanAP :13-52, Iou3dAP :90-128, OffsetAP :166-207)."""

    def __init__(self, name, threshold):
        super().__init__(name)
        self.threshold = threshold
        self.add_state('correct', default=torch.tensor(0), dist_reduce_fx='sum')
        self.add_state('total', default=torch.tensor(0), dist_reduce_fx='sum')

    def _count(self, hits):
        self.correct = self.correct.to(hits.device) + torch.sum(hits.int())
        self.total = self.total + hits.shape[0]

    def compute(self):
        return (self.correct.float() / torch.as_tensor(self.total).float()) * 100


class _RunningMean(Metric):
    """state = (state + mean of this round) / 2 (DegreeError :54-88, Iou3dAccuracy :130-164, OffsetError :209-250)."""

    def __init__(self, name, state):
        super().__init__(name)
        self._state = state
        self.add_state(state, default=torch.tensor(0), dist_reduce_fx='mean')

    def _fold(self, values):
        setattr(self, self._state, (getattr(self, self._state).to(values.device) + torch.mean(values)) / 2)

    def compute(self):
        return getattr(self, self._state)


class DegreeErrorMeanAP(_ThresholdAP):

    def __init__(self, threshold):
        super().__init__(f'degree_error_mAP_{threshold}', threshold)

    def update(self, gt_pred_matches):
        if _has(gt_pred_matches, 'quaternion'):
            q = gt_pred_matches['quaternion']
            self._count(gtf.get_quat_distance(q[0], q[1], gt_pred_matches['symmetric_ids']) < self.threshold)


class DegreeError(_RunningMean):

    def __init__(self):
        super().__init__('degree_error', 'error')

    def update(self, gt_pred_matches):
        if _has(gt_pred_matches, 'quaternion'):
            q = gt_pred_matches['quaternion']
            self._fold(gtf.get_quat_distance(q[0], q[1], gt_pred_matches['symmetric_ids']))


class Iou3dAP(_ThresholdAP):

    def __init__(self, threshold):
        super().__init__(f'3D_iou_mAP_{threshold}', threshold)

    def update(self, gt_pred_matches):
        if _has(gt_pred_matches, 'RT'):
            m = gt_pred_matches
            self._count(gtf.get_3d_ious(m['RT'][0], m['RT'][1], m['scales'][0], m['scales'][1]) > self.threshold)


class Iou3dAccuracy(_RunningMean):

    def __init__(self):
        super().__init__('3D_iou_accuracy', 'accuracy')

    def update(self, gt_pred_matches):
        if _has(gt_pred_matches, 'RT'):
            m = gt_pred_matches
            self._fold(gtf.get_3d_ious(m['RT'][0], m['RT'][1], m['scales'][0], m['scales'][1]) * 100)


class OffsetAP(_ThresholdAP):

    def __init__(self, threshold):
        super().__init__(f'offset_error_mAP_{threshold}cm', threshold)

    def update(self, gt_pred_matches):
        if _has(gt_pred_matches, 'RT'):
            self._count(gtf.from_Ts_get_offset_error(gt_pred_matches['T'][0], gt_pred_matches['T'][1]) < self.threshold)


class OffsetError(_RunningMean):

    def __init__(self):
        super().__init__('offset_error', 'error')

    def update(self, gt_pred_matches):
        if _has(gt_pred_matches, 'RT'):
            self._fold(gtf.from_RTs_get_T_offset_errors(gt_pred_matches['RT'][0], gt_pred_matches['RT'][1]))


def head_training_metrics():
    """The 'pose' block of train.py's metrics table (:208-215); the mask block uses pl.metrics.functional (dice / iou / f1
    of the arg-max mask), which stays with Lightning."""
    return {'pose': {
        'degree_error': {'D': 'matched', 'F': DegreeError()},
        'degree_error_AP_5': {'D': 'matched', 'F': DegreeErrorMeanAP(5)},
        'iou_3d_mAP_0.25': {'D': 'matched', 'F': Iou3dAP(0.25)},
        'iou_3d_accuracy': {'D': 'matched', 'F': Iou3dAccuracy()},
        'offset_error_AP_5cm': {'D': 'matched', 'F': OffsetAP(5)},
        'offset_error': {'D': 'matched', 'F': OffsetError()},
    }}
