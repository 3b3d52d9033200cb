"""Dict-key contracts of the hot path: which tensors travel under which key between the stages.

The reference declares them as TypedDict classes (F/lib/type_hinting.py:5-32); the same names are exported here,
built from one table so that the per-pixel and the per-instance key sets are stated once.
"""
import typing

import torch

_PER_PIXEL = ("mask", "quaternion", "scales", "z", "xy")                       # logits and their class-compressed form
_PER_INSTANCE = ("class_ids", "sample_ids",                                    # meta
                 "instance_masks", "quaternion", "scales", "z", "xy",          # aggregated features
                 "R", "T", "RT")                                               # pose
_MATCHED = ("class_ids", "sample_ids", "symmetric_ids") + _PER_INSTANCE[2:]    # ground truth / prediction pairs, stacked [2, m, ...]


def _contract(name, keys):
    return typing.TypedDict(name, {k: torch.Tensor for k in keys}, total=False)


LogitData = _contract("LogitData", _PER_PIXEL)
CategoricalData = _contract("CategoricalData", _PER_PIXEL)
AggData = _contract("AggData", _PER_INSTANCE)
MatchedData = _contract("MatchedData", _MATCHED)
