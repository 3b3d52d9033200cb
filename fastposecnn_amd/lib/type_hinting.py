"""Dict-key contracts of the hot path (mirrors F/lib/type_hinting.py:5-32)."""
import typing

import torch


class LogitData(typing.TypedDict, total=False):
    mask: torch.Tensor
    quaternion: torch.Tensor
    scales: torch.Tensor
    z: torch.Tensor
    xy: torch.Tensor


class CategoricalData(typing.TypedDict, total=False):
    mask: torch.Tensor
    quaternion: torch.Tensor
    scales: torch.Tensor
    z: torch.Tensor
    xy: torch.Tensor


class AggData(typing.TypedDict, total=False):
    class_ids: torch.Tensor
    sample_ids: torch.Tensor
    instance_masks: torch.Tensor
    quaternion: torch.Tensor
    scales: torch.Tensor
    z: torch.Tensor
    xy: torch.Tensor
    R: torch.Tensor
    T: torch.Tensor
    RT: torch.Tensor
