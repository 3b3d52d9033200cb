"""Drop-in for the reference's compiled extension module `ransac_voting`
(pybind exports at RV/src/ransac_voting.cpp:102-107), backed by libfpc_hip.so.

Same two live entry points, same argument order, same contracts:
inputs must be GPU-resident and contiguous (CHECK_INPUT, ransac_voting.cpp:7-9 -> RuntimeError),
`generate_hypothesis` returns a fresh tensor with degenerate pairs left at zero,
`voting_for_hypothesis` mutates the caller's u8 `inliers` in place, writing only ones.
Kernels run on torch's current HIP stream (the reference used the legacy default stream).
The two *_vanishing_point twins are dead code in FastPoseCNN and raise.
"""
import torch

from fastposecnn_amd import _native as nat


def _check_input(t, name, dtype):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}")


def generate_hypothesis(direct, coords, idxs):
    """direct f32 [tn,vn,2], coords f32 [tn,2], idxs i32 [hn,vn,2] -> f32 [hn,vn,2]."""
    _check_input(direct, "direct", torch.float32)
    _check_input(coords, "coords", torch.float32)
    _check_input(idxs, "idxs", torch.int32)
    tn, vn = direct.shape[0], direct.shape[1]
    hn = idxs.shape[0]
    if direct.shape[2] != 2 or tuple(coords.shape) != (tn, 2) or tuple(idxs.shape[1:]) != (vn, 2):
        raise RuntimeError("generate_hypothesis: shape mismatch")
    hyp = torch.empty((hn, vn, 2), dtype=torch.float32, device=direct.device)
    with torch.cuda.device(direct.device):
        nat.check(nat.lib().fpc_generate_hypothesis(nat.ptr(direct), nat.ptr(coords), nat.ptr(idxs), nat.ptr(hyp),
                                                    tn, vn, hn, nat.stream()), "fpc_generate_hypothesis")
    return hyp


def voting_for_hypothesis(direct, coords, hypo_pts, inliers, inlier_thresh):
    """inliers u8 [hn,vn,tn] is written in place with 1 where the vote of pixel ti agrees with
    hypothesis hi (cos > inlier_thresh, strict and signed)."""
    _check_input(direct, "direct", torch.float32)
    _check_input(coords, "coords", torch.float32)
    _check_input(hypo_pts, "hypo_pts", torch.float32)
    _check_input(inliers, "inliers", torch.uint8)
    tn, vn = direct.shape[0], direct.shape[1]
    hn = hypo_pts.shape[0]
    if tuple(hypo_pts.shape[1:]) != (vn, 2) or tuple(inliers.shape) != (hn, vn, tn):
        raise RuntimeError("voting_for_hypothesis: shape mismatch")
    with torch.cuda.device(direct.device):
        nat.check(nat.lib().fpc_voting_for_hypothesis(nat.ptr(direct), nat.ptr(coords), nat.ptr(hypo_pts),
                                                      nat.ptr(inliers), tn, vn, hn, float(inlier_thresh),
                                                      nat.stream()), "fpc_voting_for_hypothesis")


def generate_hypothesis_vanishing_point(*args, **kwargs):
    raise NotImplementedError("vanishing-point voting is not on FastPoseCNN's path (SURVEY.md section 2.2)")


def voting_for_hypothesis_vanishing_point(*args, **kwargs):
    raise NotImplementedError("vanishing-point voting is not on FastPoseCNN's path (SURVEY.md section 2.2)")
