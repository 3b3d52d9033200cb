import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def assert_rel(got, want, rtol=1e-4, floor=1.0, what=""):
    """north_star's floating-point bar, element by element: |got - want| <= rtol * max(|want|, floor).  floor = 1 (one pixel
    for vote centres, one unit for pose fields) keeps the bar meaningful for elements near zero."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    tol = rtol * np.maximum(np.abs(want), floor)
    bad = ~(np.abs(got - want) <= tol)
    assert not bad.any(), f"{what}: {int(bad.sum())} elements off, worst {np.abs(got - want)[bad].max():.3e} vs tol {tol[bad].min():.3e}"


def assert_pose(R, T, RT, wR, wT, wRT, what=""):
    """R (and the rotation block of RT): absolute 1e-4 (entries are O(1)); T (and RT's last column): relative 1e-4 per
    element (millimetres, |z| ~ 700); RT's bottom row (0, 0, 0, 1) to 1e-6."""
    np.testing.assert_allclose(R, wR, atol=1e-4, rtol=0, err_msg=what + " R")
    assert_rel(T, wT, what=what + " T")
    np.testing.assert_allclose(RT[:, :3, :3], wRT[:, :3, :3], atol=1e-4, rtol=0, err_msg=what + " RT[:3,:3]")
    assert_rel(RT[:, :3, 3], wRT[:, :3, 3], what=what + " RT[:3,3]")
    np.testing.assert_allclose(RT[:, 3, :], wRT[:, 3, :], atol=1e-6, rtol=0, err_msg=what + " RT[3,:]")   # (0, 0, 0, 1): the reference
                                                                                                        # gets it from a 4x4 inverse


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
