"""Ground-truth half of the dataset item (SURVEY.md 8f rank 3 remainder; VERDICT r3 Missing 2): fastposecnn_amd.tools.dataset's
NOCSDataset against the sample dicts the REFERENCE's own CAMERADataset produced for the same tiny synthetic NOCS-format data set
(tests/golden/nocs/, tests/golden/nocs_sample.npz; oracle/gen_golden_dataset.py imports and runs F/tools/dataset.py).  Runs
on the host: PNG decoding is libfpc_hip.so's host code (no GPU call)."""
import os
import pathlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = pathlib.Path(HERE) / "golden" / "nocs"


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "nocs_sample.npz"), allow_pickle=False)


def _dataset(root, classes, preprocessing=True):
    from fastposecnn_amd.tools import dataset as D
    pre = D.get_preprocessing(D.get_preprocessing_fn("resnet18", "imagenet")) if preprocessing else None
    return D.CAMERADataset(root, classes=classes, preprocessing=pre)


def test_directory_walk_and_class_filter(gold):
    classes = [str(c) for c in gold["classes"]]
    ds = _dataset(ROOT / "scene_a", classes)
    # frames without an instance of a wanted class are skipped; sub-directories are searched
    assert len(ds) == int(gold["n"]) == 2
    assert sorted(str(p.relative_to(ROOT)) for p in ds.images_fps) == sorted(str(p) for p in gold["paths"])
    assert ds.class_values_map == {0: 0, 1: 1, 6: 2} and ds.symmetric_classes == [1]


def test_sample_dict_matches_the_reference_key_by_key(gold):
    classes = [str(c) for c in gold["classes"]]
    ds = _dataset(ROOT / "scene_a", classes)
    order = {str(p.relative_to(ROOT)): i for i, p in enumerate(ds.images_fps)}
    for gi, rel in enumerate(str(p) for p in gold["paths"]):
        s = ds[order[rel]]
        assert set(s) == {"clean_image", "image", "mask", "depth", "path", "agg_data"}
        for k in ("clean_image", "image", "mask", "depth"):
            want = gold[f"s{gi}_{k}"]
            assert s[k].dtype == want.dtype and s[k].shape == want.shape, (k, s[k].dtype, want.dtype)
            assert np.array_equal(s[k], want), k                     # bit-exact incl. the float32 image
        want_keys = {k[len(f"s{gi}_agg_"):] for k in gold.files if k.startswith(f"s{gi}_agg_")}
        assert set(s["agg_data"]) == want_keys == {"class_ids", "symmetric_ids", "instance_masks", "quaternion", "scales", "xy", "z",
                                                   "T", "R", "RT"}
        for k in want_keys:
            want = gold[f"s{gi}_agg_{k}"]
            got = s["agg_data"][k]
            assert got.dtype == want.dtype and got.shape == want.shape, k
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-12, err_msg=k)      # same float64 operations
    # the laptop (class 5) of frame 0000 and the distractor (id 7) are gone; mug (6) is class 2 of the wanted list
    s0 = ds[order["scene_a/0000_color.png"]]
    assert s0["agg_data"]["class_ids"].tolist() == [1.0, 2.0] and s0["agg_data"]["symmetric_ids"].tolist() == [1.0, 0.0]
    assert set(np.unique(s0["mask"]).tolist()) == {0, 1, 2}


def test_sample_with_an_object_behind_the_camera_is_rejected(gold):
    assert bool(gold["scene_b_is_none"])
    ds = _dataset(ROOT / "scene_b", [str(c) for c in gold["classes"]], preprocessing=False)
    assert len(ds) == 1 and ds[0] is None


def test_collate_of_dataset_samples(gold):
    from fastposecnn_amd.tools import dataset as D
    ds = _dataset(ROOT / "scene_a", [str(c) for c in gold["classes"]])
    batch = D.my_collate_fn([ds[0], None, ds[1]])
    n0, n1 = (ds[i]["agg_data"]["class_ids"].shape[0] for i in range(2))
    assert tuple(batch["image"].shape) == (2, 3, 48, 64) and batch["agg_data"]["class_ids"].shape[0] == n0 + n1
    assert batch["agg_data"]["sample_ids"].tolist() == [0] * n0 + [1] * n1          # the None sample is dropped first
