#!/usr/bin/env python3
"""Generate tests/golden/nocs/ (a tiny synthetic NOCS-format data set) and tests/golden/nocs_sample.npz (what the REFERENCE's
own CAMERADataset returns for it).  Build container only (needs /root/reference); never on the GPU box.

Imported and executed unmodified from /root/reference/source_code/FastPoseCNN/tools: dataset.py (NOCSDataset /
CAMERADataset: get_image_paths_in_dir, remove_empty_samples, __getitem__, generate_agg_data), json_tools.py,
data_manipulation.py (extract_xyz_R_T_from_RTs and what it calls), project.py, transforms/.  Third-party modules absent from
this image are stood in for by the minimum the executed lines touch:
  * skimage.io.imread / cv2.imread return the arrays the fixture files were ENCODED from (the decoder is pinned separately:
    tests/test_png_decode.py against libpng); cv2's B,G,R channel order for the 3-channel depth file is reproduced;
  * skimage.img_as_float32 on a float array = astype(float32);
  * albumentations.Compose / Lambda: apply the `image=` function to the 'image' key (and declared additional targets), pass
    every other key through — what albumentations does for keys it has no target for;
  * segmentation_models_pytorch's preprocess_input is not importable: the sample is preprocessed with its published
    algorithm as restated in oracle/preprocess.py (PARITY UNPINNED for that one function, as everywhere in this repository);
  * draw / visualize (plotting), setup_env (dotenv), matplotlib, torchvision, tensorboard, imutils, pyquaternion, easydict:
    empty stand-ins, none of their code runs.

Usage: python oracle/gen_golden_dataset.py
"""
import functools
import json
import os
import pathlib
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/source_code/FastPoseCNN"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

H, W = 48, 64
CLASSES = ['bg', 'bottle', 'mug']            # 'laptop' instances are dropped, 'mug' (6) is renumbered to 2


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def rigid(rng, tz):
    """RT whose INVERSE is [R | t] with t_z = tz > 0 (F/tools/data_manipulation.py:999-1003 reads z from the inverse)."""
    a = rng.normal(size=(3, 3))
    q, _ = np.linalg.qr(a)
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    M = np.eye(4)
    M[:3, :3] = q
    M[:3, 3] = [rng.uniform(-0.2, 0.2), rng.uniform(-0.15, 0.15), tz]
    return np.linalg.inv(M)


def make_frame(rng, instances, distractor=None):
    """instances: [(mask id, CAMERA class id, (cy, cx, ry, rx))]; returns the arrays of the four files"""
    yy, xx = np.mgrid[0:H, 0:W]
    color = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    ids = np.full((H, W), 255, np.uint8)
    for mid, _, (cy, cx, ry, rx) in instances:
        ids[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = mid
    if distractor is not None:
        mid, (cy, cx, ry, rx) = distractor
        ids[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = mid
    mask = np.stack([ids, ids, ids, np.full_like(ids, 255)], -1)                    # CAMERA masks: RGBA, ids in channel 0
    d16 = rng.integers(300, 3000, (H, W)).astype(np.uint16)
    depth = np.stack([(d16 & 255).astype(np.uint8), (d16 >> 8).astype(np.uint8), np.zeros((H, W), np.uint8)], -1)   # R = low, G = high byte
    n = len(instances)
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    meta = {
        "instance_dict": {str(mid): int(cls) for mid, cls, _ in instances},
        "scales": rng.uniform(0.1, 0.6, (n, 3)).tolist(),
        "quaternions": q.tolist(),
        "RTs": [rigid(rng, rng.uniform(0.6, 1.8)).tolist() for _ in range(n)],
        "norm_factors": rng.uniform(0.3, 1.2, (n,)).tolist(),
    }
    return color, mask, depth, d16, meta


def main():
    from oracle import png_oracle, preprocess as opre
    rng = np.random.default_rng(20260404)
    root = pathlib.Path(OUT) / "nocs"
    frames = {
        root / "scene_a" / "0000": make_frame(rng, [(1, 1, (14, 16, 9, 7)), (2, 5, (30, 40, 10, 12)), (3, 6, (20, 52, 8, 6))], distractor=(7, (40, 10, 5, 6))),
        root / "scene_a" / "0001": make_frame(rng, [(1, 5, (20, 20, 8, 8)), (2, 3, (30, 45, 7, 9))]),          # no wanted class: skipped
        root / "scene_a" / "deeper" / "0002": make_frame(rng, [(4, 6, (24, 30, 12, 14))]),
    }
    arrays = {}
    for stem, (color, mask, depth, d16, meta) in frames.items():
        stem.parent.mkdir(parents=True, exist_ok=True)
        for suffix, arr in (("_color.png", color), ("_mask.png", mask), ("_depth.png", depth)):
            fp = str(stem) + suffix
            with open(fp, "wb") as f:
                f.write(png_oracle.encode(arr, filters=[0] * H, level=9))
            arrays[fp] = arr
        with open(str(stem) + "_meta+.json", "w") as f:
            json.dump(meta, f)

    # ---- the reference's dataset module, third-party imports stood in for
    import torch

    def sk_imread(path, *a, **k):
        return arrays[str(path)].copy()

    def cv_imread(path, flag=None):
        a = arrays[str(path)]
        # cv2 hands 3-channel files over as B, G, R.  As int32: numpy >= 2 (NEP 50) refuses `uint8_array * 256`
        # (data_manipulation.py:157), which the numpy 1.x the reference was written for evaluated in a wider type
        return a[:, :, ::-1].astype(np.int32) if a.ndim == 3 else a.copy()

    sk = _stub("skimage", img_as_float32=lambda a: np.asarray(a).astype(np.float32))
    sk.io = _stub("skimage.io", imread=sk_imread)
    sk.transform = _stub("skimage.transform")
    _stub("cv2", imread=cv_imread)
    _stub("imutils")
    _stub("pyquaternion", Quaternion=object)
    _stub("easydict", EasyDict=type("EasyDict", (dict,), {"__getattr__": dict.get, "__setattr__": dict.__setitem__}))
    mpl = _stub("matplotlib"); mpl.pyplot = _stub("matplotlib.pyplot"); mpl.cm = _stub("matplotlib.cm", get_cmap=lambda *a, **k: (lambda x: (0.0, 0.0, 0.0, 1.0)))      # colour maps: plotting only
    tv = _stub("torchvision"); tv.transforms = _stub("torchvision.transforms"); tv.transforms.functional = _stub("torchvision.transforms.functional")
    _stub("torch.utils.tensorboard")
    _stub("segmentation_models_pytorch")
    _stub("pytorch_lightning", LightningDataModule=object)
    _stub("setup_env")
    _stub("draw"); _stub("visualize")
    _stub("numpy.lib.arraysetops", isin=np.isin)

    class Lambda:
        def __init__(self, image=None, mask=None, **k):
            self.image = image

    class Compose:
        def __init__(self, transforms, additional_targets=None, **k):
            self.transforms, self.extra = transforms, dict(additional_targets or {})

        def __call__(self, **data):
            out = dict(data)
            for t in self.transforms:
                for key in list(out):
                    if (key == "image" or self.extra.get(key) == "image") and t.image is not None and out[key] is not None:
                        out[key] = t.image(out[key])
            return out

    class Unused:                       # any other albumentations transform named in a default argument / module-level table
        def __init__(self, *a, **k):
            pass

    albu = _stub("albumentations", Compose=Compose, Lambda=Lambda)
    albu.__getattr__ = lambda name: Unused
    albu.pytorch = _stub("albumentations.pytorch", ToTensor=object)

    os.environ["TOOLS_DIR"] = REF + "/tools"
    sys.path.insert(0, REF + "/tools")
    try:
        import dataset as refds
    except Exception as e:
        raise SystemExit(f"the reference's tools/dataset.py did not import: {e!r}")
    import transforms as reft

    from fastposecnn_amd.tools.dataset import IMAGENET_PARAMS
    pre_fn = functools.partial(opre.smp_preprocess_input, **IMAGENET_PARAMS)
    ds = refds.CAMERADataset(root, classes=CLASSES, preprocessing=reft.pose.get_preprocessing(pre_fn))
    paths = [str(p.relative_to(root)) for p in ds.images_fps]
    out = {"n": np.array(len(ds)), "paths": np.array(paths), "classes": np.array(CLASSES)}
    for i in range(len(ds)):
        s = ds[i]
        for k in ("clean_image", "image", "mask", "depth"):
            out[f"s{i}_{k}"] = np.asarray(s[k])
        for k, v in s["agg_data"].items():
            out[f"s{i}_agg_{k}"] = np.asarray(v)
    # the z <= 0 rejection: the same frame with one transform flipped behind the camera
    stem = root / "scene_b" / "0003"
    color, mask, depth, d16, meta = make_frame(rng, [(1, 1, (14, 16, 9, 7)), (2, 6, (30, 40, 10, 12))])
    M = np.linalg.inv(np.array(meta["RTs"][1])); M[2, 3] = -0.4; meta["RTs"][1] = np.linalg.inv(M).tolist()
    stem.parent.mkdir(parents=True, exist_ok=True)
    for suffix, arr in (("_color.png", color), ("_mask.png", mask), ("_depth.png", depth)):
        fp = str(stem) + suffix
        with open(fp, "wb") as f:
            f.write(png_oracle.encode(arr, filters=[0] * H, level=9))
        arrays[fp] = arr
    with open(str(stem) + "_meta+.json", "w") as f:
        json.dump(meta, f)
    ds_b = refds.CAMERADataset(root / "scene_b", classes=CLASSES, preprocessing=None)
    assert len(ds_b) == 1 and ds_b[0] is None
    out["scene_b_is_none"] = np.array(True)
    np.savez_compressed(os.path.join(OUT, "nocs_sample.npz"), **out)
    print("wrote", os.path.join(OUT, "nocs_sample.npz"), "with", len(ds), "samples:", paths)


if __name__ == "__main__":
    main()
