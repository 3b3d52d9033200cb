"""Input side: F/tools/dataset.py:249-262 on the device, and the collate mirror."""
import numpy as np
import pytest
import torch

from oracle import preprocess as opre


def _frames(rng, B, H, W, kind):
    if kind == "random":
        return rng.integers(0, 256, size=(B, H, W, 3), dtype=np.uint8)
    if kind == "dark":                      # all bytes <= 1: smp's "x.max() > 1" rule leaves the frame unscaled
        return rng.integers(0, 2, size=(B, H, W, 3), dtype=np.uint8)
    if kind == "constant":
        return np.full((B, H, W, 3), 117, dtype=np.uint8)
    if kind == "narrow":                    # the extreme lies at the low end of one channel only
        x = rng.integers(100, 140, size=(B, H, W, 3), dtype=np.uint8)
        x[..., 2] = rng.integers(0, 30, size=(B, H, W), dtype=np.uint8)
        return x
    if kind == "mixed":                     # per-image statistics: frame 0 dark, frame 1 bright
        x = rng.integers(0, 256, size=(B, H, W, 3), dtype=np.uint8)
        x[0] = x[0] % 2
        return x
    raise KeyError(kind)


def test_oracle_preprocess_properties():
    from fastposecnn_amd.tools.dataset import IMAGENET_PARAMS
    rng = np.random.default_rng(0)
    x = _frames(rng, 1, 12, 20, "random")[0]
    y = opre.preprocess_frame(x, IMAGENET_PARAMS)
    assert y.dtype == np.float32 and y.shape == (3, 12, 20)
    assert np.max(np.abs(y)) == np.float32(1.0)
    # hand-derived: one pixel, bytes (255, 0, 51): ((1 - .485)/.229, -.456/.224, (.2 - .406)/.225) / max|.|
    one = np.array([[[255, 0, 51]]], dtype=np.uint8)
    v = np.array([(1.0 - 0.485) / 0.229, (0.0 - 0.456) / 0.224, (51 / 255.0 - 0.406) / 0.225])
    np.testing.assert_array_equal(opre.preprocess_frame(one, IMAGENET_PARAMS).ravel(), (v / np.abs(v).max()).astype(np.float32))
    # all bytes <= 1: no / 255
    dark = np.array([[[1, 0, 1]]], dtype=np.uint8)
    v = np.array([(1.0 - 0.485) / 0.229, (0.0 - 0.456) / 0.224, (1.0 - 0.406) / 0.225])
    np.testing.assert_array_equal(opre.preprocess_frame(dark, IMAGENET_PARAMS).ravel(), (v / np.abs(v).max()).astype(np.float32))


def test_collate_mirror():
    from fastposecnn_amd.tools.dataset import my_collate_fn
    def sample(n, fill):
        return {"image": np.full((3, 4, 4), fill, np.float32), "mask": np.zeros((4, 4), np.int64), "path": "p%d" % fill,
                "agg_data": {"class_ids": np.arange(n), "z": np.full((n, 1), fill, np.float32)}}
    out = my_collate_fn([sample(2, 1), None, sample(0, 2), sample(3, 3)])
    assert out["image"].shape == (3, 3, 4, 4) and out["mask"].dtype == torch.int64
    assert out["path"] == ["p1", "p2", "p3"]
    assert out["agg_data"]["sample_ids"].tolist() == [0, 0, 2, 2, 2]
    assert out["agg_data"]["class_ids"].tolist() == [0, 1, 0, 1, 2]
    assert out["agg_data"]["z"].shape == (5, 1)
    assert my_collate_fn([None, None]) is None


def test_preprocessing_params():
    from fastposecnn_amd.tools.dataset import get_preprocessing_params
    p = get_preprocessing_params("resnet34")
    assert p["mean"] == [0.485, 0.456, 0.406] and p["std"] == [0.229, 0.224, 0.225] and p["input_range"] == [0, 1]
    with pytest.raises(ValueError):
        get_preprocessing_params("resnet18", pretrained=None)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["random", "dark", "constant", "narrow", "mixed"])
@pytest.mark.parametrize("shape", [(2, 480, 640), (1, 37, 53), (3, 8, 16)])
def test_preprocess_bit_exact(kind, shape):
    from fastposecnn_amd.tools.dataset import IMAGENET_PARAMS, preprocess_frames
    B, H, W = shape
    x = _frames(np.random.default_rng(B * H + W), B, H, W, kind)
    want = np.stack([opre.preprocess_frame(f, IMAGENET_PARAMS) for f in x])
    got = preprocess_frames(torch.from_numpy(x).cuda()).cpu().numpy()
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_preprocess_single_frame_and_errors():
    from fastposecnn_amd.tools.dataset import IMAGENET_PARAMS, preprocess_frames
    x = _frames(np.random.default_rng(5), 1, 30, 44, "random")[0]
    got = preprocess_frames(torch.from_numpy(x).cuda())
    assert got.shape == (3, 30, 44)
    np.testing.assert_array_equal(got.cpu().numpy(), opre.preprocess_frame(x, IMAGENET_PARAMS))
    with pytest.raises(TypeError):
        preprocess_frames(torch.zeros((1, 4, 4, 3), device="cuda"))
    with pytest.raises(ValueError):
        preprocess_frames(torch.zeros((1, 4, 4, 4), dtype=torch.uint8, device="cuda"))
    with pytest.raises(RuntimeError):
        preprocess_frames(torch.zeros((1, 4, 4, 3), dtype=torch.uint8))


@pytest.mark.gpu
def test_frame_uploader_matches_direct():
    from fastposecnn_amd.tools.dataset import IMAGENET_PARAMS, FrameUploader
    rng = np.random.default_rng(9)
    up = FrameUploader(2, 64, 96)
    batches = [_frames(rng, 2, 64, 96, "random") for _ in range(5)]
    for b in batches:
        t, ev = up.upload(b)
        torch.cuda.current_stream().wait_event(ev)
        got = t.clone()
        want = np.stack([opre.preprocess_frame(f, IMAGENET_PARAMS) for f in b])
        np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.gpu
def test_uploader_from_png_files_equals_preprocessing_of_the_decoded_bytes():
    """`*_color.png` bytes -> native decode into the pinned slot -> H2D -> fpc_preprocess_u8 == the numpy chain of
    F/tools/dataset.py:249-262 on the frames the files encode (fixtures pinned by libpng: tests/test_png_decode.py)."""
    import os
    from conftest import GOLDEN, load_golden
    from fastposecnn_amd.tools.dataset import IMAGENET_PARAMS, FrameUploader
    exp = load_golden("png_expected.npz")
    names = ["color_rgb", "color_noise_rgb", "mask_rgba"]                 # 48 x 64; the RGBA file loses its alpha
    files = [open(os.path.join(GOLDEN, "png", n + ".png"), "rb").read() for n in names]
    up = FrameUploader(3, 48, 64)
    for threads in (1, 3):
        t, ev = up.upload_png(files, threads=threads)
        torch.cuda.current_stream().wait_event(ev)
        want = np.stack([opre.preprocess_frame(exp[n][:, :, :3], IMAGENET_PARAMS) for n in names])
        np.testing.assert_array_equal(t.cpu().numpy(), want)
    with pytest.raises(ValueError):
        up.upload_png(files[:2])
