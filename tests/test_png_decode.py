"""Frame decoding (F/tools/dataset.py:158-176): the native PNG reader (csrc/png_decode.hip, host
code over zlib) and the pure-Python oracle against the committed fixtures — files written by
tests/golden/make_png_fixtures.py, where libpng itself (what skimage / cv2 call in the reference) decoded every one of
them to the same bytes.  No GPU needed: the library only has to load."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden

PNG_DIR = os.path.join(GOLDEN, "png")


@pytest.fixture(scope="module")
def ds():
    import fastposecnn_amd.lib          # noqa: F401  (puts the drop-in modules on sys.path)
    from fastposecnn_amd.tools import dataset
    return dataset


def _files():
    return sorted(f[:-4] for f in os.listdir(PNG_DIR) if f.endswith(".png"))


def test_fixture_set_is_complete():
    exp = load_golden("png_expected.npz")
    assert _files() == sorted(exp) and len(exp) >= 10


@pytest.mark.parametrize("name", _files())
def test_native_and_oracle_decode_equal_the_encoded_array(ds, name):
    from oracle import png_oracle
    want = load_golden("png_expected.npz")[name]
    data = open(os.path.join(PNG_DIR, name + ".png"), "rb").read()
    got = ds.imread_png(data)
    assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want)
    assert np.array_equal(ds.imread_png(os.path.join(PNG_DIR, name + ".png")), want)          # by path, as imread is called
    ora, info = png_oracle.decode(data)
    assert np.array_equal(ora.reshape(want.shape), want) and info["width"] == want.shape[1]
    rgb = ds.imread_png(data, rgb8=True)                                                          # the uploader's form
    w3 = want if want.ndim == 3 else want[:, :, None]
    w8 = (w3 >> 8).astype(np.uint8) if w3.dtype == np.uint16 else w3
    ref = np.repeat(w8[:, :, :1], 3, 2) if w8.shape[2] <= 2 else w8[:, :, :3]
    assert rgb.dtype == np.uint8 and np.array_equal(rgb, ref)


def test_damaged_and_unsupported_files_are_refused(ds):
    from oracle import png_oracle
    good = open(os.path.join(PNG_DIR, "color_rgb.png"), "rb").read()
    cases = {"signature": b"\x00" + good[1:], "truncated": good[:len(good) // 2], "empty": b"", "crc": good[:100] + bytes([good[100] ^ 1]) + good[101:]}
    interlaced = bytearray(png_oracle.encode(np.zeros((4, 4, 3), np.uint8)))
    interlaced[28] = 1                                                                            # IHDR interlace method (CRC now wrong too)
    cases["interlaced"] = bytes(interlaced)
    for what, data in cases.items():
        with pytest.raises(RuntimeError, match="PNG|invalid"):
            ds.imread_png(data)


def test_batch_decode_fills_the_staging_layout(ds):
    import ctypes
    from fastposecnn_amd import _native as nat
    exp = load_golden("png_expected.npz")
    names = ["color_rgb", "color_noise_rgb", "depth_rgb_encoded", "mask_rgba"]                   # all 48 x 64
    files = [np.frombuffer(open(os.path.join(PNG_DIR, n + ".png"), "rb").read(), np.uint8) for n in names] * 3
    B = len(files)
    out = np.full((B, 48, 64, 3), 7, np.uint8)
    ptrs = (ctypes.c_void_p * B)(*[f.ctypes.data for f in files])
    sizes = (ctypes.c_size_t * B)(*[f.size for f in files])
    for threads in (1, 4):
        out[:] = 7
        nat.check(nat.lib().fpc_png_decode_batch(ptrs, sizes, B, out.ctypes.data, 48, 64, threads), "batch")
        for i in range(B):
            assert np.array_equal(out[i], exp[names[i % 4]][:, :, :3]), (threads, i)
    rc = nat.lib().fpc_png_decode_batch(ptrs, sizes, B, out.ctypes.data, 48, 60, 2)              # wrong size: nothing is decoded
    assert rc == -1


def test_frame_files_follow_the_reference_item(ds, tmp_path):
    """F/tools/dataset.py:158-176: mask = first channel of the CAMERA RGBA file as float with 255 -> 0; depth = the
    16-bit grey file, or G * 256 + R of an RGB-encoded one (data_manipulation.standardize_depth on cv2's BGR)."""
    exp = load_golden("png_expected.npz")
    import shutil
    for src, dst in (("color_rgb", "0001_color.png"), ("mask_rgba", "0001_mask.png"), ("depth_u16", "0001_depth.png"),
                     ("color_noise_rgb", "0002_color.png"), ("mask_grey", "0002_mask.png"), ("depth_rgb_encoded", "0002_depth.png"),
                     ("odd_size", "0003_color.png")):
        shutil.copy(os.path.join(PNG_DIR, src + ".png"), tmp_path / dst)
    a = ds.read_frame_files(tmp_path / "0001_color.png", camera=True)
    assert np.array_equal(a["image"], exp["color_rgb"]) and a["mask"].dtype == np.float64
    m = exp["mask_rgba"][:, :, 0].astype(float); m[m == 255] = 0
    assert np.array_equal(a["mask"], m) and set(np.unique(a["mask"])) == {0.0, 3.0}
    assert a["depth"].dtype == np.uint16 and np.array_equal(a["depth"], exp["depth_u16"])
    b = ds.read_frame_files(tmp_path / "0002_color.png", camera=False)
    e = exp["depth_rgb_encoded"]
    assert np.array_equal(b["depth"], e[:, :, 1].astype(np.uint16) * 256 + e[:, :, 0]) and set(np.unique(b["mask"])) == {0.0, 7.0}
    c = ds.read_frame_files(tmp_path / "0003_color.png")
    assert set(c) == {"image"}


def test_prefetcher_yields_every_batch_in_order(ds):
    from oracle import png_oracle
    rng = np.random.default_rng(4)
    frames = [rng.integers(0, 256, (16, 24, 3)).astype(np.uint8) for _ in range(7)]
    files = [png_oracle.encode(f) for f in frames]
    pre = ds.PngFramePrefetcher(lambda k: [files[(2 * k) % 7], files[(2 * k + 1) % 7]], 9, 2, 16, 24, workers=3, ahead=4)
    got = list(pre)
    assert len(got) == 9
    for k, b in enumerate(got):
        assert b.shape == (2, 16, 24, 3) and np.array_equal(b[0], frames[(2 * k) % 7]) and np.array_equal(b[1], frames[(2 * k + 1) % 7])
