"""Writes the synthetic PNG fixtures of tests/test_png_decode.py: tests/golden/png/*.png and png_expected.npz (the arrays
that were encoded: PNG is lossless, so they ARE the expected decode).  The files come from oracle/png_oracle.py's
encoder; every one is also decoded here by libpng itself (libpng16.so through its simplified API, what skimage / cv2
call underneath in the reference, F/tools/dataset.py:158-176) and must give the same bytes — that pins the fixtures to
the reference's decoder, not only to our own encoder.      python tests/golden/make_png_fixtures.py
"""
import ctypes
import ctypes.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import png_oracle as P          # noqa: E402


class PngImage(ctypes.Structure):          # png.h: png_image, version 1
    _fields_ = [("opaque", ctypes.c_void_p), ("version", ctypes.c_uint32), ("width", ctypes.c_uint32), ("height", ctypes.c_uint32),
                ("format", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("colormap_entries", ctypes.c_uint32),
                ("warning_or_error", ctypes.c_uint32), ("message", ctypes.c_char * 64)]


def libpng_decode(data, channels, sixteen):
    """libpng's own decode to GRAY / GA / RGB / RGBA, 8-bit (sRGB bytes as stored) — 16-bit files are compared through
    their 8-bit reduction only when the simplified API is asked for 8 bits, so they are checked by the high byte."""
    lib = ctypes.CDLL(ctypes.util.find_library("png16") or "libpng16.so.16")
    img = PngImage()
    img.version = 1
    buf = (ctypes.c_ubyte * len(data)).from_buffer_copy(data)
    assert lib.png_image_begin_read_from_memory(ctypes.byref(img), buf, ctypes.c_size_t(len(data))), img.message
    img.format = {1: 0, 2: 1, 3: 2, 4: 3}[channels]            # PNG_FORMAT_GRAY / GA / RGB / RGBA
    out = np.zeros((img.height, img.width, channels), np.uint8)
    assert lib.png_image_finish_read(ctypes.byref(img), None, out.ctypes.data_as(ctypes.c_void_p), 0, None), img.message
    return out


def main():
    rng = np.random.default_rng(2026)
    H, W = 48, 64
    yy, xx = np.mgrid[0:H, 0:W]
    smooth = ((np.sin(xx / 7.0) + np.cos(yy / 5.0)) * 60 + 128).astype(np.uint8)
    cases = {
        # what the NOCS files look like: colour RGB8, CAMERA masks RGBA8 (instance id in channel 0, 255 = background),
        # depth 16-bit grey, REAL depth as RGB8-encoded (G * 256 + R)
        "color_rgb": np.stack([smooth, smooth.T[:H, :W] if False else np.roll(smooth, 5, 1), rng.integers(0, 256, (H, W))], -1).astype(np.uint8),
        "color_noise_rgb": rng.integers(0, 256, (H, W, 3)).astype(np.uint8),
        "mask_rgba": np.stack([np.where((xx - 30) ** 2 + (yy - 20) ** 2 < 150, 3, 255)] * 3 + [np.full((H, W), 255)], -1).astype(np.uint8),
        "mask_grey": np.where((xx - 20) ** 2 + (yy - 25) ** 2 < 200, 7, 255).astype(np.uint8),
        "depth_u16": (rng.integers(300, 4000, (H, W)) + xx).astype(np.uint16),
        "depth_rgb_encoded": np.stack([rng.integers(0, 256, (H, W)), rng.integers(0, 16, (H, W)), np.zeros((H, W))], -1).astype(np.uint8),
        "grey_alpha": rng.integers(0, 256, (H, W, 2)).astype(np.uint8),
        "rgb16": rng.integers(0, 65536, (9, 13, 3)).astype(np.uint16),
        "one_pixel": np.array([[[1, 2, 3]]], np.uint8),
        "odd_size": rng.integers(0, 256, (7, 5, 3)).astype(np.uint8),
    }
    filters = {"color_noise_rgb": [4] * H, "mask_grey": [1] * H, "depth_u16": [y % 5 for y in range(H)], "one_pixel": [3]}
    expected = {}
    out_dir = os.path.join(HERE, "png")
    os.makedirs(out_dir, exist_ok=True)
    for name, arr in cases.items():
        data = P.encode(arr, filters=filters.get(name), idat_split=997 if name.startswith("color") else None)
        a3 = arr if arr.ndim == 3 else arr[:, :, None]
        got, _ = P.decode(data)
        assert np.array_equal(got, a3), name
        ref = libpng_decode(data, a3.shape[2], arr.dtype == np.uint16)
        want8 = (a3 >> 8).astype(np.uint8) if arr.dtype == np.uint16 else a3
        # libpng's 16 -> 8 reduction of the simplified API is (v * 255 + 32895) >> 16, not a plain high byte: compare 8-bit files
        if arr.dtype != np.uint16:
            assert np.array_equal(ref, want8), f"libpng disagrees on {name}"
        open(os.path.join(out_dir, name + ".png"), "wb").write(data)
        expected[name] = arr
    pal = rng.integers(0, 256, (11, 3)).astype(np.uint8)
    idx = rng.integers(0, 11, (10, 12)).astype(np.uint8)
    data = P.encode(idx, palette=pal)
    assert np.array_equal(libpng_decode(data, 3, False), pal[idx])
    open(os.path.join(out_dir, "palette.png"), "wb").write(data)
    expected["palette"] = pal[idx]
    np.savez_compressed(os.path.join(HERE, "png_expected.npz"), **expected)
    print("wrote", len(expected), "fixtures;", sum(os.path.getsize(os.path.join(out_dir, f)) for f in os.listdir(out_dir)), "bytes of PNG")


if __name__ == "__main__":
    main()
