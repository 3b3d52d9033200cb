"""PNG decode / encode in plain Python (zlib + numpy): the CHECKER of fastposecnn_amd/csrc/png_decode.hip and the writer
of the synthetic fixtures under tests/golden/png/.  TEST INFRASTRUCTURE ONLY (tests/, bench.py's fixture set-up).

The reference reads frames with skimage.io.imread / cv2.imread (F/tools/dataset.py:158-176), i.e. libpng: the format
is the published PNG specification (ISO/IEC 15948): 8-byte signature, IHDR, IDAT chunks = one zlib stream of scanlines,
each prefixed by a filter type 0-4 (None, Sub, Up, Average, Paeth) over bytes-per-pixel strides, IEND.  Colour types
0 (grey), 2 (RGB), 3 (palette), 4 (grey + alpha), 6 (RGBA), bit depths 8 and 16 (big-endian samples), no interlace.
"""
import struct
import zlib

import numpy as np

SIG = b"\x89PNG\r\n\x1a\n"
CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def _chunks(data):
    assert data[:8] == SIG, "not a PNG"
    o = 8
    while o < len(data):
        n, typ = struct.unpack(">I4s", data[o:o + 8])
        body = data[o + 8:o + 8 + n]
        crc, = struct.unpack(">I", data[o + 8 + n:o + 12 + n])
        assert zlib.crc32(typ + body) & 0xffffffff == crc, "bad CRC"
        yield typ, body
        o += 12 + n


def _paeth(a, b, c):
    p = a.astype(np.int32) + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))


def decode(data):
    """bytes -> (array [H, W, C] uint8 or uint16 as stored — palette expanded to RGB), info dict."""
    ihdr, idat, plte = None, [], None
    for typ, body in _chunks(data):
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
    W, H, depth, ctype, _, _, interlace = ihdr
    assert interlace == 0 and depth in (8, 16) and ctype in CHANNELS
    C = CHANNELS[ctype]
    bpp = C * depth // 8
    stride = W * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(H, stride + 1)
    out = np.zeros((H, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(H):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        cur = np.zeros(stride, np.int32)
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:                                   # Sub / Average / Paeth depend on the pixel to the left: bpp-strided columns
            for x in range(0, stride, bpp):
                a = cur[x - bpp:x] if x else np.zeros(bpp, np.int32)
                b = prev[x:x + bpp]
                c = prev[x - bpp:x] if x else np.zeros(bpp, np.int32)
                pred = a if f == 1 else ((a + b) >> 1 if f == 3 else _paeth(a, b, c))
                cur[x:x + bpp] = (line[x:x + bpp] + pred) & 255
        out[y] = cur
        prev = cur
    if depth == 16:
        arr = out.reshape(H, W, C, 2).astype(np.uint16)
        arr = (arr[..., 0] << 8) | arr[..., 1]
    else:
        arr = out.reshape(H, W, C)
    if ctype == 3:
        arr = plte[arr[..., 0]]
    return arr, {"width": W, "height": H, "bit_depth": depth, "color_type": ctype, "channels": arr.shape[2]}


def encode(arr, filters=None, level=6, idat_split=None, palette=None):
    """[H, W, C] (or [H, W]) uint8 / uint16 -> PNG bytes.  `filters`: per-row filter types (default: cycle 0..4 so that
    every unfilter path is exercised); `idat_split`: cut the zlib stream into IDAT chunks of this many bytes;
    `palette` [n,3] uint8: write colour type 3 with arr as indices."""
    arr = np.asarray(arr)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    H, W, C = arr.shape
    depth = 16 if arr.dtype == np.uint16 else 8
    ctype = 3 if palette is not None else {1: 0, 2: 4, 3: 2, 4: 6}[C]
    if depth == 16:
        rows = np.stack([(arr >> 8).astype(np.uint8), (arr & 255).astype(np.uint8)], -1).reshape(H, -1)
    else:
        rows = arr.astype(np.uint8).reshape(H, -1)
    bpp = C * depth // 8
    stride = rows.shape[1]
    filters = [y % 5 for y in range(H)] if filters is None else list(filters)
    body = bytearray()
    prev = np.zeros(stride, np.int32)
    for y in range(H):
        f, cur = filters[y], rows[y].astype(np.int32)
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        pred = {0: 0, 1: a, 2: prev, 3: (a + prev) >> 1, 4: _paeth(a, prev, c)}[f]
        body.append(f)
        body += ((cur - pred) & 255).astype(np.uint8).tobytes()
        prev = cur
    z = zlib.compress(bytes(body), level)

    def chunk(typ, payload):
        return struct.pack(">I", len(payload)) + typ + payload + struct.pack(">I", zlib.crc32(typ + payload) & 0xffffffff)

    out = SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, ctype, 0, 0, 0))
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    out += chunk(b"tEXt", b"Comment\x00fastposecnn_amd synthetic fixture")          # an ancillary chunk decoders must skip
    step = idat_split or len(z)
    for o in range(0, len(z), step):
        out += chunk(b"IDAT", z[o:o + step])
    return out + chunk(b"IEND", b"")
