// png_decode.hip — host-side PNG decoding for the input side of the hot path: the reference's
// dataset item reads `*_color.png`, `*_mask.png` and `*_depth.png` with skimage.io.imread / cv2.imread, i.e. libpng
// (F/tools/dataset.py:158-176).  libpng's headers are not in this image; zlib's are, so the container format is read
// here (ISO/IEC 15948): signature, IHDR, the IDAT chunks' single zlib stream inflated chunk by chunk, scanlines
// un-filtered in place (None, Sub, Up, Average, Paeth).  Colour types 0 / 2 / 3 / 4 / 6 at 8 or 16 bits, no interlace
// (the NOCS files are plain 8-bit RGB(A) colour / mask and 16-bit grey depth).  No GPU code: `fpc_png_decode_batch`
// fills the pinned staging slots of tools/dataset.py's FrameUploader from a small pool of host threads, the decoded
// bytes then take the existing upload + fpc_preprocess_u8 path.
#include <string.h>
#include <zlib.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../include/fpc.h"

namespace {

struct PngHeader { uint32_t w, h; int depth, ctype, channels, bpp; };

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};

// IHDR of a PNG in memory; FPC_OK, or FPC_EFORMAT for anything this decoder does not read
int parse_header(const uint8_t* d, size_t n, PngHeader& h) {
    if (n < 8 + 25 || memcmp(d, kSig, 8) != 0) return FPC_EFORMAT;
    if (be32(d + 8) != 13 || memcmp(d + 12, "IHDR", 4) != 0) return FPC_EFORMAT;
    const uint8_t* p = d + 16;
    h.w = be32(p); h.h = be32(p + 4); h.depth = p[8]; h.ctype = p[9];
    const int compression = p[10], filter = p[11], interlace = p[12];
    if (h.w == 0 || h.h == 0 || h.w > 65535 || h.h > 65535) return FPC_EFORMAT;
    if (compression != 0 || filter != 0 || interlace != 0) return FPC_EFORMAT;      // Adam7 files are not produced by the dataset tools
    if (h.depth != 8 && h.depth != 16) return FPC_EFORMAT;
    switch (h.ctype) {
        case 0: h.channels = 1; break;
        case 2: h.channels = 3; break;
        case 3: h.channels = 1; if (h.depth != 8) return FPC_EFORMAT; break;
        case 4: h.channels = 2; break;
        case 6: h.channels = 4; break;
        default: return FPC_EFORMAT;
    }
    h.bpp = h.channels * h.depth / 8;
    return FPC_OK;
}

inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// scanlines [h][1 + stride] (filter byte first) -> pixel bytes, in place in the rows' own storage
int unfilter(uint8_t* raw, const PngHeader& h) {
    const size_t stride = (size_t)h.w * h.bpp;
    const int bpp = h.bpp;
    const uint8_t* prev = nullptr;
    for (uint32_t y = 0; y < h.h; ++y) {
        uint8_t* line = raw + (size_t)y * (stride + 1);
        uint8_t* cur = line + 1;
        switch (line[0]) {
            case 0: break;
            case 1:
                for (size_t x = bpp; x < stride; ++x) cur[x] = (uint8_t)(cur[x] + cur[x - bpp]);
                break;
            case 2:
                if (prev) for (size_t x = 0; x < stride; ++x) cur[x] = (uint8_t)(cur[x] + prev[x]);
                break;
            case 3:
                for (size_t x = 0; x < stride; ++x) {
                    const int a = x >= (size_t)bpp ? cur[x - bpp] : 0, b = prev ? prev[x] : 0;
                    cur[x] = (uint8_t)(cur[x] + ((a + b) >> 1));
                }
                break;
            case 4:
                for (size_t x = 0; x < stride; ++x) {
                    const int a = x >= (size_t)bpp ? cur[x - bpp] : 0, b = prev ? prev[x] : 0;
                    const int c = (prev && x >= (size_t)bpp) ? prev[x - bpp] : 0;
                    cur[x] = (uint8_t)(cur[x] + paeth(a, b, c));
                }
                break;
            default: return FPC_EFORMAT;
        }
        prev = cur;
    }
    return FPC_OK;
}

// The whole file -> un-filtered scanlines in `raw` (h x (1 + stride) bytes) and the palette (if any)
int inflate_scanlines(const uint8_t* d, size_t n, const PngHeader& h, std::vector<uint8_t>& raw, uint8_t* palette /*[768]*/,
                      int* palette_n) {
    const size_t stride = (size_t)h.w * h.bpp;
    // the header alone must not size a multi-gigabyte buffer (a 60-byte file can claim 65535 x 65535 x 8): frames of this
    // path are megabytes; beyond 1 GiB of scanlines the file is refused (also keeps zlib's 32-bit avail_out exact)
    if ((size_t)h.h * (stride + 1) > ((size_t)1 << 30)) return FPC_EFORMAT;
    raw.resize((size_t)h.h * (stride + 1));
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) return FPC_EFORMAT;
    zs.next_out = raw.data();
    zs.avail_out = (uInt)raw.size();
    size_t o = 8;
    bool done = false, end = false;
    int rc = FPC_OK;
    *palette_n = 0;
    while (o + 12 <= n && !end) {
        const uint32_t len = be32(d + o);
        const uint8_t* typ = d + o + 4;
        if ((size_t)len > n - o - 12) { rc = FPC_EFORMAT; break; }
        const uint8_t* body = d + o + 8;
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), typ, 4 + len) != be32(body + len)) { rc = FPC_EFORMAT; break; }
        if (memcmp(typ, "IDAT", 4) == 0 && !done) {
            zs.next_in = const_cast<Bytef*>(body);
            zs.avail_in = len;
            const int zr = inflate(&zs, Z_NO_FLUSH);
            if (zr == Z_STREAM_END) done = true;
            else if (zr != Z_OK && zr != Z_BUF_ERROR) { rc = FPC_EFORMAT; break; }
        } else if (memcmp(typ, "PLTE", 4) == 0) {
            if (len % 3 != 0 || len > 768) { rc = FPC_EFORMAT; break; }
            memcpy(palette, body, len);
            *palette_n = (int)(len / 3);
        } else if (memcmp(typ, "IEND", 4) == 0) {
            end = true;
        }                                           // every other chunk (tEXt, gAMA, pHYs, ...) is skipped, as imread does
        o += 12 + (size_t)len;
    }
    inflateEnd(&zs);
    if (rc != FPC_OK) return rc;
    if (zs.avail_out != 0) return FPC_EFORMAT;                                        // truncated image data
    if (h.ctype == 3 && *palette_n == 0) return FPC_EFORMAT;
    return unfilter(raw.data(), h);
}

// mode 0: samples as stored ([H, W, C] u8, or u16 in HOST byte order for 16-bit files; a palette is expanded to RGB8);
// mode 3: 8-bit RGB whatever the file holds (grey replicated, alpha dropped, 16-bit samples by their high byte)
int decode_one_unguarded(const uint8_t* d, size_t n, void* out, size_t out_bytes, int mode);

// no exception may cross the C boundary (or unwind out of a worker thread): an allocation failure is a format error here
int decode_one(const uint8_t* d, size_t n, void* out, size_t out_bytes, int mode) {
    try {
        return decode_one_unguarded(d, n, out, out_bytes, mode);
    } catch (...) {
        return FPC_EFORMAT;
    }
}

int decode_one_unguarded(const uint8_t* d, size_t n, void* out, size_t out_bytes, int mode) {
    PngHeader h;
    int rc = parse_header(d, n, h);
    if (rc != FPC_OK) return rc;
    std::vector<uint8_t> raw;
    uint8_t palette[768];
    int pn = 0;
    rc = inflate_scanlines(d, n, h, raw, palette, &pn);
    if (rc != FPC_OK) return rc;
    const size_t stride = (size_t)h.w * h.bpp, px = (size_t)h.w * h.h;
    const int sb = h.depth / 8;                                                       // bytes per sample
    if (mode == 3) {
        if (out_bytes < px * 3) return FPC_EINVAL;
        uint8_t* o = (uint8_t*)out;
        for (uint32_t y = 0; y < h.h; ++y) {
            const uint8_t* s = raw.data() + (size_t)y * (stride + 1) + 1;
            for (uint32_t x = 0; x < h.w; ++x, s += h.bpp, o += 3) {
                if (h.ctype == 3) { const int i = s[0] < pn ? s[0] : 0; o[0] = palette[3 * i]; o[1] = palette[3 * i + 1]; o[2] = palette[3 * i + 2]; }
                else if (h.channels <= 2) { o[0] = o[1] = o[2] = s[0]; }
                else { o[0] = s[0]; o[1] = s[sb]; o[2] = s[2 * sb]; }
            }
        }
        return FPC_OK;
    }
    if (mode != 0) return FPC_EINVAL;
    if (h.ctype == 3) {
        if (out_bytes < px * 3) return FPC_EINVAL;
        uint8_t* o = (uint8_t*)out;
        for (uint32_t y = 0; y < h.h; ++y) {
            const uint8_t* s = raw.data() + (size_t)y * (stride + 1) + 1;
            for (uint32_t x = 0; x < h.w; ++x, o += 3) { const int i = s[x] < pn ? s[x] : 0; memcpy(o, palette + 3 * i, 3); }
        }
        return FPC_OK;
    }
    if (out_bytes < px * (size_t)h.bpp) return FPC_EINVAL;
    if (h.depth == 8) {
        uint8_t* o = (uint8_t*)out;
        for (uint32_t y = 0; y < h.h; ++y) memcpy(o + (size_t)y * stride, raw.data() + (size_t)y * (stride + 1) + 1, stride);
    } else {
        uint16_t* o = (uint16_t*)out;
        for (uint32_t y = 0; y < h.h; ++y) {
            const uint8_t* s = raw.data() + (size_t)y * (stride + 1) + 1;
            for (size_t k = 0; k < stride / 2; ++k) *o++ = (uint16_t)((s[2 * k] << 8) | s[2 * k + 1]);   // big-endian samples
        }
    }
    return FPC_OK;
}

}  // namespace

extern "C" int fpc_png_info(const uint8_t* data, size_t nbytes, int32_t* out5) {
    if (!data || !out5) return FPC_EINVAL;
    PngHeader h;
    const int rc = parse_header(data, nbytes, h);
    if (rc != FPC_OK) return rc;
    out5[0] = (int32_t)h.w; out5[1] = (int32_t)h.h; out5[2] = h.depth; out5[3] = h.ctype;
    out5[4] = h.ctype == 3 ? 3 : h.channels;                                          // channels of the mode-0 output
    return FPC_OK;
}

extern "C" int fpc_png_decode(const uint8_t* data, size_t nbytes, void* out, size_t out_bytes, int mode) {
    if (!data || !out) return FPC_EINVAL;
    return decode_one(data, nbytes, out, out_bytes, mode);
}

extern "C" int fpc_png_decode_batch(const uint8_t* const* datas, const size_t* sizes, int n, uint8_t* out, int H, int W,
                                    int threads) {
    if (n < 0 || H < 1 || W < 1 || (n > 0 && (!datas || !sizes || !out))) return FPC_EINVAL;
    if (n == 0) return FPC_OK;
    for (int i = 0; i < n; ++i) {                     // every file must be an H x W image: checked before any byte is written
        PngHeader h;
        if (!datas[i]) return FPC_EINVAL;
        const int rc = parse_header(datas[i], sizes[i], h);
        if (rc != FPC_OK) return rc;
        if ((int)h.w != W || (int)h.h != H) return FPC_EINVAL;
    }
    const size_t frame = (size_t)H * W * 3;
    const int nt = threads < 1 ? 1 : (threads > n ? n : threads);
    std::atomic<int> next(0), err(FPC_OK);
    auto work = [&]() {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            const int rc = decode_one(datas[i], sizes[i], out + (size_t)i * frame, frame, 3);
            if (rc != FPC_OK) err.store(rc);
        }
    };
    if (nt == 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
    }
    return err.load();
}
