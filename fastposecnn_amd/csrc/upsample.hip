// upsample.hip — bilinear upsampling with align_corners = True, forward and backward, for the training step
// (BASELINE.json configs[4]).  The reference's network (segmentation_models_pytorch FPN: the x2 steps of the
// segmentation blocks and the x4 of every head, call sites F/lib/pose_regressor.py:608-666) runs these as
// torch.nn.functional.interpolate / nn.UpsamplingBilinear2d under autograd; at batch 8 torch's kernels take 8 ms forward
// and 2.4-3.4 ms backward per step (rocprofv3, profiles/), about 50x the time the bytes need.
//
// Arithmetic follows ATen's area_pixel_compute_source_index for align_corners: r = (in - 1) / (out - 1) in f32,
// src = r * o, i0 = (int)src, i1 = i0 + (i0 < in - 1), l1 = src - i0, l0 = 1 - l1;
// value = l0y * (l0x * v00 + l1x * v01) + l1y * (l0x * v10 + l1x * v11).
// Either tensor may be NCHW or channel-last: the input is read through its four strides, the thread order follows
// the tensor that is WRITTEN (its innermost dimension fastest), so stores are coalesced in both layouts.
// The backward is the exact adjoint as a gather (no atomics, deterministic), one axis at a time: an input coordinate collects
// from the outputs whose source index is its own or the one before.
#include <algorithm>

#include "common.hpp"

namespace fpc {

struct UpArgs {
    const float* src;     // forward: input [B,C,h,w]; backward: output gradient [B,C,H,W]
    float* dst;           // forward: output; backward: input gradient
    long long s_b, s_c, s_y, s_x;      // element strides of src
    long long d_b, d_c, d_y, d_x;      // element strides of dst (contiguous NCHW or channel-last)
    int B, C, h, w, H, W;              // h, w: small side; H, W: large side
    int order_nhwc;                    // thread order: 1 = (x, c) with c fastest, 0 = x fastest — the layout of the LARGE tensor
    float ry, rx;
};

__device__ __forceinline__ void src_index(float r, int o, int n, int& i0, int& i1, float& l0, float& l1) {
    const float s = r * (float)o;
    i0 = (int)s;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

// Thread -> element (b, c, y, x) of dst (Y x X per image) without 64-bit divisions.  x fastest (NCHW): blockIdx.z = b * C + c
// and the x-dimension of the grid runs over the plane's Y * X elements (full workgroups also on 160-wide maps); channel
// fastest: blockIdx.z = b, blockIdx.y = y and the x-dimension runs over (x, c).  The order is that of the LARGE tensor's
// layout (the output in the forward, the output gradient in the backward): its accesses are the coalesced ones.
__device__ __forceinline__ bool dst_element(const UpArgs& a, int X, int Y, int& b, int& c, int& y, int& x, long long& off) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (a.order_nhwc) {
        if (j >= X * a.C) return false;
        y = blockIdx.y;
        x = j / a.C; c = j - x * a.C; b = blockIdx.z;
    } else {
        if (j >= X * Y) return false;
        y = j / X; x = j - y * X; b = blockIdx.z / a.C; c = blockIdx.z - b * a.C;
    }
    off = b * a.d_b + c * a.d_c + y * a.d_y + x * a.d_x;
    return true;
}

__global__ __launch_bounds__(256) void k_up_bilinear_fwd(const UpArgs a) {
    {
        int b, c, oy, ox;
        long long i;
        if (!dst_element(a, a.W, a.H, b, c, oy, ox, i)) return;
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        src_index(a.ry, oy, a.h, y0, y1, ly0, ly1);
        src_index(a.rx, ox, a.w, x0, x1, lx0, lx1);
        const float* p = a.src + b * a.s_b + c * a.s_c;
        const float v00 = p[y0 * a.s_y + x0 * a.s_x], v01 = p[y0 * a.s_y + x1 * a.s_x];
        const float v10 = p[y1 * a.s_y + x0 * a.s_x], v11 = p[y1 * a.s_y + x1 * a.s_x];
        a.dst[i] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
    }
}

// Output coordinates that can feed input coordinate i: those whose source index is i - 1 or i, i.e. o in
// [(i - 1) / r, (i + 1) / r); one coordinate of slack on both sides for the f32 rounding of r * o, every candidate is
// tested exactly with the forward's own arithmetic.
__device__ __forceinline__ void adjoint_range(float r, int i, int n_out, int& lo, int& hi) {
    if (r > 0.f) {
        lo = max(0, (int)floorf((float)(i - 1) / r) - 1);
        hi = min(n_out - 1, (int)ceilf((float)(i + 1) / r) + 1);
    } else {
        lo = 0; hi = n_out - 1;
    }
}
__device__ __forceinline__ float adjoint_weight(float r, int o, int n_in, int i) {
    int i0, i1;
    float l0, l1;
    src_index(r, o, n_in, i0, i1, l0, l1);
    return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

constexpr int kUpTaps = 12;      // outputs per axis that can touch one input coordinate at scale <= 4 on maps >= 4 wide (2 / r + 3)

// The adjoint is separable: first along x (dout [.., H, W] -> tmp [.., H, w]), then along y (tmp -> din [.., h, w]); a pass
// gathers <= 2 scale + 3 values per element instead of the (2 scale + 3)^2 of the one-pass form (x4: 81 loads per element).
// AXIS 0: dst[b,c,y,i] = sum_o w(o -> i) src[b,c,y,o] over the x axis;  AXIS 1: dst[b,c,i,x] = sum_o w(o -> i) src[b,c,o,x].
// a.h / a.w are dst's sizes, n_in / n_out / r the reduced axis' small and large extent and ratio.
template <int AXIS>
__global__ __launch_bounds__(256) void k_up_bilinear_bwd_1d(const UpArgs a, int n_in, int n_out, float r) {
    int b, c, y, x;
    long long off;
    if (!dst_element(a, a.w, a.h, b, c, y, x, off)) return;
    const int i = AXIS == 0 ? x : y;
    int lo, hi;
    adjoint_range(r, i, n_out, lo, hi);
    const float* p = a.src + b * a.s_b + c * a.s_c + (AXIS == 0 ? y * a.s_y : x * a.s_x);
    const long long st = AXIS == 0 ? a.s_x : a.s_y;
    float acc = 0.f;
    if (hi - lo < kUpTaps) {
        float wgt[kUpTaps], val[kUpTaps];
#pragma unroll
        for (int k = 0; k < kUpTaps; ++k) {      // every load issued before the first use
            wgt[k] = lo + k <= hi ? adjoint_weight(r, lo + k, n_in, i) : 0.f;
            val[k] = wgt[k] != 0.f ? p[(lo + k) * st] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < kUpTaps; ++k) acc += wgt[k] * val[k];
    } else {      // tiny maps (1 / r up to 2 * scale - 1)
        for (int o = lo; o <= hi; ++o) {
            const float wv = adjoint_weight(r, o, n_in, i);
            if (wv != 0.f) acc += wv * p[o * st];
        }
    }
    a.dst[off] = acc;
}


// ---- vector forms (round 3).  The one-element-per-thread kernels above spend their time on index arithmetic (an integer
// division, four 64-bit address chains, the source-index computation) per 4 bytes stored: 1.3-1.7 TB/s on the training
// step's tensors.  These do the arithmetic once per FOUR elements that share it and move 16 bytes per access.

// channel-last source and destination, C % 4 == 0: thread = (output pixel, four channels): four float4 loads, one store
__global__ __launch_bounds__(256) void k_up_fwd_nhwc4(const UpArgs a) {
    const int C4 = a.C >> 2;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.W * C4) return;
    const int ox = j / C4, c4 = j - ox * C4, oy = blockIdx.y, b = blockIdx.z;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    src_index(a.ry, oy, a.h, y0, y1, ly0, ly1);
    src_index(a.rx, ox, a.w, x0, x1, lx0, lx1);
    const float* p = a.src + b * a.s_b + 4 * c4;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(p + y0 * a.s_y + x0 * a.s_x), v01 = *reinterpret_cast<const f32x4*>(p + y0 * a.s_y + x1 * a.s_x);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(p + y1 * a.s_y + x0 * a.s_x), v11 = *reinterpret_cast<const f32x4*>(p + y1 * a.s_y + x1 * a.s_x);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = ly0 * (lx0 * v00[e] + lx1 * v01[e]) + ly1 * (lx0 * v10[e] + lx1 * v11[e]);
    *reinterpret_cast<f32x4*>(a.dst + b * a.d_b + (long long)oy * a.d_y + (long long)ox * a.d_x + 4 * c4) = o;
}

// x-fastest (NCHW) destination, W % 4 == 0, any source layout: block (64, 4) = 64 groups of four consecutive ox x 4 rows of
// plane blockIdx.z; the row's y terms once per thread, one 16-byte store
__global__ __launch_bounds__(256) void k_up_fwd_nchw4(const UpArgs a) {
    const int x4 = blockIdx.x * 64 + threadIdx.x, oy = blockIdx.y * 4 + threadIdx.y;
    if (4 * x4 >= a.W || oy >= a.H) return;
    const int b = blockIdx.z / a.C, c = blockIdx.z - b * a.C;
    int y0, y1;
    float ly0, ly1;
    src_index(a.ry, oy, a.h, y0, y1, ly0, ly1);
    const float* p0 = a.src + b * a.s_b + c * a.s_c + y0 * a.s_y;
    const float* p1 = a.src + b * a.s_b + c * a.s_c + y1 * a.s_y;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int x0, x1;
        float lx0, lx1;
        src_index(a.rx, 4 * x4 + e, a.w, x0, x1, lx0, lx1);
        o[e] = ly0 * (lx0 * p0[x0 * a.s_x] + lx1 * p0[x1 * a.s_x]) + ly1 * (lx0 * p1[x0 * a.s_x] + lx1 * p1[x1 * a.s_x]);
    }
    *reinterpret_cast<f32x4*>(a.dst + b * a.d_b + c * a.d_c + (long long)oy * a.d_y + 4 * x4) = o;
}

// channel-last source (a pixel's channels contiguous, pixels of a row contiguous) -> x-fastest (NCHW) destination: the heads'
// x4 (24 or fewer channels at 120 x 160 -> 480 x 640).  A thread of k_up_fwd_nchw4 gathers 16 values 4 C bytes apart per
// 16 bytes stored.  Here a workgroup owns 64 output columns x 4 output rows x all channels: the source patch (<= 4 rows x
// <= 34 pixels x C) goes to LDS with coalesced loads, a thread keeps ONE output position and its four weights and walks the
// channels: per channel four LDS reads and one store that is 256 contiguous bytes per wave.  dynamic LDS: 4 * 34 * C floats.
constexpr int kUpPatchW = 34, kUpPatchH = 4;
__global__ __launch_bounds__(256) void k_up_fwd_nhwc_to_nchw(const UpArgs a) {
    extern __shared__ float s_patch[];                               // [row][pixel][C]
    const int C = a.C, b = blockIdx.z;
    const int ox0 = blockIdx.x * 64, oy0 = blockIdx.y * 4;
    const int ox = ox0 + (threadIdx.x & 63), oy = oy0 + (threadIdx.x >> 6);
    // the patch: source rows / columns the block's outputs read (src_index is monotone in o)
    const int ox_last = min(a.W, ox0 + 64) - 1, oy_last = min(a.H, oy0 + 4) - 1;
    int ya, yb, xa, xb, t0, t1;
    float f0, f1;
    src_index(a.ry, oy0, a.h, ya, t1, f0, f1);
    src_index(a.ry, oy_last, a.h, t0, yb, f0, f1);
    src_index(a.rx, ox0, a.w, xa, t1, f0, f1);
    src_index(a.rx, ox_last, a.w, t0, xb, f0, f1);
    const int pw = xb - xa + 1, ph = yb - ya + 1, rowf = pw * C;       // <= kUpPatchW, <= kUpPatchH (host checks the scale)
    const float* src = a.src + b * a.s_b + ya * a.s_y + xa * a.s_x;
    for (int r = 0; r < ph; ++r)
        for (int i = threadIdx.x; i < rowf; i += 256) s_patch[r * kUpPatchW * C + i] = src[r * a.s_y + i];      // s_x == C, s_c == 1
    __syncthreads();
    if (ox >= a.W || oy >= a.H) return;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    src_index(a.ry, oy, a.h, y0, y1, ly0, ly1);
    src_index(a.rx, ox, a.w, x0, x1, lx0, lx1);
    const float* p00 = s_patch + ((y0 - ya) * kUpPatchW + (x0 - xa)) * C;
    const float* p01 = s_patch + ((y0 - ya) * kUpPatchW + (x1 - xa)) * C;
    const float* p10 = s_patch + ((y1 - ya) * kUpPatchW + (x0 - xa)) * C;
    const float* p11 = s_patch + ((y1 - ya) * kUpPatchW + (x1 - xa)) * C;
    float* d = a.dst + b * a.d_b + (long long)oy * a.d_y + ox;
#pragma unroll 4
    for (int c = 0; c < C; ++c) d[c * a.d_c] = ly0 * (lx0 * p00[c] + lx1 * p01[c]) + ly1 * (lx0 * p10[c] + lx1 * p11[c]);
}

// adjoint, channel-last on both sides, C % 4 == 0: thread = (dst pixel, four channels); the taps' weights once per four
// channels, float4 loads.  Same tap order as k_up_bilinear_bwd_1d: identical sums.
template <int AXIS>
__global__ __launch_bounds__(256) void k_up_bwd_1d_nhwc4(const UpArgs a, int n_in, int n_out, float r) {
    const int C4 = a.C >> 2;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.w * C4) return;
    const int x = j / C4, c4 = j - x * C4, y = blockIdx.y, b = blockIdx.z;
    const int i = AXIS == 0 ? x : y;
    int lo, hi;
    adjoint_range(r, i, n_out, lo, hi);
    const float* p = a.src + b * a.s_b + 4 * c4 + (AXIS == 0 ? y * a.s_y : x * a.s_x);
    const long long st = AXIS == 0 ? a.s_x : a.s_y;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (hi - lo < kUpTaps) {
        float wgt[kUpTaps];
        f32x4 val[kUpTaps];
#pragma unroll
        for (int k = 0; k < kUpTaps; ++k) {
            wgt[k] = lo + k <= hi ? adjoint_weight(r, lo + k, n_in, i) : 0.f;
            val[k] = wgt[k] != 0.f ? *reinterpret_cast<const f32x4*>(p + (lo + k) * st) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < kUpTaps; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += wgt[k] * val[k][e];
    } else {
        for (int o = lo; o <= hi; ++o) {
            const float wv = adjoint_weight(r, o, n_in, i);
            if (wv != 0.f) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p + o * st);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += wv * v[e];
            }
        }
    }
    *reinterpret_cast<f32x4*>(a.dst + b * a.d_b + (long long)y * a.d_y + (long long)x * a.d_x + 4 * c4) = acc;
}

// adjoint along x, x-fastest source and destination: block (64, 4): thread = dst column i of FOUR consecutive rows of plane
// blockIdx.z (rows 4 (4 blockIdx.y + threadIdx.y) ..+3): the weights depend on i only
__global__ __launch_bounds__(256) void k_up_bwd_x_nchw(const UpArgs a, int n_in, int n_out, float r) {
    const int i = blockIdx.x * 64 + threadIdx.x, y0 = (blockIdx.y * 4 + threadIdx.y) * 4;
    if (i >= a.w || y0 >= a.h) return;
    const int b = blockIdx.z / a.C, c = blockIdx.z - b * a.C;
    int lo, hi;
    adjoint_range(r, i, n_out, lo, hi);
    const float* p = a.src + b * a.s_b + c * a.s_c + y0 * a.s_y;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int rows = min(4, a.h - y0);
    if (hi - lo < kUpTaps) {
        float wgt[kUpTaps];
#pragma unroll
        for (int k = 0; k < kUpTaps; ++k) wgt[k] = lo + k <= hi ? adjoint_weight(r, lo + k, n_in, i) : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q >= rows) break;
            float val[kUpTaps];
#pragma unroll
            for (int k = 0; k < kUpTaps; ++k) val[k] = wgt[k] != 0.f ? p[q * a.s_y + (lo + k) * a.s_x] : 0.f;
#pragma unroll
            for (int k = 0; k < kUpTaps; ++k) acc[q] += wgt[k] * val[k];
        }
    } else {
        for (int o = lo; o <= hi; ++o) {
            const float wv = adjoint_weight(r, o, n_in, i);
            if (wv != 0.f)
                for (int q = 0; q < rows; ++q) acc[q] += wv * p[q * a.s_y + o * a.s_x];
        }
    }
    float* d = a.dst + b * a.d_b + c * a.d_c + (long long)y0 * a.d_y + i;
    for (int q = 0; q < rows; ++q) d[q * a.d_y] = acc[q];
}

// adjoint along y, x-fastest source (rows contiguous, w % 4 == 0): block (64, 4): thread = four consecutive columns of dst
// row blockIdx.y * 4 + threadIdx.y; float4 loads down the taps.  dst through its strides (either layout).
__global__ __launch_bounds__(256) void k_up_bwd_y_nchw4(const UpArgs a, int n_in, int n_out, float r) {
    const int x4 = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    if (4 * x4 >= a.w || i >= a.h) return;
    const int b = blockIdx.z / a.C, c = blockIdx.z - b * a.C;
    int lo, hi;
    adjoint_range(r, i, n_out, lo, hi);
    const float* p = a.src + b * a.s_b + c * a.s_c + 4 * x4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (hi - lo < kUpTaps) {
        float wgt[kUpTaps];
        f32x4 val[kUpTaps];
#pragma unroll
        for (int k = 0; k < kUpTaps; ++k) {
            wgt[k] = lo + k <= hi ? adjoint_weight(r, lo + k, n_in, i) : 0.f;
            val[k] = wgt[k] != 0.f ? *reinterpret_cast<const f32x4*>(p + (lo + k) * a.s_y) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < kUpTaps; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += wgt[k] * val[k][e];
    } else {
        for (int o = lo; o <= hi; ++o) {
            const float wv = adjoint_weight(r, o, n_in, i);
            if (wv != 0.f) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p + o * a.s_y);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += wv * v[e];
            }
        }
    }
    float* d = a.dst + b * a.d_b + c * a.d_c + (long long)i * a.d_y + (long long)(4 * x4) * a.d_x;
    if (a.d_x == 1) *reinterpret_cast<f32x4*>(d) = acc;
    else
        for (int e = 0; e < 4; ++e) d[e * a.d_x] = acc[e];
}

}  // namespace fpc

using namespace fpc;

// dst: contiguous [B, C, Y, X], channel-last if dst_nhwc
static int up_args(UpArgs& a, const float* src, int64_t sb, int64_t sc, int64_t sy, int64_t sx, float* dst, int B, int C, int h,
                   int w, int scale, int dst_nhwc, int Y, int X) {
    if (!src || !dst || B < 1 || C < 1 || h < 1 || w < 1 || (scale != 1 && scale != 2 && scale != 4)) return FPC_EINVAL;
    a.src = src; a.dst = dst; a.s_b = sb; a.s_c = sc; a.s_y = sy; a.s_x = sx;
    a.B = B; a.C = C; a.h = h; a.w = w; a.H = h * scale; a.W = w * scale;
    if (dst_nhwc) { a.d_c = 1; a.d_x = C; a.d_y = (long long)X * C; a.d_b = (long long)Y * X * C; }
    else { a.d_x = 1; a.d_y = X; a.d_c = (long long)Y * X; a.d_b = (long long)C * Y * X; }
    a.ry = a.H > 1 ? (float)(h - 1) / (float)(a.H - 1) : 0.f;
    a.rx = a.W > 1 ? (float)(w - 1) / (float)(a.W - 1) : 0.f;
    return FPC_OK;
}

// grid over dst (Y x X per image), see dst_element
static bool up_grid(const UpArgs& a, int X, int Y, dim3& g) {
    const long long zs = a.order_nhwc ? a.B : (long long)a.B * a.C, xs = a.order_nhwc ? (long long)X * a.C : (long long)X * Y;
    if (zs > 65535 || Y > 65535 || xs > 0x7FFFFFFF - 256) return false;
    g = dim3((unsigned)((xs + 255) / 256), a.order_nhwc ? (unsigned)Y : 1u, (unsigned)zs);
    return true;
}

// out [B,C,h*scale,w*scale] (contiguous; channel-last if out_nhwc) = bilinear(in), align_corners = True; scale 2 or 4;
// `in` through its element strides (any layout).
extern "C" int fpc_upsample_bilinear_fwd(const float* in, int64_t sb, int64_t sc, int64_t sh, int64_t sw, float* out, int B, int C,
                                         int h, int w, int scale, int out_nhwc, fpc_stream_t stream) {
    if (scale != 2 && scale != 4) return FPC_EINVAL;
    UpArgs a{};
    int rc = up_args(a, in, sb, sc, sh, sw, out, B, C, h, w, scale, out_nhwc, h * scale, w * scale);
    if (rc) return rc;
    a.order_nhwc = out_nhwc ? 1 : 0;
    dim3 g;
    if (!up_grid(a, a.W, a.H, g)) return FPC_EINVAL;
    const bool al16 = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    if (out_nhwc && C % 4 == 0 && sc == 1 && al16 && !((sb | sh | sw) & 3)) {
        hipLaunchKernelGGL(k_up_fwd_nhwc4, dim3((unsigned)(((long long)a.W * (C / 4) + 255) / 256), (unsigned)a.H, (unsigned)B), dim3(256), 0,
                           (hipStream_t)stream, a);
    } else if (!out_nhwc && sc == 1 && sw == C && C <= 32 && (a.H + 3) / 4 <= 65535 && B <= 65535) {
        hipLaunchKernelGGL(k_up_fwd_nhwc_to_nchw, dim3((unsigned)((a.W + 63) / 64), (unsigned)((a.H + 3) / 4), (unsigned)B), dim3(256),
                           sizeof(float) * kUpPatchH * kUpPatchW * C, (hipStream_t)stream, a);
    } else if (!out_nhwc && a.W % 4 == 0 && al16 && (a.H + 3) / 4 <= 65535) {
        hipLaunchKernelGGL(k_up_fwd_nchw4, dim3((unsigned)((a.W / 4 + 63) / 64), (unsigned)((a.H + 3) / 4), g.z), dim3(64, 4), 0,
                           (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(k_up_bilinear_fwd, g, dim3(256), 0, (hipStream_t)stream, a);
    }
    return check_launch();
}

// floats of scratch the backward needs: the x-pass result [B, C, h * scale, w]
extern "C" size_t fpc_upsample_bilinear_bwd_scratch_floats(int B, int C, int h, int w, int scale) {
    return (B < 1 || C < 1 || h < 1 || w < 1 || scale < 1) ? 0 : (size_t)B * C * h * scale * w;
}

// din [B,C,h,w] (contiguous; channel-last if din_nhwc, overwritten) = adjoint of the above applied to dout
// [B,C,h*scale,w*scale] (through its element strides).  scratch: fpc_upsample_bilinear_bwd_scratch_floats floats.
extern "C" int fpc_upsample_bilinear_bwd(const float* dout, int64_t sb, int64_t sc, int64_t sh, int64_t sw, float* din, float* scratch,
                                         int B, int C, int h, int w, int scale, int din_nhwc, fpc_stream_t stream) {
    if (!scratch || (scale != 2 && scale != 4) || h < 1 || w < 1) return FPC_EINVAL;
    const int H = h * scale;
    const int order = (C > 1 && sc == 1) ? 1 : 0;      // thread order = dout's layout: its reads are the many
    // pass 1 (x axis): dout -> tmp [B, C, H, w], laid out like dout
    UpArgs a{};
    int rc = up_args(a, dout, sb, sc, sh, sw, scratch, B, C, H, w, 1, order, H, w);
    if (rc) return rc;
    UpArgs a1 = a;
    a1.h = H; a1.w = w; a1.order_nhwc = order;
    const float rx = w * scale > 1 ? (float)(w - 1) / (float)(w * scale - 1) : 0.f;
    const float ry = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    dim3 g;
    if (!up_grid(a1, w, H, g)) return FPC_EINVAL;
    const bool al16 = (((uintptr_t)dout | (uintptr_t)din | (uintptr_t)scratch) & 15) == 0;
    const bool vec_c = order && C % 4 == 0 && al16 && !((sb | sh | sw) & 3);           // channel-last, four channels per thread
    const bool vec_x = !order && sw == 1 && al16 && w % 4 == 0 && !((sb | sc | sh) & 3) && (H + 15) / 16 <= 65535;   // x fastest
    if (vec_c)
        hipLaunchKernelGGL(k_up_bwd_1d_nhwc4<0>, dim3((unsigned)(((long long)w * (C / 4) + 255) / 256), (unsigned)H, (unsigned)B), dim3(256), 0,
                           (hipStream_t)stream, a1, w, w * scale, rx);
    else if (vec_x)
        hipLaunchKernelGGL(k_up_bwd_x_nchw, dim3((unsigned)((w + 63) / 64), (unsigned)((H + 15) / 16), g.z), dim3(64, 4), 0, (hipStream_t)stream,
                           a1, w, w * scale, rx);
    else
        hipLaunchKernelGGL(k_up_bilinear_bwd_1d<0>, g, dim3(256), 0, (hipStream_t)stream, a1, w, w * scale, rx);
    rc = check_launch();
    if (rc) return rc;
    // pass 2 (y axis): tmp -> din [B, C, h, w]
    UpArgs a2{};
    const int64_t t_b = order ? (int64_t)H * w * C : (int64_t)C * H * w, t_c = order ? 1 : (int64_t)H * w;
    const int64_t t_y = order ? (int64_t)w * C : w, t_x = order ? C : 1;
    rc = up_args(a2, scratch, t_b, t_c, t_y, t_x, din, B, C, h, w, 1, din_nhwc, h, w);
    if (rc) return rc;
    a2.h = h; a2.w = w; a2.order_nhwc = order;
    if (!up_grid(a2, w, h, g)) return FPC_EINVAL;
    if (vec_c && din_nhwc)
        hipLaunchKernelGGL(k_up_bwd_1d_nhwc4<1>, dim3((unsigned)(((long long)w * (C / 4) + 255) / 256), (unsigned)h, (unsigned)B), dim3(256), 0,
                           (hipStream_t)stream, a2, h, H, ry);
    else if (vec_x)
        hipLaunchKernelGGL(k_up_bwd_y_nchw4, dim3((unsigned)((w / 4 + 63) / 64), (unsigned)((h + 3) / 4), g.z), dim3(64, 4), 0, (hipStream_t)stream,
                           a2, h, H, ry);
    else
        hipLaunchKernelGGL(k_up_bilinear_bwd_1d<1>, g, dim3(256), 0, (hipStream_t)stream, a2, h, H, ry);
    return check_launch();
}
