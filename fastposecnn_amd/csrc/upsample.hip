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
// The backward is the exact adjoint as a gather (no atomics, deterministic): an input pixel collects from the
// outputs whose source rows / columns are its own or the one before.
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

// Thread -> element (b, c, y, x) of dst (Y x X per image) without 64-bit divisions: blockIdx.y is the row; x fastest:
// blockIdx.z = b * C + c and the x-dimension of the grid runs over X; channel fastest: blockIdx.z = b and the x-dimension runs
// over (x, c).  The order is that of the LARGE tensor's layout (the output in the forward, the output gradient in the
// backward): its accesses are the coalesced ones.
__device__ __forceinline__ bool dst_element(const UpArgs& a, int X, int& b, int& c, int& y, int& x, long long& off) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    y = blockIdx.y;
    if (a.order_nhwc) {
        if (j >= X * a.C) return false;
        x = j / a.C; c = j - x * a.C; b = blockIdx.z;
    } else {
        if (j >= X) return false;
        x = j; b = blockIdx.z / a.C; c = blockIdx.z - b * a.C;
    }
    off = b * a.d_b + c * a.d_c + y * a.d_y + x * a.d_x;
    return true;
}

__global__ __launch_bounds__(256) void k_up_bilinear_fwd(const UpArgs a) {
    {
        int b, c, oy, ox;
        long long i;
        if (!dst_element(a, a.W, b, c, oy, ox, i)) return;
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

__global__ __launch_bounds__(256) void k_up_bilinear_bwd(const UpArgs a) {
    int b, c, iy, ix;
    long long i;
    if (!dst_element(a, a.w, b, c, iy, ix, i)) return;
    int y_lo, y_hi, x_lo, x_hi;
    adjoint_range(a.ry, iy, a.H, y_lo, y_hi);
    adjoint_range(a.rx, ix, a.W, x_lo, x_hi);
    const float* p = a.src + b * a.s_b + c * a.s_c;
    float acc = 0.f;
    if (x_hi - x_lo < kUpTaps) {
        // the column weights once, in registers (the inner loop is then one multiply-add per live tap)
        float wx[kUpTaps];
#pragma unroll
        for (int k = 0; k < kUpTaps; ++k) wx[k] = x_lo + k <= x_hi ? adjoint_weight(a.rx, x_lo + k, a.w, ix) : 0.f;
        for (int oy = y_lo; oy <= y_hi; ++oy) {
            const float wy = adjoint_weight(a.ry, oy, a.h, iy);
            if (wy == 0.f) continue;
            const float* q = p + oy * a.s_y + x_lo * a.s_x;
            float row = 0.f;
#pragma unroll
            for (int k = 0; k < kUpTaps; ++k)
                if (wx[k] != 0.f) row += wx[k] * q[k * a.s_x];
            acc += wy * row;
        }
    } else {      // tiny maps (1 / r up to 2 * scale - 1): the same sums, weights recomputed per tap
        for (int oy = y_lo; oy <= y_hi; ++oy) {
            const float wy = adjoint_weight(a.ry, oy, a.h, iy);
            if (wy == 0.f) continue;
            const float* q = p + oy * a.s_y;
            float row = 0.f;
            for (int ox = x_lo; ox <= x_hi; ++ox) {
                const float wx = adjoint_weight(a.rx, ox, a.w, ix);
                if (wx != 0.f) row += wx * q[ox * a.s_x];
            }
            acc += wy * row;
        }
    }
    a.dst[i] = acc;
}

}  // namespace fpc

using namespace fpc;

// dst: contiguous [B, C, Y, X], channel-last if dst_nhwc
static int up_args(UpArgs& a, const float* src, int64_t sb, int64_t sc, int64_t sy, int64_t sx, float* dst, int B, int C, int h,
                   int w, int scale, int dst_nhwc, int Y, int X) {
    if (!src || !dst || B < 1 || C < 1 || h < 1 || w < 1 || (scale != 2 && scale != 4)) return FPC_EINVAL;
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
    const long long zs = a.order_nhwc ? a.B : (long long)a.B * a.C, xs = a.order_nhwc ? (long long)X * a.C : X;
    if (zs > 65535 || Y > 65535 || xs > 0x7FFFFFFF - 256) return false;
    g = dim3((unsigned)((xs + 255) / 256), (unsigned)Y, (unsigned)zs);
    return true;
}

// out [B,C,h*scale,w*scale] (contiguous; channel-last if out_nhwc) = bilinear(in), align_corners = True; scale 2 or 4;
// `in` through its element strides (any layout).
extern "C" int fpc_upsample_bilinear_fwd(const float* in, int64_t sb, int64_t sc, int64_t sh, int64_t sw, float* out, int B, int C,
                                         int h, int w, int scale, int out_nhwc, fpc_stream_t stream) {
    UpArgs a{};
    int rc = up_args(a, in, sb, sc, sh, sw, out, B, C, h, w, scale, out_nhwc, h * scale, w * scale);
    if (rc) return rc;
    a.order_nhwc = out_nhwc ? 1 : 0;
    dim3 g;
    if (!up_grid(a, a.W, a.H, g)) return FPC_EINVAL;
    hipLaunchKernelGGL(k_up_bilinear_fwd, g, dim3(256), 0, (hipStream_t)stream, a);
    return check_launch();
}

// din [B,C,h,w] (contiguous; channel-last if din_nhwc, overwritten) = adjoint of the above applied to dout
// [B,C,h*scale,w*scale] (through its element strides).
extern "C" int fpc_upsample_bilinear_bwd(const float* dout, int64_t sb, int64_t sc, int64_t sh, int64_t sw, float* din, int B, int C,
                                         int h, int w, int scale, int din_nhwc, fpc_stream_t stream) {
    UpArgs a{};
    int rc = up_args(a, dout, sb, sc, sh, sw, din, B, C, h, w, scale, din_nhwc, h, w);
    if (rc) return rc;
    a.order_nhwc = (C > 1 && sc == 1) ? 1 : 0;      // the order of dout's layout: its reads are the many
    dim3 g;
    if (!up_grid(a, w, h, g)) return FPC_EINVAL;
    hipLaunchKernelGGL(k_up_bilinear_bwd, g, dim3(256), 0, (hipStream_t)stream, a);
    return check_launch();
}
