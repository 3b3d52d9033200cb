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
    hipLaunchKernelGGL(k_up_bilinear_fwd, g, dim3(256), 0, (hipStream_t)stream, a);
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
    hipLaunchKernelGGL(k_up_bilinear_bwd_1d<1>, g, dim3(256), 0, (hipStream_t)stream, a2, h, H, ry);
    return check_launch();
}
