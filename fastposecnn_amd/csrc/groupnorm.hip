// groupnorm.hip — GroupNorm(32 groups) + ReLU on channel-last activations, forward and backward, for the training step
// (BASELINE.json configs[4]).  The reference's decoder blocks are conv3x3 -> GroupNorm(32, 128) -> ReLU
// (segmentation_models_pytorch Conv3x3GNReLU, call sites F/lib/pose_regressor.py:608-666) under autograd; torch's
// GroupNorm kernels work on NCHW, which costs a layout copy on both sides of every block when the convolutions are
// channel-last (1.1 GB per step at batch 8, tools_dev/train_copy_census.py).
//
// With 128 channels in 32 groups a group is FOUR consecutive channels = one float4 of a pixel.  Per (image, group):
//   forward   mean, rstd over H W x 4 values; y = relu((x - mean) rstd gamma + beta)
//   backward  z = xhat gamma + beta, dz = dy [z > 0];  dbeta_c = sum dz, dgamma_c = sum dz xhat;
//             dx = rstd (dz gamma - s1 / N - xhat s2 / N), s1 = sum_group dz gamma, s2 = sum_group dz gamma xhat
// Two passes each way: per-chunk partial sums (f32 per thread over <= 128 values, combined in double, fixed order:
// deterministic), then the elementwise pass, whose workgroups first fold the image's partials.
#include <algorithm>

#include "common.hpp"

namespace fpc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGnChunk = 256;     // pixels per partial-sum workgroup

struct GnArgs {
    const float* x;       // conv output [B, HW, C] channel-last
    const float* dy;      // backward: gradient of the block's output
    float* out;           // forward: y; backward: dx
    const float* gamma; const float* beta;
    float* part;          // forward [B][chunks][Q][2]; backward [B][chunks][C][2]
    float* stats;         // [B][Q][2] mean, rstd (written by the forward, read by the backward)
    int B, HW, C, Q, chunks;
    float eps;
};

// per group (= channel quad) of the chunk: the sum of its values and the sum of squares ABOUT THE CHUNK'S OWN MEAN (second sweep,
// served by the caches).  E[x^2] - mean^2 from plain sums loses every digit once |mean| >> std (activations with a large
// offset: the variance clamped to 0 and rstd blown up where torch's Welford form is exact); per-chunk centred sums combined
// by Chan's formula in double do not.
__global__ __launch_bounds__(256) void k_gn4_stats(const GnArgs a) {
    __shared__ float red[256][2];
    __shared__ float s_mean[64];
    const int t = threadIdx.x, Q = a.Q, q = t % Q, r = t / Q, R = 256 / Q;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int p0 = chunk * kGnChunk, p1 = min(a.HW, p0 + kGnChunk);
    float s = 0.f;
    for (int p = p0 + r; p < p1; p += R) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + ((size_t)b * a.HW + p) * a.C + 4 * q);
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    red[t][0] = s;
    __syncthreads();
    double ds = 0.0;
    if (r == 0) {
        for (int k = 0; k < R; ++k) ds += red[k * Q + q][0];
        ds = (double)(float)ds;                                   // the partial as it is stored: the fold recomputes this very mean
        s_mean[q] = (float)(ds / (4.0 * (p1 - p0)));
    }
    __syncthreads();
    const float m = s_mean[q];
    float ss = 0.f;
    for (int p = p0 + r; p < p1; p += R) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + ((size_t)b * a.HW + p) * a.C + 4 * q);
        const float d0 = v[0] - m, d1 = v[1] - m, d2 = v[2] - m, d3 = v[3] - m;
        ss += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    red[t][1] = ss;
    __syncthreads();
    if (r == 0) {
        double dss = 0.0;
        for (int k = 0; k < R; ++k) dss += red[k * Q + q][1];
        // the centred sum is about the ROUNDED chunk mean m: sum (x - m)^2 = M2 + n (mean_c - m)^2, folded back below through
        // the exact identity with the stored sum
        float* o = a.part + (((size_t)b * a.chunks + chunk) * Q + q) * 2;
        o[0] = (float)ds; o[1] = (float)dss;
    }
}

// mean / rstd of every group of image b from the partials, into LDS (and to a.stats from block 0 of the image)
__device__ __forceinline__ void gn4_fold_stats(const GnArgs& a, int b, float (*st)[2], bool publish) {
    const int t = threadIdx.x;
    if (t < a.Q) {
        double s = 0.0;
        for (int k = 0; k < a.chunks; ++k) s += a.part[(((size_t)b * a.chunks + k) * a.Q + t) * 2];
        const double n = 4.0 * a.HW, mean = s / n;
        // sum (x - mean)^2 = sum_c [ sum (x - m_c)^2 + 2 (m_c - mean) (sum_c - n_c m_c) + n_c (m_c - mean)^2 ],  m_c = the f32 chunk mean
        double m2 = 0.0;
        for (int k = 0; k < a.chunks; ++k) {
            const float* o = a.part + (((size_t)b * a.chunks + k) * a.Q + t) * 2;
            const double nc = 4.0 * (min(a.HW, (k + 1) * kGnChunk) - k * kGnChunk);
            const double mc = (double)(float)((double)o[0] / nc), d = mc - mean;
            m2 += (double)o[1] + 2.0 * d * ((double)o[0] - nc * mc) + nc * d * d;
        }
        double var = m2 / n;
        if (var < 0.0) var = 0.0;
        st[t][0] = (float)mean;
        st[t][1] = (float)(1.0 / sqrt(var + (double)a.eps));
        if (publish) { a.stats[((size_t)b * a.Q + t) * 2] = st[t][0]; a.stats[((size_t)b * a.Q + t) * 2 + 1] = st[t][1]; }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_gn4_apply_relu(const GnArgs a) {
    __shared__ float st[64][2];
    const int b = blockIdx.y;
    gn4_fold_stats(a, b, st, blockIdx.x == 0);
    const int Q = a.Q;
    const size_t n4 = (size_t)a.HW * Q;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i & (size_t)(Q - 1));      // Q divides 256: a power of two
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + (size_t)b * a.HW * a.C + 4 * i);
        const f32x4 g = *reinterpret_cast<const f32x4*>(a.gamma + 4 * q), be = *reinterpret_cast<const f32x4*>(a.beta + 4 * q);
        const float mean = st[q][0], rstd = st[q][1];
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = fmaxf((v[e] - mean) * rstd * g[e] + be[e], 0.f);
        *reinterpret_cast<f32x4*>(a.out + (size_t)b * a.HW * a.C + 4 * i) = y;
    }
}

// per channel: sum dz and sum dz * xhat over the chunk
__global__ __launch_bounds__(256) void k_gn4_bwd_stats(const GnArgs a) {
    __shared__ float red[256][8];
    const int t = threadIdx.x, Q = a.Q, q = t % Q, r = t / Q, R = 256 / Q;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int p0 = chunk * kGnChunk, p1 = min(a.HW, p0 + kGnChunk);
    const float mean = a.stats[((size_t)b * Q + q) * 2], rstd = a.stats[((size_t)b * Q + q) * 2 + 1];
    const f32x4 g = *reinterpret_cast<const f32x4*>(a.gamma + 4 * q), be = *reinterpret_cast<const f32x4*>(a.beta + 4 * q);
    float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = p0 + r; p < p1; p += R) {
        const size_t o = ((size_t)b * a.HW + p) * a.C + 4 * q;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + o), d = *reinterpret_cast<const f32x4*>(a.dy + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (v[e] - mean) * rstd;
            const float dz = (xh * g[e] + be[e] > 0.f) ? d[e] : 0.f;
            sa[e] += dz; sb[e] += dz * xh;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[t][e] = sa[e]; red[t][4 + e] = sb[e]; }
    __syncthreads();
    if (r == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double da = 0.0, db = 0.0;
            for (int k = 0; k < R; ++k) { da += red[k * Q + q][e]; db += red[k * Q + q][4 + e]; }
            float* o = a.part + (((size_t)b * a.chunks + chunk) * a.C + 4 * q + e) * 2;
            o[0] = (float)da; o[1] = (float)db;
        }
    }
}

__global__ __launch_bounds__(256) void k_gn4_bwd_dx(const GnArgs a) {
    __shared__ float sg[64][2];      // s1 / N, s2 / N per group
    const int b = blockIdx.y, Q = a.Q, t = threadIdx.x;
    if (t < Q) {
        double s1 = 0.0, s2 = 0.0;
        for (int e = 0; e < 4; ++e) {
            double da = 0.0, db = 0.0;
            for (int k = 0; k < a.chunks; ++k) {
                const float* o = a.part + (((size_t)b * a.chunks + k) * a.C + 4 * t + e) * 2;
                da += o[0]; db += o[1];
            }
            s1 += da * a.gamma[4 * t + e]; s2 += db * a.gamma[4 * t + e];
        }
        const double n = 4.0 * a.HW;
        sg[t][0] = (float)div_ieee(s1, n); sg[t][1] = (float)div_ieee(s2, n);
    }
    __syncthreads();
    const size_t n4 = (size_t)a.HW * Q;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i & (size_t)(Q - 1));      // Q divides 256: a power of two
        const size_t o = (size_t)b * a.HW * a.C + 4 * i;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + o), d = *reinterpret_cast<const f32x4*>(a.dy + o);
        const f32x4 g = *reinterpret_cast<const f32x4*>(a.gamma + 4 * q), be = *reinterpret_cast<const f32x4*>(a.beta + 4 * q);
        const float mean = a.stats[((size_t)b * Q + q) * 2], rstd = a.stats[((size_t)b * Q + q) * 2 + 1];
        const float m1 = sg[q][0], m2 = sg[q][1];
        f32x4 dx;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (v[e] - mean) * rstd;
            const float dz = (xh * g[e] + be[e] > 0.f) ? d[e] : 0.f;
            dx[e] = rstd * (dz * g[e] - m1 - xh * m2);
        }
        *reinterpret_cast<f32x4*>(a.out + o) = dx;
    }
}

static int gn_args(GnArgs& a, const float* x, const float* gamma, const float* beta, float* out, float* part, float* stats, int B,
                   int HW, int C, int groups, float eps) {
    if (!x || !gamma || !beta || !out || !part || !stats || B < 1 || HW < 1 || B > 65535) return FPC_EINVAL;
    if (groups < 1 || C != 4 * groups || (256 % groups) != 0 || groups > 64) return FPC_EINVAL;      // one float4 per group
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15)) return FPC_EINVAL;
    a.x = x; a.gamma = gamma; a.beta = beta; a.out = out; a.part = part; a.stats = stats;
    a.B = B; a.HW = HW; a.C = C; a.Q = groups; a.chunks = cdiv(HW, kGnChunk); a.eps = eps;
    return FPC_OK;
}

}  // namespace fpc

using namespace fpc;

// floats of scratch (`part`) for the forward or the backward of one call
extern "C" size_t fpc_groupnorm4_relu_scratch_floats(int B, int HW, int C) {
    return (B < 1 || HW < 1 || C < 4) ? 0 : (size_t)B * cdiv(HW, kGnChunk) * C * 2;
}

// y = relu(GroupNorm(x)) for x [B, HW, C] channel-last with C = 4 * groups (groups divides 256, <= 64); stats [B][groups][2]
// receives mean and rstd for the backward.
extern "C" int fpc_groupnorm4_relu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, float* part,
                                       int B, int HW, int C, int groups, float eps, fpc_stream_t stream) {
    GnArgs a{};
    int rc = gn_args(a, x, gamma, beta, y, part, stats, B, HW, C, groups, eps);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_gn4_stats, dim3(a.chunks, B), dim3(256), 0, s, a);
    rc = check_launch();
    if (rc) return rc;
    const unsigned gx = (unsigned)std::min<size_t>(((size_t)HW * groups + 255) / 256, 256);
    hipLaunchKernelGGL(k_gn4_apply_relu, dim3(gx, B), dim3(256), 0, s, a);
    return check_launch();
}

// dx (overwritten) and the per-chunk channel sums part [B][chunks][C][2] = {sum dz, sum dz xhat}: dbeta / dgamma are their
// sums over the first two axes (left to the caller: one small reduction).
extern "C" int fpc_groupnorm4_relu_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* stats,
                                       float* dx, float* part, int B, int HW, int C, int groups, fpc_stream_t stream) {
    GnArgs a{};
    int rc = gn_args(a, x, gamma, beta, dx, part, const_cast<float*>(stats), B, HW, C, groups, 0.f);
    if (rc) return rc;
    if (!dy || ((uintptr_t)dy & 15)) return FPC_EINVAL;
    a.dy = dy;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_gn4_bwd_stats, dim3(a.chunks, B), dim3(256), 0, s, a);
    rc = check_launch();
    if (rc) return rc;
    const unsigned gx = (unsigned)std::min<size_t>(((size_t)HW * groups + 255) / 256, 256);
    hipLaunchKernelGGL(k_gn4_bwd_dx, dim3(gx, B), dim3(256), 0, s, a);
    return check_launch();
}
