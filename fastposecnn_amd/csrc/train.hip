// train.hip — the backward side of the post-network path and the optimiser step (BASELINE config 5).
//
// The reference trains through torch autograd over its Python forward: the one-hot class gathers
// (F/lib/gpu_tensor_funcs.py:52-99), the [N,A,H,W] masked expansions of the aggregation layer
// (F/lib/aggregation_layer.py:119-156) and the final least squares of the vote (RV/ransac_voting_gpu.py:583-599; the
// extension calls before it are non-differentiable selectors).  Here the forward is the inference kernels and the backward is
//   k_post_backward        d loss / d categorical planes from the per-instance gradients: one pass over the label plane,
//                          no N-fold expansion; the vote's term is the derivative of the 2x2 normal-equation solve
//   k_vote_refine_backward the same derivative for a direct ransac_voting_layer_v3 caller (mask / vertex planes)
//   k_cc_backward          d loss / d logits: L2-normalisation Jacobian + scatter into the arg-max class's channel group
//   k_lookahead_radam      Lookahead(RAdam) (F/lib/pose_regressor.py:417-423; catalyst.contrib.nn, upstream) on a flat
//                          parameter shard, gradient clipping (F/train.py: gradient_clip_val) and the inf / NaN guard
//                          (F/lib/pose_regressor.py:341-415) folded in as device-side scalars
//   k_sumsq                per-shard sum of squares + non-finite flag for the two above
// All HBM-bound streaming kernels: lanes own consecutive pixels / elements, every plane row is a coalesced 256-byte access.
#include <algorithm>
#include <cmath>

#include "common.hpp"

namespace fpc {

// d x / d direct for one inlier of the refinement  x = (sum n n^T)^-1 sum n (n . p),  n = (dy, -dx):
//   dL/dn = lam (n . (p - x)) + (lam . n) (p - x),   lam = (sum n n^T)^-1 dL/dx;   dL/d(dx) = -dL/dn_y, dL/d(dy) = dL/dn_x
__device__ __forceinline__ void refine_grad(float px, float py, float dx, float dy, double x0, double x1, double l0,
                                            double l1, float& gdx, float& gdy) {
    const double nx = (double)dy, ny = -(double)dx;
    const double rx = (double)px - x0, ry = (double)py - x1;
    const double r = nx * rx + ny * ry, ln = l0 * nx + l1 * ny;
    const double gnx = l0 * r + ln * rx, gny = l1 * r + ln * ry;
    gdx = (float)(-gny);
    gdy = (float)gnx;
}

constexpr int kTab = 16;   // per-instance table of k_post_backward (doubles)
// [0..3] dL/d(pixel quaternion)  [4..6] dL/d(pixel scales)  [7] dL/d(pixel z)   (already divided by the pixel count)
// [8..9] lam  [10..11] refined x  [12..13] winning hypothesis  [14] foreground count (thinning when > max_num)  [15] vote active

// grid (ceil(HW / 256), B)
__global__ __launch_bounds__(256) void k_post_backward(const int32_t* __restrict__ labels, const float* __restrict__ cat_xy,
                                                       int W, int HW, int N, const int32_t* __restrict__ n_dev,
                                                       const double* __restrict__ tab, float thresh, int max_num,
                                                       uint64_t seed, const uint8_t* __restrict__ keep,
                                                       float* __restrict__ g_q, float* __restrict__ g_s,
                                                       float* __restrict__ g_xy, float* __restrict__ g_z) {
    if (n_dev) N = min(N, *n_dev);
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const size_t o = (size_t)b * HW + p;
    int l = labels[o];
    if (l > N) l = 0;
    float q[4] = {0, 0, 0, 0}, s[3] = {0, 0, 0}, v[2] = {0, 0}, z = 0;
    if (l > 0) {
        const double* t = tab + (size_t)(l - 1) * kTab;
#pragma unroll
        for (int a = 0; a < 4; ++a) q[a] = (float)t[a];
#pragma unroll
        for (int a = 0; a < 3; ++a) s[a] = (float)t[4 + a];
        z = (float)t[7];
        if (t[15] != 0.0) {
            const float dx = cat_xy[(size_t)b * 2 * HW + p], dy = cat_xy[((size_t)b * 2 + 1) * HW + p];
            const float px = (float)(p % W), py = (float)(p / W);
            const int fg = (int)t[14];
            bool in = true;
            if (fg > max_num)
                in = keep ? keep[(size_t)(l - 1) * HW + p] != 0
                          : fpc_rand_keep(seed, (uint32_t)(l - 1), (uint32_t)p, (uint32_t)fg, (uint32_t)max_num) != 0;
            in = in && pair_is_inlier(px, py, dx, dy, sqrtf(dx * dx + dy * dy), (float)t[12], (float)t[13], thresh);
            if (in) refine_grad(px, py, dx, dy, t[10], t[11], t[8], t[9], v[0], v[1]);
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) g_q[((size_t)b * 4 + a) * HW + p] = q[a];
#pragma unroll
    for (int a = 0; a < 3; ++a) g_s[((size_t)b * 3 + a) * HW + p] = s[a];
    g_xy[(size_t)b * 2 * HW + p] = v[0];
    g_xy[((size_t)b * 2 + 1) * HW + p] = v[1];
    g_z[o] = z;
}

// grid (ceil(HW / 256), n); tab [n][8] doubles: lam(2), x(2), win(2), fg, active
__global__ __launch_bounds__(256) void k_vote_refine_backward(const float* __restrict__ mask, const float* __restrict__ vertex,
                                                              int64_t vs_n, int64_t vs_h, int64_t vs_w, int64_t vs_c, int W,
                                                              int HW, const double* __restrict__ tab, float thresh,
                                                              int max_num, uint64_t seed, const uint8_t* __restrict__ keep,
                                                              float* __restrict__ g_vertex /* [n,2,HW] */) {
    const int i = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const double* t = tab + (size_t)i * 8;
    float gx = 0.0f, gy = 0.0f;
    if (t[7] != 0.0 && mask[(size_t)i * HW + p] != 0.0f) {
        const int y = p / W, x = p - y * W;
        const float* vp = vertex + i * vs_n + y * vs_h + x * vs_w;
        const float dx = vp[0], dy = vp[vs_c];
        const int fg = (int)t[6];
        bool in = true;
        if (fg > max_num)
            in = keep ? keep[(size_t)i * HW + p] != 0
                      : fpc_rand_keep(seed, (uint32_t)i, (uint32_t)p, (uint32_t)fg, (uint32_t)max_num) != 0;
        in = in && pair_is_inlier((float)x, (float)y, dx, dy, sqrtf(dx * dx + dy * dy), (float)t[4], (float)t[5], thresh);
        if (in) refine_grad((float)x, (float)y, dx, dy, t[2], t[3], t[0], t[1], gx, gy);
    }
    g_vertex[((size_t)i * 2) * HW + p] = gx;
    g_vertex[((size_t)i * 2 + 1) * HW + p] = gy;
}

// out = sel / |sel| (|sel| == 0 -> sel):  d sel = (g - out (out . g)) / |sel|
template <int A>
__device__ __forceinline__ void normalize_backward(const float* sel, const float* g, float* gs) {
    float n2 = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) n2 += sel[a] * sel[a];
    const float n = sqrtf(n2);
    if (n == 0.0f) {
#pragma unroll
        for (int a = 0; a < A; ++a) gs[a] = g[a];
        return;
    }
    float dot = 0.0f;
#pragma unroll
    for (int a = 0; a < A; ++a) dot += (sel[a] / n) * g[a];
#pragma unroll
    for (int a = 0; a < A; ++a) gs[a] = (g[a] - (sel[a] / n) * dot) / n;
}

// grid (ceil(HW / 256), B).  Every output channel of every pixel is written (zeros outside the pixel's class group).
__global__ __launch_bounds__(256) void k_cc_backward(const int64_t* __restrict__ cat_mask, const float* __restrict__ quat,
                                                     const float* __restrict__ xy, const float* __restrict__ go_q,
                                                     const float* __restrict__ go_s, const float* __restrict__ go_xy,
                                                     const float* __restrict__ go_z, int C, int HW,
                                                     float* __restrict__ g_q, float* __restrict__ g_s,
                                                     float* __restrict__ g_xy, float* __restrict__ g_z) {
    const int b = blockIdx.y, G = C - 1;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const long long c = cat_mask[(size_t)b * HW + p];
    const int g = (c <= 0 || c >= C) ? -1 : (int)c - 1;
    float dq[4] = {0, 0, 0, 0}, ds[3] = {0, 0, 0}, dv[2] = {0, 0}, dz = 0;
    if (g >= 0) {
        float sel[4], gg[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            sel[a] = quat[((size_t)b * 4 * G + 4 * g + a) * HW + p];
            gg[a] = go_q ? go_q[((size_t)b * 4 + a) * HW + p] : 0.0f;
        }
        normalize_backward<4>(sel, gg, dq);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            sel[a] = xy[((size_t)b * 2 * G + 2 * g + a) * HW + p];
            gg[a] = go_xy ? go_xy[((size_t)b * 2 + a) * HW + p] : 0.0f;
        }
        normalize_backward<2>(sel, gg, dv);
#pragma unroll
        for (int a = 0; a < 3; ++a) ds[a] = go_s ? go_s[((size_t)b * 3 + a) * HW + p] : 0.0f;
        dz = go_z ? go_z[(size_t)b * HW + p] : 0.0f;
    }
    for (int k = 0; k < G; ++k) {
        const bool on = k == g;
#pragma unroll
        for (int a = 0; a < 4; ++a) g_q[((size_t)b * 4 * G + 4 * k + a) * HW + p] = on ? dq[a] : 0.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) g_s[((size_t)b * 3 * G + 3 * k + a) * HW + p] = on ? ds[a] : 0.0f;
#pragma unroll
        for (int a = 0; a < 2; ++a) g_xy[((size_t)b * 2 * G + 2 * k + a) * HW + p] = on ? dv[a] : 0.0f;
        g_z[((size_t)b * G + k) * HW + p] = on ? dz : 0.0f;
    }
}

// ---- mask losses --------------------------------------------------------------------------------------------------
// CE + CCE + Focal of F/lib/loss.py:26-98 on the mask logits [B,C,HW] in ONE pass each way.  The reference (and the
// torch-op mirror in lib/loss.py) runs log_softmax three times, two NLL reductions and, for Focal, ~8 elementwise
// kernels per class over the 17 M logits of a batch-8 step.  Per pixel: y = log_softmax(x);
//   CE  = -y[t]                          (nn.CrossEntropyLoss: its own ignore_index = -100)
//   CCE = -y[t]  unless t == ignore      (nn.NLLLoss(ignore_index))
//   Focal = sum over classes c of  a_c (1 - pt)^gamma bce,  bce = BCE-with-logits(y_c, [t == c]),  pt = exp(-bce),
//           a_c = alpha [t == c] + (1 - alpha) [t != c], unless t == ignore     (pytorch_toolbelt, applied to y as the
//           reference does)
// forward: sums and counts (fp64 atomics, one set per workgroup); backward: d/dx of  w0 CE + w1 CCE + w2 Focal  with the
// three weights (upstream gradient / count) read from device memory: no host synchronisation.
struct FocalTerm { float loss, dldy; };

__device__ __forceinline__ FocalTerm focal_term(float y, bool pos, float alpha, float gamma) {
    const float tgt = pos ? 1.0f : 0.0f;
    const float bce = fmaxf(y, 0.0f) - y * tgt + log1pf(expf(-fabsf(y)));
    const float pt = expf(-bce);
    const float om = 1.0f - pt;
    const float a = pos ? alpha : 1.0f - alpha;
    const float fg = gamma == 2.0f ? om * om : powf(om, gamma);
    const float sig = 1.0f / (1.0f + expf(-y));
    const float dbce = sig - tgt;
    // d/dy [ om^g bce ] = g om^(g-1) pt dbce bce + om^g dbce
    const float dfg = gamma == 2.0f ? 2.0f * om : (om > 0.0f ? gamma * powf(om, gamma - 1.0f) : 0.0f);
    FocalTerm r;
    r.loss = a * fg * bce;
    r.dldy = a * dbce * (dfg * pt * bce + fg);
    return r;
}

template <int MAXC, bool BWD>
__global__ __launch_bounds__(256) void k_mask_losses(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                     int C, int HW, long long ignore_ce, long long ignore_cce, float alpha,
                                                     float gamma, double* __restrict__ sums /* fwd: [6] */,
                                                     const float* __restrict__ w3 /* bwd */, float* __restrict__ grad) {
    __shared__ double s_red[4][6];
    const int b = blockIdx.y;
    double acc[6] = {0, 0, 0, 0, 0, 0};        // ce sum, ce count, cce sum, cce count, focal sum, focal count
    float w0 = 0.f, w1 = 0.f, w2 = 0.f;
    if (BWD) { w0 = w3[0]; w1 = w3[1]; w2 = w3[2]; }
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
        const float* x = logits + (size_t)b * C * HW + p;
        const long long t = target[(size_t)b * HW + p];
        float v[MAXC];
        float mx = x[0];
        v[0] = mx;
#pragma unroll
        for (int c = 1; c < MAXC; ++c)
            if (c < C) { v[c] = x[(size_t)c * HW]; mx = fmaxf(mx, v[c]); }
        float se = 0.0f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) se += expf(v[c] - mx);
        const float lse = mx + logf(se);
        const bool ce_on = t != ignore_ce && t >= 0 && t < C, cce_on = t != ignore_cce && t >= 0 && t < C;
        const bool foc_on = t != ignore_cce;
        float gy[MAXC];
        float gsum = 0.0f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) {
                const float y = v[c] - lse;
                const bool pos = t == c;
                float g = 0.0f;
                if (pos && ce_on) { acc[0] -= (double)y; g -= w0; }
                if (pos && cce_on) { acc[2] -= (double)y; g -= w1; }
                if (foc_on) {
                    const FocalTerm f = focal_term(y, pos, alpha, gamma);
                    acc[4] += (double)f.loss;
                    g += w2 * f.dldy;
                }
                gy[c] = g;
                gsum += g;
            }
        acc[1] += ce_on ? 1.0 : 0.0;
        acc[3] += cce_on ? 1.0 : 0.0;
        acc[5] += foc_on ? 1.0 : 0.0;
        if (BWD) {      // y = x - lse(x):  dL/dx_j = g_j - softmax_j sum_c g_c
            float* go = grad + (size_t)b * C * HW + p;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) go[(size_t)c * HW] = gy[c] - expf(v[c] - lse) * gsum;
        }
    }
    if (!BWD) {
        const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double r = wave_reduce_add(acc[a]);
            if (lane == 0) s_red[wv][a] = r;
        }
        __syncthreads();
        if (threadIdx.x < 6) unsafeAtomicAdd(&sums[threadIdx.x], s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
    }
}

// ---- optimiser --------------------------------------------------------------------------------------------------
// out[0] += sum g^2 (fp64), out[1] != 0 when a non-finite element was seen.  One atomic pair per workgroup.
__global__ __launch_bounds__(256) void k_sumsq(const float* __restrict__ g, size_t n, double* __restrict__ out) {
    __shared__ double s_part[4];
    __shared__ int s_bad[4];
    double acc = 0.0;
    int bad = 0;
    const size_t n4 = n / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = g4[i];
        acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        bad |= (int)!(fabsf(v.x) <= 3.4028235e38f) | (int)!(fabsf(v.y) <= 3.4028235e38f) |
               (int)!(fabsf(v.z) <= 3.4028235e38f) | (int)!(fabsf(v.w) <= 3.4028235e38f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[n4 * 4 + threadIdx.x];
        acc += (double)v * v;
        bad |= !(fabsf(v) <= 3.4028235e38f);
    }
    acc = wave_reduce_add(acc);
    bad = wave_reduce_add(bad);
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    if (lane == 0) { s_part[w] = acc; s_bad[w] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsafeAtomicAdd(&out[0], s_part[0] + s_part[1] + s_part[2] + s_part[3]);
        if (s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3]) unsafeAtomicAdd(&out[1], 1.0);
    }
}

struct OptArgs {
    float lr, beta1, beta2, eps, weight_decay, step_size, la_alpha;
    int rectified, la_sync, la_init;
};

// ctl (device, f32[2]): [0] gradient scale (clip coefficient, 1 / world already in); [1] != 0: the inf / NaN guard fired.
// The reference's guard (F/lib/pose_regressor.py:341-415) clears the gradients (`model.zero_grad()`: zeros, not None, in its
// torch) and lets the optimiser step run: m and v decay, weight decay applies, the parameters keep moving along the
// momentum, RAdam's and Lookahead's step counters advance.  Same here: the step runs with g = 0 (selected, not multiplied:
// the gradient holds inf / NaN), so the host-side step count and the device state can never disagree.
__global__ __launch_bounds__(256) void k_lookahead_radam(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         float* __restrict__ slow, size_t n, OptArgs a,
                                                         const float* __restrict__ ctl) {
    const float scale = ctl ? ctl[0] : 1.0f;
    const bool guarded = ctl && ctl[1] != 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = guarded ? 0.0f : g[i] * scale;
        const float vi = v[i] * a.beta2 + (1.0f - a.beta2) * gi * gi;
        const float mi = m[i] * a.beta1 + (1.0f - a.beta1) * gi;
        float pi = p[i];
        if (a.weight_decay != 0.0f) pi += (-a.weight_decay * a.lr) * pi;
        if (a.rectified) pi += -a.step_size * (mi / (sqrtf(vi) + a.eps));
        else pi += -a.step_size * mi;
        v[i] = vi;
        m[i] = mi;
        if (a.la_sync) {
            float si = a.la_init ? pi : slow[i];
            si += (pi - si) * a.la_alpha;
            slow[i] = si;
            pi = si;
        }
        p[i] = pi;
    }
}

}  // namespace fpc

using namespace fpc;

extern "C" int fpc_post_network_backward(const int32_t* labels, const float* cat_xy, int B, int H, int W, int N,
                                         const int32_t* n_dev, const double* table, float inlier_thresh, int max_num,
                                         uint64_t seed, const uint8_t* keep, float* g_q, float* g_s, float* g_xy,
                                         float* g_z, fpc_stream_t stream) {
    if (B < 0 || H < 1 || W < 1 || N < 0 || (int64_t)H * W > (1 << 30)) return FPC_EINVAL;
    if (B == 0) return FPC_OK;
    if (B > 65535) return FPC_EINVAL;
    if (!labels || !cat_xy || !g_q || !g_s || !g_xy || !g_z || (N > 0 && !table)) return FPC_EINVAL;
    const int HW = H * W;
    hipLaunchKernelGGL(k_post_backward, dim3(cdiv(HW, 256), B), dim3(256), 0, (hipStream_t)stream, labels, cat_xy, W, HW, N,
                       n_dev, table, inlier_thresh, max_num, seed, keep, g_q, g_s, g_xy, g_z);
    return check_launch();
}

extern "C" int fpc_vote_refine_backward(const float* mask, const float* vertex, int64_t vs_n, int64_t vs_h, int64_t vs_w,
                                        int64_t vs_c, int n, int H, int W, const double* table, float inlier_thresh,
                                        int max_num, uint64_t seed, const uint8_t* keep, float* g_vertex,
                                        fpc_stream_t stream) {
    if (n < 0 || H < 1 || W < 1 || (int64_t)H * W > (1 << 30)) return FPC_EINVAL;
    if (n == 0) return FPC_OK;
    if (n > 65535 || !mask || !vertex || !table || !g_vertex) return FPC_EINVAL;
    const int HW = H * W;
    hipLaunchKernelGGL(k_vote_refine_backward, dim3(cdiv(HW, 256), n), dim3(256), 0, (hipStream_t)stream, mask, vertex, vs_n,
                       vs_h, vs_w, vs_c, W, HW, table, inlier_thresh, max_num, seed, keep, g_vertex);
    return check_launch();
}

extern "C" int fpc_class_compress_backward(const int64_t* cat_mask, const float* quat, const float* xy, const float* go_q,
                                           const float* go_s, const float* go_xy, const float* go_z, int B, int C, int HW,
                                           float* g_q, float* g_s, float* g_xy, float* g_z, fpc_stream_t stream) {
    if (B < 0 || C < 2 || C > 32 || HW < 1) return FPC_EINVAL;
    if (B == 0) return FPC_OK;
    if (B > 65535 || !cat_mask || !quat || !xy || !g_q || !g_s || !g_xy || !g_z) return FPC_EINVAL;
    hipLaunchKernelGGL(k_cc_backward, dim3(cdiv(HW, 256), B), dim3(256), 0, (hipStream_t)stream, cat_mask, quat, xy, go_q, go_s,
                       go_xy, go_z, C, HW, g_q, g_s, g_xy, g_z);
    return check_launch();
}

extern "C" int fpc_grad_sumsq(const float* g, size_t n, double* out2, fpc_stream_t stream) {
    if (!out2 || (n > 0 && !g)) return FPC_EINVAL;
    if (n == 0) return FPC_OK;
    if (((uintptr_t)g & 15) != 0) return FPC_EINVAL;
    const int grid = (int)std::min<size_t>(2048, (n / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(k_sumsq, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n, out2);
    return check_launch();
}

extern "C" int fpc_lookahead_radam_step(float* p, const float* g, float* m, float* v, float* slow, size_t n, float lr,
                                        float beta1, float beta2, float eps, float weight_decay, int64_t step, int la_k,
                                        float la_alpha, const float* ctl, fpc_stream_t stream) {
    if (step < 1 || la_k < 1 || !(beta1 >= 0.0f && beta1 < 1.0f) || !(beta2 >= 0.0f && beta2 < 1.0f)) return FPC_EINVAL;
    if (n == 0) return FPC_OK;
    if (!p || !g || !m || !v || !slow) return FPC_EINVAL;
    // RAdam's rectification term (Liu et al. 2020, as in catalyst.contrib.nn.RAdam): in double on the host, once per step
    OptArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.la_alpha = la_alpha;
    const double b2t = pow((double)beta2, (double)step);
    const double sma_max = 2.0 / (1.0 - (double)beta2) - 1.0;
    const double sma = sma_max - 2.0 * (double)step * b2t / (1.0 - b2t);
    const double bias1 = 1.0 - pow((double)beta1, (double)step);
    a.rectified = sma >= 5.0 ? 1 : 0;
    a.step_size = a.rectified
                      ? (float)((double)lr * sqrt((1.0 - b2t) * (sma - 4.0) / (sma_max - 4.0) * (sma - 2.0) / sma * sma_max /
                                                  (sma_max - 2.0)) / bias1)
                      : (float)((double)lr / bias1);
    // Lookahead (catalyst): slow weights move on steps 1, 1 + k, 1 + 2k, ...; on step 1 they are the fast weights
    a.la_sync = ((step - 1) % la_k) == 0 ? 1 : 0;
    a.la_init = step == 1 ? 1 : 0;
    const int grid = (int)std::min<size_t>(4096, (n + 255) / 256);
    hipLaunchKernelGGL(k_lookahead_radam, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, slow, n, a, ctl);
    return check_launch();
}

extern "C" int fpc_mask_losses(const float* logits, const int64_t* target, int B, int C, int HW, int64_t ignore_ce,
                               int64_t ignore_cce, float alpha, float gamma, double* sums6, const float* w3, float* grad,
                               fpc_stream_t stream) {
    if (B < 0 || C < 2 || C > 32 || HW < 1) return FPC_EINVAL;
    if (B == 0) return FPC_OK;
    if (B > 65535 || !logits || !target || (!sums6 && !grad) || (grad && !w3)) return FPC_EINVAL;
    // the forward ends in six f64 atomics per workgroup on the same six addresses: few, long-running workgroups (device-scope
    // f64 atomics serialise at ~1 us each once they contend: csrc/aggregate.hip); the backward has none and fills the chip
    const dim3 grid(grad ? std::min(cdiv(HW, 256), 1024) : std::min(cdiv(HW, 256), 96), B);
    hipStream_t s = (hipStream_t)stream;
#define FPC_ML(MAXC, BWD)                                                                                              \
    hipLaunchKernelGGL((k_mask_losses<MAXC, BWD>), grid, dim3(256), 0, s, logits, target, C, HW, (long long)ignore_ce,     \
                       (long long)ignore_cce, alpha, gamma, sums6, w3, grad)
    if (grad) { if (C <= 8) FPC_ML(8, true); else FPC_ML(32, true); }
    else { if (C <= 8) FPC_ML(8, false); else FPC_ML(32, false); }
#undef FPC_ML
    return check_launch();
}
