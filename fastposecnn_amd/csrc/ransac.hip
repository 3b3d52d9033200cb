// ransac.hip — PVNet-derived RANSAC hough voting for gfx950.
//
//  * fpc_generate_hypothesis / fpc_voting_for_hypothesis: B1-compatible kernels
//    (reference: RV/src/ransac_voting_kernel.cu:11-167).
//  * fpc_ransac_voting_v3: the whole of ransac_voting_layer_v3
//    (RV/ransac_voting_gpu.py:518-607) for a batch of instances, without a host round
//    trip and without the hn x tn inlier matrix:
//      k_chunk_count   foreground count per 1024-pixel chunk              (HBM: mask plane)
//      k_chunk_kept    only for instances above max_num: Bernoulli keep   (rare)
//      k_compact       order-preserving stream compaction -> float4 {x,y,dx,dy} list
//                                                                          (HBM: mask + vote planes)
//      k_count_hi      pair sampling + 2-line intersection in the prologue, then an UPPER
//                      BOUND of every hypothesis' inlier count: 4 FMA + 2 compares per
//                      (pixel, hypothesis), wavefront ballot + popcount, no sqrt / divide
//      k_select        candidates in decreasing bound order are re-counted with the exact
//                      reference arithmetic until the bound of the next one cannot beat the
//                      best exact count: the winner, its count, its inlier set and hence the
//                      result are those of the reference's exhaustive vote, bit for bit.
//                      The same pass accumulates the fp64 normal equations; 16 workgroups per
//                      instance, last arriver (agent-scope release/acquire ticket) finishes.
//
// Why the bound is sound (DESIGN.md "vote filter"): the reference accepts a pair when
// fl(cos) > th where fl(cos) carries at most 8 ulp(1) of rounding, so every accepted pair has
// true cos >= th' = th - 1e-6, i.e. |d x e| <= kappa' (d . e) with kappa' = sqrt(1-th'^2)/th'.
// Both u1 = kappa' d.e + d x e and u2 = kappa' d.e - d x e are affine in the hypothesis, so each
// costs two FMAs against per-pixel constants; their own rounding is covered by E_h =
// 2e-6 (|hx|+|hy|+W+H).  Accepted pair  =>  u1 >= -E_h and u2 >= -E_h.
//
// One RANSAC round: the reference's rounds re-evaluate identical samples.
#include "common.hpp"

namespace fpc {

// ----------------------------------------------------------------------------
// B1 kernels

__device__ __forceinline__ void intersect(float4 a, float4 b, float& x, float& y) {
    // RV/src/ransac_voting_kernel.cu:28-45, normal = (dy, -dx); a, b = {cx, cy, dx, dy}
    float nx0 = a.w, ny0 = -a.z, cx0 = a.x, cy0 = a.y;
    float nx1 = b.w, ny1 = -b.z, cx1 = b.x, cy1 = b.y;
    float det_y = nx1 * ny0 - nx0 * ny1;
    float det_x = ny1 * nx0 - ny0 * nx1;
    x = 0.0f; y = 0.0f;
    if (!below_eps(fabsf(det_y)) && !below_eps(fabsf(det_x))) {
        y = (nx1 * (nx0 * cx0 + ny0 * cy0) - nx0 * (nx1 * cx1 + ny1 * cy1)) / det_y;
        x = (ny1 * (nx0 * cx0 + ny0 * cy0) - ny0 * (nx1 * cx1 + ny1 * cy1)) / det_x;
    }
}

__global__ void k_b1_generate_hypothesis(const float* __restrict__ direct, const float* __restrict__ coords,
                                         const int32_t* __restrict__ idxs, float* __restrict__ hyp,
                                         int tn, int vn, int hn) {
    int hvi = blockIdx.x * blockDim.x + threadIdx.x;
    if (hvi >= hn * vn) return;
    int hi = hvi / vn, vi = hvi - hi * vn;
    float x = 0.0f, y = 0.0f;
    int t0 = idxs[hi * vn * 2 + vi * 2];
    int t1 = idxs[hi * vn * 2 + vi * 2 + 1];
    if (t0 >= 0 && t0 < tn && t1 >= 0 && t1 < tn) {  // the reference reads out of bounds here
        float4 a = make_float4(coords[(size_t)t0 * 2], coords[(size_t)t0 * 2 + 1],
                               direct[(size_t)t0 * vn * 2 + vi * 2], direct[(size_t)t0 * vn * 2 + vi * 2 + 1]);
        float4 b = make_float4(coords[(size_t)t1 * 2], coords[(size_t)t1 * 2 + 1],
                               direct[(size_t)t1 * vn * 2 + vi * 2], direct[(size_t)t1 * vn * 2 + vi * 2 + 1]);
        intersect(a, b, x, y);
    }
    hyp[hi * vn * 2 + vi * 2] = x;
    hyp[hi * vn * 2 + vi * 2 + 1] = y;
}

// grid (ceil(vn*tn/256), hn): consecutive lanes = consecutive pixels (coalesced
// coords/direct loads and u8 stores), the hypothesis is uniform per block.
__global__ void k_b1_vote(const float* __restrict__ direct, const float* __restrict__ coords,
                          const float* __restrict__ hyp, uint8_t* __restrict__ inliers,
                          int tn, int vn, int hn, float thresh) {
    int vti = blockIdx.x * blockDim.x + threadIdx.x;
    int hi = blockIdx.y;
    if (vti >= vn * tn) return;
    int vi = vti / tn, ti = vti - vi * tn;
    float cx = coords[(size_t)ti * 2], cy = coords[(size_t)ti * 2 + 1];
    float hx = hyp[hi * vn * 2 + vi * 2], hy = hyp[hi * vn * 2 + vi * 2 + 1];
    float nx = direct[(size_t)ti * vn * 2 + vi * 2], ny = direct[(size_t)ti * vn * 2 + vi * 2 + 1];
    float norm1 = sqrtf(nx * nx + ny * ny);
    if (pair_is_inlier(cx, cy, nx, ny, norm1, hx, hy, thresh)) inliers[((size_t)hi * vn + vi) * tn + ti] = 1;
}

// ----------------------------------------------------------------------------
// fused v3

constexpr int kChunk = 1024;       // pixels per compaction chunk = 256 threads x 4
constexpr int kTile = 256;         // pixels staged in LDS per step of the exact count
constexpr int kMeta = 8;           // i32 per instance: fg, tn, win_idx, win_cnt, inl
constexpr int kT = 4;              // pixel tiles (of 64) held in registers per lane in k_count_hi
constexpr int kBlkPx = 4 * 64 * kT;  // pixels per k_count_hi workgroup pass (4 waves)
constexpr int kSelP = 16;          // workgroups per instance in k_select
constexpr int kPartial = 8;        // doubles per k_select partial record

struct Ws {
    int32_t* counts;      // [n, hn]  upper bounds (exact counts in exact mode); zeroed per call
    int32_t* counts_ex;   // [n, hn]  exact counts, only when diagnostics are requested; zeroed
    int32_t* tickets;     // [2, n]   zeroed (k_select, k_refine)
    int32_t* chunk_fg;    // [n, nch]
    int32_t* chunk_kept;  // [n, nch]
    int32_t* meta;        // [n, kMeta]
    float* hyp;           // [n, hn, 2]
    double* partial;      // [n, kSelP, kPartial]   k_refine partial sums
    int32_t* partial_i;   // [n, kSelP, kCand]      k_select partial counts
    float4* px;           // [n, HW]  {x, y, dx, dy}
    size_t zero_bytes;    // leading bytes cleared per call
    size_t total;
};

static Ws carve(void* base, int n, int H, int W, int hn) {
    Ws w;
    size_t HW = (size_t)H * W;
    int nch = cdiv((int)HW, kChunk);
    char* p = (char*)base;
    size_t off = 0;
    w.counts = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)n * hn, 256);
    w.counts_ex = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)n * hn, 256);
    w.tickets = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * 2 * (size_t)n, 256);
    w.zero_bytes = off;
    w.chunk_fg = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)n * nch, 256);
    w.chunk_kept = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)n * nch, 256);
    w.meta = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)n * kMeta, 256);
    w.hyp = (float*)(p + off); off = align_up(off + sizeof(float) * (size_t)n * hn * 2, 256);
    w.partial = (double*)(p + off); off = align_up(off + sizeof(double) * (size_t)n * kSelP * kPartial, 256);
    w.partial_i = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)n * kSelP * 16, 256);
    w.px = (float4*)(p + off); off = align_up(off + sizeof(float4) * (size_t)n * HW, 256);
    w.total = off;
    return w;
}

// foreground (optionally thinned) flags of this thread's 4 pixels
template <bool THIN>
__device__ __forceinline__ int chunk_flags(const float* __restrict__ m, const uint8_t* __restrict__ keep,
                                           int inst, int p0, int HW, uint64_t seed, int fg, int max_num,
                                           bool flag[4]) {
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int p = p0 + k;
        bool f = false;
        if (p < HW) {
            f = m[p] != 0.0f;
            if (THIN && f)
                f = keep ? (keep[(size_t)inst * HW + p] != 0)
                         : (fpc_rand_keep(seed, (uint32_t)inst, (uint32_t)p, (uint32_t)fg, (uint32_t)max_num) != 0);
        }
        flag[k] = f;
        c += f;
    }
    return c;
}

__global__ __launch_bounds__(256) void k_chunk_count(const float* __restrict__ mask, int HW, int nch,
                                                     int32_t* __restrict__ chunk_fg, const int32_t* __restrict__ n_dev) {
    if (n_dev && (int)(blockIdx.y) >= *n_dev) return;   // capacity rows past the device-side instance count
    __shared__ int scratch[4];
    int inst = blockIdx.y, c = blockIdx.x;
    const float* m = mask + (size_t)inst * HW;
    bool flag[4];
    int cnt = chunk_flags<false>(m, nullptr, inst, c * kChunk + threadIdx.x * 4, HW, 0, 0, 0, flag);
    int tot = block_sum_bcast(cnt, scratch);
    if (threadIdx.x == 0) chunk_fg[inst * nch + c] = tot;
}

// Sum of arr[0..nch) and of arr[0..c) for one instance, broadcast to the block.
__device__ __forceinline__ void total_and_prefix(const int32_t* __restrict__ arr, int nch, int c, int* scratch,
                                                 int& total, int& prefix) {
    int t = 0, pf = 0;
    for (int i = threadIdx.x; i < nch; i += blockDim.x) {
        int v = arr[i];
        t += v;
        if (i < c) pf += v;
    }
    total = block_sum_bcast(t, scratch);
    prefix = block_sum_bcast(pf, scratch);
}

__global__ __launch_bounds__(256) void k_chunk_kept(const float* __restrict__ mask, const uint8_t* __restrict__ keep,
                                                    int HW, int nch, uint64_t seed, int max_num,
                                                    const int32_t* __restrict__ chunk_fg,
                                                    int32_t* __restrict__ chunk_kept, const int32_t* __restrict__ n_dev) {
    if (n_dev && (int)(blockIdx.y) >= *n_dev) return;   // capacity rows past the device-side instance count
    __shared__ int scratch[4];
    int inst = blockIdx.y, c = blockIdx.x;
    int fg, pf;
    total_and_prefix(chunk_fg + inst * nch, nch, c, scratch, fg, pf);
    if (fg <= max_num) return;  // not thinned: k_compact uses chunk_fg
    bool flag[4];
    int cnt = chunk_flags<true>(mask + (size_t)inst * HW, keep, inst, c * kChunk + threadIdx.x * 4, HW, seed, fg,
                                max_num, flag);
    int tot = block_sum_bcast(cnt, scratch);
    if (threadIdx.x == 0) chunk_kept[inst * nch + c] = tot;
}

__global__ __launch_bounds__(256) void k_compact(const float* __restrict__ mask, const float* __restrict__ vertex,
                                                 int64_t vs_n, int64_t vs_h, int64_t vs_w, int64_t vs_c,
                                                 const uint8_t* __restrict__ keep, int W, int HW, int nch,
                                                 uint64_t seed, int min_num, int max_num,
                                                 const int32_t* __restrict__ chunk_fg,
                                                 const int32_t* __restrict__ chunk_kept,
                                                 int32_t* __restrict__ meta, float4* __restrict__ px, const int32_t* __restrict__ n_dev) {
    if (n_dev && (int)(blockIdx.y) >= *n_dev) return;   // capacity rows past the device-side instance count
    __shared__ int scratch[4];
    __shared__ int wave_off[4];
    int inst = blockIdx.y, c = blockIdx.x;
    int fg, pf;
    total_and_prefix(chunk_fg + inst * nch, nch, c, scratch, fg, pf);
    int tn = fg;
    bool thin = fg > max_num;
    if (thin) total_and_prefix(chunk_kept + inst * nch, nch, c, scratch, tn, pf);
    if (fg < min_num) tn = 0;  // RV/ransac_voting_gpu.py:536-539
    if (c == 0 && threadIdx.x == 0) {
        meta[inst * kMeta + 0] = fg;
        meta[inst * kMeta + 1] = tn;
    }
    if (tn == 0) return;

    const float* m = mask + (size_t)inst * HW;
    int p0 = c * kChunk + threadIdx.x * 4;
    bool flag[4];
    int cnt = thin ? chunk_flags<true>(m, keep, inst, p0, HW, seed, fg, max_num, flag)
                   : chunk_flags<false>(m, keep, inst, p0, HW, seed, fg, max_num, flag);
    // exclusive scan of cnt over the block (wave scan + wave totals)
    int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        int t = __shfl_up(incl, o, kWave);
        if (lane >= o) incl += t;
    }
    if (lane == kWave - 1) wave_off[w] = incl;
    __syncthreads();
    int base = pf;
    for (int i = 0; i < w; ++i) base += wave_off[i];
    int pos = base + incl - cnt;
    const float* v = vertex + (int64_t)inst * vs_n;
    float4* out = px + (size_t)inst * HW;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!flag[k]) continue;
        int p = p0 + k;
        int y = p / W, x = p - y * W;
        int64_t o = (int64_t)y * vs_h + (int64_t)x * vs_w;
        out[pos++] = make_float4((float)x, (float)y, v[o], v[o + vs_c]);
    }
}

// Hypothesis hi of instance inst (RV/ransac_voting_gpu.py:552,559).
__device__ __forceinline__ void make_hypothesis(const float4* __restrict__ P, int tn, int hn, int inst, int hi,
                                                const int32_t* __restrict__ idxs, uint64_t seed, float& x, float& y) {
    x = 0.0f; y = 0.0f;
    if (tn <= 0) return;
    int t0, t1;
    if (idxs) {
        t0 = idxs[((size_t)inst * hn + hi) * 2];
        t1 = idxs[((size_t)inst * hn + hi) * 2 + 1];
    } else {
        t0 = fpc_rand_index(seed, (uint32_t)inst, (uint32_t)hi, 0u, (uint32_t)tn);
        t1 = fpc_rand_index(seed, (uint32_t)inst, (uint32_t)hi, 1u, (uint32_t)tn);
    }
    if (t0 >= 0 && t0 < tn && t1 >= 0 && t1 < tn) intersect(P[t0], P[t1], x, y);
}

__global__ __launch_bounds__(256) void k_hypothesis(const float4* __restrict__ px, int HW, int hn,
                                                    const int32_t* __restrict__ idxs, uint64_t seed,
                                                    const int32_t* __restrict__ meta, float* __restrict__ hyp, const int32_t* __restrict__ n_dev) {
    if (n_dev && (int)(blockIdx.y) >= *n_dev) return;   // capacity rows past the device-side instance count
    int inst = blockIdx.y;
    int hi = blockIdx.x * blockDim.x + threadIdx.x;
    if (hi >= hn) return;
    float x, y;
    make_hypothesis(px + (size_t)inst * HW, meta[inst * kMeta + 1], hn, inst, hi, idxs, seed, x, y);
    hyp[((size_t)inst * hn + hi) * 2] = x;
    hyp[((size_t)inst * hn + hi) * 2 + 1] = y;
}

// Exact inlier counts, one lane per hypothesis, pixel tile broadcast from LDS.
// grid (ceil(hn/256), S, n).  Used for the diagnostics output and when the threshold is
// outside the filter's domain (th <= 2e-6).
__global__ __launch_bounds__(256) void k_count_exact(const float4* __restrict__ px, int HW, int hn, float thresh,
                                                     const int32_t* __restrict__ meta, const float* __restrict__ hyp,
                                                     int32_t* __restrict__ counts, const int32_t* __restrict__ n_dev) {
    if (n_dev && (int)(blockIdx.z) >= *n_dev) return;   // capacity rows past the device-side instance count
    __shared__ float4 s_px[kTile];
    __shared__ float s_n1[kTile];
    int inst = blockIdx.z;
    int tn = meta[inst * kMeta + 1];
    if (tn == 0) return;
    int hi = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = hi < hn;
    float hx = 0.0f, hy = 0.0f;
    if (live) {
        hx = hyp[((size_t)inst * hn + hi) * 2];
        hy = hyp[((size_t)inst * hn + hi) * 2 + 1];
    }
    const float4* P = px + (size_t)inst * HW;
    int cnt = 0;
    int ntiles = (tn + kTile - 1) / kTile;
    for (int t = blockIdx.y; t < ntiles; t += gridDim.y) {
        int j = t * kTile + threadIdx.x;
        __syncthreads();
        if (j < tn) {
            float4 q = P[j];
            s_px[threadIdx.x] = q;
            s_n1[threadIdx.x] = sqrtf(q.z * q.z + q.w * q.w);
        }
        __syncthreads();
        int m = min(kTile, tn - t * kTile);
        if (live) {
#pragma unroll 4
            for (int k = 0; k < m; ++k) {
                float4 q = s_px[k];
                cnt += pair_is_inlier(q.x, q.y, q.z, q.w, s_n1[k], hx, hy, thresh);
            }
        }
    }
    if (live && cnt) atomicAdd(&counts[(size_t)inst * hn + hi], cnt);
}

// Upper bound of every hypothesis' inlier count.  grid (ceil(hn/64), S, n), 256 threads.
// Lane g of every wave owns hypothesis h0+g (generated in the prologue, RV/ransac_voting_gpu.py
// :552,559) and pixel slots t*64+lane of the wave's T tiles (six affine constants each, in
// registers).  For g = 0..63 the hypothesis is broadcast with v_readlane into SGPRs, so a
// (tile, hypothesis) step is 4 FMA + 2 compares per lane, one s_and, one s_bcnt1, one s_add —
// no LDS or memory traffic inside the loop.
template <int T>
__global__ __launch_bounds__(256) void k_count_hi(const float4* __restrict__ px, int HW, int hn, float wh,
                                                  float kappa, const int32_t* __restrict__ idxs, uint64_t seed,
                                                  const int32_t* __restrict__ meta, const float* __restrict__ hyp,
                                                  int32_t* __restrict__ counts, const int32_t* __restrict__ n_dev) {
    if (n_dev && (int)blockIdx.z >= *n_dev) return;
    constexpr int kBlk = 4 * kWave * T;
    __shared__ int s_cnt[kWave];
    int inst = blockIdx.z;
    int tn = meta[inst * kMeta + 1];
    int h0 = blockIdx.x * kWave;
    int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    const float4* P = px + (size_t)inst * HW;
    if (threadIdx.x < kWave) s_cnt[threadIdx.x] = 0;

    // this lane's hypothesis (generated once by k_hypothesis): (hx, hy, -E_h); (0, 0, -inf) outside the
    // filter's domain (every voting pixel then counts: still an upper bound), (0, 0, +inf) past hn
    float hx = 0.f, hy = 0.f, ne = __builtin_huge_valf();
    int hi = h0 + lane;
    if (hi < hn) {
        float x = hyp[((size_t)inst * hn + hi) * 2], y = hyp[((size_t)inst * hn + hi) * 2 + 1];
        float s = fabsf(x) + fabsf(y);
        if (s <= 1e18f) { hx = x; hy = y; ne = -2e-6f * (s + wh); }   // false for inf / NaN
        else ne = -__builtin_huge_valf();
    }
    if (tn == 0) return;   // uniform
    __syncthreads();

    int cnt_v = 0;
    int nblk = (tn + kBlk - 1) / kBlk;
    const float qnan = __builtin_nanf("");
    for (int blk = blockIdx.y; blk < nblk; blk += gridDim.y) {
        float a1x[T], a1y[T], c1[T], a2x[T], a2y[T], c2[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            int j = blk * kBlk + (w * T + t) * kWave + lane;
            a1x[t] = a1y[t] = a2x[t] = a2y[t] = 0.f;
            c1[t] = c2[t] = qnan;   // NaN >= x is false: empty slots and zero votes never count
            if (j < tn) {
                float4 q = P[j];
                float n1 = sqrtf(q.z * q.z + q.w * q.w);
                if (!below_eps(n1) && n1 <= 3.0e38f) {
                    float ex = q.z / n1, ey = q.w / n1;
                    a1x[t] = kappa * ex + ey; a1y[t] = kappa * ey - ex;
                    a2x[t] = kappa * ex - ey; a2y[t] = kappa * ey + ex;
                    c1[t] = -(q.x * a1x[t] + q.y * a1y[t]);
                    c2[t] = -(q.x * a2x[t] + q.y * a2y[t]);
                }
            }
        }
#pragma unroll 4
        for (int g = 0; g < kWave; ++g) {
            float gx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hx), g));
            float gy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hy), g));
            float ge = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ne), g));
            int c = 0;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float u1 = __builtin_fmaf(a1x[t], gx, __builtin_fmaf(a1y[t], gy, c1[t]));
                float u2 = __builtin_fmaf(a2x[t], gx, __builtin_fmaf(a2y[t], gy, c2[t]));
                c += __popcll(__builtin_amdgcn_ballot_w64(u1 >= ge && u2 >= ge));
            }
            cnt_v += (lane == g) ? c : 0;
        }
    }
    if (cnt_v) atomicAdd(&s_cnt[lane], cnt_v);
    __syncthreads();
    if (threadIdx.x < kWave && h0 + threadIdx.x < hn) {
        int tot = s_cnt[threadIdx.x];
        if (tot) atomicAdd(&counts[(size_t)inst * hn + h0 + threadIdx.x], tot);
    }
}

// ---- k_select / k_refine ------------------------------------------------------

constexpr int kCand = 8;           // candidates re-counted exactly in the first, parallel pass
constexpr int kSelLds = 4096;      // counts of up to this many hypotheses are staged in LDS for the arg-max rounds

// (max count, lowest index) over counts[0..hn) skipping entries whose bit is set in `done`.
// Result broadcast to the block through s_int[0..7].
__device__ __forceinline__ void block_argmax(const int32_t* counts, int hn, const uint32_t* done,
                                             int* s_int, int& bc, int& bi) {
    bc = -1; bi = 0x7fffffff;
    for (int h = threadIdx.x; h < hn; h += blockDim.x) {
        if ((done[h >> 5] >> (h & 31)) & 1u) continue;
        int c = counts[h];
        if (c > bc) { bc = c; bi = h; }
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        int oc = __shfl_down(bc, o, kWave), oi = __shfl_down(bi, o, kWave);
        if (oc > bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    }
    int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    __syncthreads();
    if (lane == 0) { s_int[w] = bc; s_int[4 + w] = bi; }
    __syncthreads();
    bc = s_int[0]; bi = s_int[4];
    for (int i = 1; i < 4; ++i)
        if (s_int[i] > bc || (s_int[i] == bc && s_int[4 + i] < bi)) { bc = s_int[i]; bi = s_int[4 + i]; }
    __syncthreads();
}

// Exact inlier count of ONE hypothesis over pixels first, first+stride, ...; block total in thread 0.
__device__ __forceinline__ int exact_count(const float4* __restrict__ P, int tn, int first, int stride, float wx,
                                           float wy, float thresh, int* s_int) {
    int c = 0;
    for (int j0 = first; j0 < tn; j0 += 4 * stride) {
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int j = j0 + u * stride;
            q[u] = j < tn ? P[j] : make_float4(0.f, 0.f, 0.f, 0.f);   // zero vote: never an inlier
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            c += pair_is_inlier(q[u].x, q[u].y, q[u].z, q[u].w, sqrtf(q[u].z * q[u].z + q[u].w * q[u].w), wx, wy,
                                thresh);
    }
    c = wave_reduce_add(c);
    int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    __syncthreads();
    if (lane == 0) s_int[w] = c;
    __syncthreads();
    if (threadIdx.x == 0) c = s_int[0] + s_int[1] + s_int[2] + s_int[3];
    __syncthreads();
    return c;
}

// Publish this workgroup's partial record and learn whether it arrived last
// (agent-scope release -> relaxed ticket; the last arriver acquires) — cdna_hip_programming.md G16.
__device__ __forceinline__ bool arrive_last(int32_t* ticket, int nwg, int* s_flag) {
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = (t == nwg - 1);
    }
    __syncthreads();
    bool last = *s_flag != 0;
    if (last) {
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    return last;
}

// grid (kSelP, n), 256 threads.  `counts` are upper bounds of the exact inlier counts (or the exact
// counts themselves).  The kCand hypotheses with the largest bounds are re-counted exactly by all
// workgroups together; the last arriver decides.  If a bound outside the candidate set can still
// win, that workgroup keeps walking candidates in decreasing bound order on its own (rare).
// Writes meta[2] = winner index (-1: no hypothesis has an inlier), meta[3] = its exact count.
__global__ __launch_bounds__(256) void k_select(const float4* __restrict__ px, int HW, int hn, float thresh,
                                                int32_t* __restrict__ meta, const float* __restrict__ hyp,
                                                const int32_t* __restrict__ counts_all,
                                                int32_t* __restrict__ partial_all, int32_t* __restrict__ tickets, const int32_t* __restrict__ n_dev) {
    if (n_dev && (int)(blockIdx.y) >= *n_dev) return;   // capacity rows past the device-side instance count
    __shared__ int s_int[8];
    __shared__ int s_flag;
    __shared__ int s_cidx[kCand], s_chi[kCand];
    __shared__ int s_wc[4][kCand];
    __shared__ uint32_t s_done[2048];
    __shared__ int32_t s_counts[kSelLds];
    int inst = blockIdx.y;
    int tn = meta[inst * kMeta + 1];
    if (tn == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { meta[inst * kMeta + 2] = -1; meta[inst * kMeta + 3] = 0; }
        return;
    }
    const float4* P = px + (size_t)inst * HW;
    const int32_t* counts = counts_all + (size_t)inst * hn;
    const float* H = hyp + (size_t)inst * hn * 2;
    int32_t* partial = partial_all + (size_t)inst * kSelP * kCand;

    for (int i = threadIdx.x; i < 2048; i += blockDim.x) s_done[i] = 0;
    // arg-max rounds run on an LDS copy of the bounds (a global re-scan per round costs an L2 round trip each)
    const bool in_lds = hn <= kSelLds;
    if (in_lds)
        for (int i = threadIdx.x; i < hn; i += blockDim.x) s_counts[i] = counts[i];
    __syncthreads();
    const int32_t* cnt_src = in_lds ? s_counts : counts;
    // the same candidate list in every workgroup
    int ncand = 0;
    for (int k = 0; k < kCand; ++k) {
        int c_hi, c_idx;
        block_argmax(cnt_src, hn, s_done, s_int, c_hi, c_idx);
        if (c_hi <= 0) break;
        if (threadIdx.x == 0) { s_cidx[k] = c_idx; s_chi[k] = c_hi; s_done[c_idx >> 5] |= 1u << (c_idx & 31); }
        __syncthreads();
        ++ncand;
    }
    if (ncand == 0) {   // no hypothesis has even a possible inlier
        if (blockIdx.x == 0 && threadIdx.x == 0) { meta[inst * kMeta + 2] = -1; meta[inst * kMeta + 3] = 0; }
        return;
    }
    // exact counts of the candidates over this workgroup's slice of the pixels
    float cx[kCand], cy[kCand];
    int cc[kCand];
#pragma unroll
    for (int k = 0; k < kCand; ++k) {
        int idx = s_cidx[k < ncand ? k : 0];
        cx[k] = H[2 * idx]; cy[k] = H[2 * idx + 1]; cc[k] = 0;
    }
    for (int j = blockIdx.x * 256 + threadIdx.x; j < tn; j += kSelP * 256) {
        float4 q = P[j];
        float n1 = sqrtf(q.z * q.z + q.w * q.w);
#pragma unroll
        for (int k = 0; k < kCand; ++k) cc[k] += pair_is_inlier(q.x, q.y, q.z, q.w, n1, cx[k], cy[k], thresh);
    }
    int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < kCand; ++k) {
        int c = wave_reduce_add(cc[k]);
        if (lane == 0) s_wc[w][k] = c;
    }
    __syncthreads();
    if (threadIdx.x < kCand)
        partial[blockIdx.x * kCand + threadIdx.x] =
            s_wc[0][threadIdx.x] + s_wc[1][threadIdx.x] + s_wc[2][threadIdx.x] + s_wc[3][threadIdx.x];
    __syncthreads();
    if (!arrive_last(&tickets[inst], kSelP, &s_flag)) return;

    // last arriver: exact totals, best = (largest count, lowest index)
    int best_cnt = 0, best_idx = -1;
    for (int k = 0; k < ncand; ++k) {
        int tot = 0;
        for (int i = 0; i < kSelP; ++i) tot += partial[i * kCand + k];
        int idx = s_cidx[k];
        if (tot > best_cnt || (tot == best_cnt && tot > 0 && idx < best_idx)) { best_cnt = tot; best_idx = idx; }
    }
    // anything outside the candidate set whose bound can still win (or tie with a lower index)?
    while (true) {
        int n_hi, n_idx;
        block_argmax(cnt_src, hn, s_done, s_int, n_hi, n_idx);
        if (n_hi <= 0 || n_hi < best_cnt || (n_hi == best_cnt && best_idx >= 0 && n_idx > best_idx)) break;
        int qc = exact_count(P, tn, threadIdx.x, 256, H[2 * n_idx], H[2 * n_idx + 1], thresh, s_int);
        if (threadIdx.x == 0) { s_done[n_idx >> 5] |= 1u << (n_idx & 31); s_int[0] = qc; }
        __syncthreads();
        qc = s_int[0];
        __syncthreads();
        if (qc > best_cnt || (qc == best_cnt && qc > 0 && n_idx < best_idx)) { best_cnt = qc; best_idx = n_idx; }
    }
    if (threadIdx.x == 0) {
        meta[inst * kMeta + 2] = best_idx;
        meta[inst * kMeta + 3] = best_idx >= 0 ? best_cnt : 0;
    }
}

// b_inv (RV/ransac_voting_gpu.py:503-516): inverse when regular, pseudo-inverse when singular.
__device__ __forceinline__ void solve2_sym(double a00, double a01, double a11, double b0, double b1, double& x0,
                                           double& x1) {
    x0 = 0.0; x1 = 0.0;
    double tr = a00 + a11, det = a00 * a11 - a01 * a01;
    if (!(tr > 0.0)) return;
    if (det <= 1e-12 * tr * tr) {
        double s = 1.0 / (tr * tr);
        x0 = (a00 * b0 + a01 * b1) * s;
        x1 = (a01 * b0 + a11 * b1) * s;
    } else {
        double inv = 1.0 / det;
        x0 = (a11 * b0 - a01 * b1) * inv;
        x1 = (-a01 * b0 + a00 * b1) * inv;
    }
}

// grid (kSelP, n), 256 threads.  Winner re-vote + fp64 normal equations
// (RV/ransac_voting_gpu.py:583-599); partial sums are combined by the last arriver in fixed
// workgroup order, so the result is bit-reproducible.
__global__ __launch_bounds__(256) void k_refine(const float4* __restrict__ px, int HW, int hn, float thresh,
                                                int32_t* __restrict__ meta, const float* __restrict__ hyp,
                                                double* __restrict__ partial_all, int32_t* __restrict__ tickets,
                                                float* __restrict__ out_xy, const int32_t* __restrict__ n_dev) {
    if (n_dev && (int)(blockIdx.y) >= *n_dev) return;   // capacity rows past the device-side instance count
    __shared__ double s_sum[4][6];
    __shared__ int s_flag;
    int inst = blockIdx.y;
    int tn = meta[inst * kMeta + 1];
    if (tn == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            out_xy[inst * 2] = 0.0f; out_xy[inst * 2 + 1] = 0.0f; meta[inst * kMeta + 4] = 0;
        }
        return;
    }
    const float4* P = px + (size_t)inst * HW;
    double* partial = partial_all + (size_t)inst * kSelP * kPartial;
    int widx = meta[inst * kMeta + 2];
    // all_win_pts stays (0,0) unless some hypothesis has an inlier (:571-574)
    float wx = 0.f, wy = 0.f;
    if (widx >= 0) { wx = hyp[((size_t)inst * hn + widx) * 2]; wy = hyp[((size_t)inst * hn + widx) * 2 + 1]; }
    double v[6] = {0, 0, 0, 0, 0, 0};   // inliers, a00, a01, a11, b0, b1
    for (int j = blockIdx.x * 256 + threadIdx.x; j < tn; j += kSelP * 256) {
        float4 q = P[j];
        float n1 = sqrtf(q.z * q.z + q.w * q.w);
        if (!pair_is_inlier(q.x, q.y, q.z, q.w, n1, wx, wy, thresh)) continue;
        double nx = (double)q.w, ny = -(double)q.z;   // normal = (dy, -dx) :584-586
        double bb = nx * (double)q.x + ny * (double)q.y;
        v[0] += 1.0; v[1] += nx * nx; v[2] += nx * ny; v[3] += ny * ny; v[4] += nx * bb; v[5] += ny * bb;
    }
    int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        double r = wave_reduce_add(v[a]);
        if (lane == 0) s_sum[w][a] = r;
    }
    __syncthreads();
    if (threadIdx.x < 6)
        partial[blockIdx.x * kPartial + threadIdx.x] =
            s_sum[0][threadIdx.x] + s_sum[1][threadIdx.x] + s_sum[2][threadIdx.x] + s_sum[3][threadIdx.x];
    __syncthreads();
    if (!arrive_last(&tickets[inst], kSelP, &s_flag)) return;
    if (threadIdx.x == 0) {
        double t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < kSelP; ++i)
            for (int a = 0; a < 6; ++a) t[a] += partial[i * kPartial + a];
        double x0, x1;
        solve2_sym(t[1], t[2], t[3], t[4], t[5], x0, x1);
        out_xy[inst * 2] = (float)x0;
        out_xy[inst * 2 + 1] = (float)x1;
        meta[inst * kMeta + 4] = (int)t[0];
    }
}

__global__ void k_export_meta(const int32_t* __restrict__ meta, int n, int32_t* out_tn, int32_t* out_win_idx,
                              int32_t* out_win_count, int32_t* out_inl) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (out_tn) out_tn[i] = meta[i * kMeta + 1];
    if (out_win_idx) out_win_idx[i] = meta[i * kMeta + 2];
    if (out_win_count) out_win_count[i] = meta[i * kMeta + 3];
    if (out_inl) out_inl[i] = meta[i * kMeta + 4];
}

}  // namespace fpc

using namespace fpc;

extern "C" int fpc_generate_hypothesis(const float* direct, const float* coords, const int32_t* idxs, float* hyp,
                                       int tn, int vn, int hn, fpc_stream_t stream) {
    if (tn < 0 || vn < 1 || hn < 0) return FPC_EINVAL;
    if (hn == 0) return FPC_OK;
    if (!direct || !coords || !idxs || !hyp) return FPC_EINVAL;
    int total = hn * vn;
    hipLaunchKernelGGL(k_b1_generate_hypothesis, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, direct,
                       coords, idxs, hyp, tn, vn, hn);
    return check_launch();
}

extern "C" int fpc_voting_for_hypothesis(const float* direct, const float* coords, const float* hyp,
                                         uint8_t* inliers, int tn, int vn, int hn, float inlier_thresh,
                                         fpc_stream_t stream) {
    if (tn < 0 || vn < 1 || hn < 0) return FPC_EINVAL;
    if (hn == 0 || tn == 0) return FPC_OK;
    if (!direct || !coords || !hyp || !inliers) return FPC_EINVAL;
    if (hn > 65535) return FPC_EINVAL;
    hipLaunchKernelGGL(k_b1_vote, dim3(cdiv(vn * tn, 256), hn), dim3(256), 0, (hipStream_t)stream, direct, coords,
                       hyp, inliers, tn, vn, hn, inlier_thresh);
    return check_launch();
}

extern "C" size_t fpc_ransac_workspace_bytes(int n, int H, int W, int hn) {
    if (n <= 0 || H < 1 || W < 1 || hn < 1) return 256;
    return carve(nullptr, n, H, W, hn).total;
}

extern "C" int fpc_ransac_voting_v3(const float* mask, const float* vertex, int64_t vs_n, int64_t vs_h, int64_t vs_w,
                                    int64_t vs_c, int n, const int32_t* n_dev, int H, int W, int hn,
                                    const int32_t* idxs, const uint8_t* keep, uint64_t seed, float inlier_thresh,
                                    int min_num, int max_num, float* out_xy, int32_t* out_tn, int32_t* out_win_idx,
                                    int32_t* out_win_count, int32_t* out_inl_count, float* out_hyp,
                                    int32_t* out_counts, void* ws, size_t ws_bytes, fpc_stream_t stream) {
    if (n < 0 || H < 1 || W < 1 || hn < 1 || hn > 65536 || max_num < 1) return FPC_EINVAL;
    if ((int64_t)H * W > (1 << 30)) return FPC_EINVAL;
    if (n == 0) return FPC_OK;
    if (!mask || !vertex || !out_xy || !ws) return FPC_EINVAL;
    if (n > 65535) return FPC_EINVAL;
    if (((uintptr_t)ws & 255) != 0) return FPC_EWORKSPACE;
    Ws w = carve(ws, n, H, W, hn);
    if (ws_bytes < w.total) return FPC_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int HW = H * W, nch = cdiv(HW, kChunk);

    hipError_t e = hipMemsetAsync(w.counts, 0, w.zero_bytes, s);
    if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    hipLaunchKernelGGL(k_chunk_count, dim3(nch, n), dim3(256), 0, s, mask, HW, nch, w.chunk_fg, n_dev);
    hipLaunchKernelGGL(k_chunk_kept, dim3(nch, n), dim3(256), 0, s, mask, keep, HW, nch, seed, max_num, w.chunk_fg,
                       w.chunk_kept, n_dev);
    hipLaunchKernelGGL(k_compact, dim3(nch, n), dim3(256), 0, s, mask, vertex, vs_n, vs_h, vs_w, vs_c, keep, W, HW,
                       nch, seed, min_num, max_num, w.chunk_fg, w.chunk_kept, w.meta, w.px, n_dev);
    hipLaunchKernelGGL(k_hypothesis, dim3(cdiv(hn, 256), n), dim3(256), 0, s, w.px, HW, hn, idxs, seed, w.meta, w.hyp,
                       n_dev);

    // the filter needs th' = th - 1e-6 > 0; otherwise count exactly
    bool fast = inlier_thresh > 2e-6f && inlier_thresh < 3.0e38f;
    auto exact_counts = [&](int32_t* dst) {
        int hb = cdiv(hn, 256);
        int split = 2048 / (n * hb);
        split = split < 8 ? 8 : (split > 128 ? 128 : split);
        hipLaunchKernelGGL(k_count_exact, dim3(hb, split, n), dim3(256), 0, s, w.px, HW, hn, inlier_thresh, w.meta,
                           w.hyp, dst, n_dev);
    };
    if (fast) {
        double thp = (double)inlier_thresh - 1e-6;
        double k2 = 1.0 - thp * thp;
        float kappa = (float)((k2 > 0.0 ? sqrt(k2) : 0.0) / thp) * (1.0f + 1e-6f);
        float wh = (float)(W + H);
        int split = cdiv(max_num < HW ? max_num + max_num / 8 + 64 : HW, kBlkPx);
        split = split < 1 ? 1 : (split > 64 ? 64 : split);
        hipLaunchKernelGGL(k_count_hi<kT>, dim3(cdiv(hn, kWave), split, n), dim3(256), 0, s, w.px, HW, hn, wh, kappa,
                           idxs, seed, w.meta, w.hyp, w.counts, n_dev);
        if (out_counts) exact_counts(w.counts_ex);
    } else {
        exact_counts(w.counts);
    }
    hipLaunchKernelGGL(k_select, dim3(kSelP, n), dim3(256), 0, s, w.px, HW, hn, inlier_thresh, w.meta, w.hyp,
                       w.counts, w.partial_i, w.tickets, n_dev);
    hipLaunchKernelGGL(k_refine, dim3(kSelP, n), dim3(256), 0, s, w.px, HW, hn, inlier_thresh, w.meta, w.hyp,
                       w.partial, w.tickets + n, out_xy, n_dev);
    if (out_tn || out_win_idx || out_win_count || out_inl_count)
        hipLaunchKernelGGL(k_export_meta, dim3(cdiv(n, 256)), dim3(256), 0, s, w.meta, n, out_tn, out_win_idx,
                           out_win_count, out_inl_count);
    if (out_hyp) {
        e = hipMemcpyAsync(out_hyp, w.hyp, sizeof(float) * (size_t)n * hn * 2, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    }
    if (out_counts) {
        e = hipMemcpyAsync(out_counts, fast ? w.counts_ex : w.counts, sizeof(int32_t) * (size_t)n * hn,
                           hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    }
    return check_launch();
}
