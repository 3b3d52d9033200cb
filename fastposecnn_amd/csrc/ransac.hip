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
//      k_hypothesis    pair sampling + 2-line intersection
//      k_count_hi      EXACT inlier count of every hypothesis in one pass: per (pixel, hypothesis) two
//                      affine forms t = d.e and s = d x e (4 FMA) decide almost every pair — "surely an
//                      inlier" (inside a slightly narrower cone) or "surely not" (outside a slightly wider
//                      one); only the pairs in the thin band between the cones (~1e-3 of them) take the
//                      reference's own arithmetic (sqrt, divide).  Wavefront ballot + popcount.
//      k_refine        arg-max (lowest index on ties), winner re-vote, fp64 normal equations; 16 workgroups
//                      per instance, last arriver (agent-scope release/acquire ticket) finishes.
//
// Why the two cones are sound (DESIGN.md "vote filter"): the reference accepts a pair when fl(cos) > th
// where fl(cos) carries at most 8 ulp(1) < 1e-6 of rounding.  So an accepted pair has true cos >= th' =
// th - 1e-6, i.e. |s| <= kappa' t with kappa' = sqrt(1-th'^2)/th' ("maybe"), and a pair with true
// cos >= th'' = th + 1e-6, i.e. |s| <= kappa'' t, is accepted for sure.  t and s are affine in the hypothesis
// (two FMAs each against per-pixel constants); their own rounding is covered by E_h = 2e-6 (|hx|+|hy|+W+H):
//     accepted  =>  |s| <= kappa' t + E_h        (computed values);   |s| <= kappa'' t - E_h  =>  accepted.
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
constexpr int kT = 2;              // pixel tiles (of 64) held in registers per lane in k_count_hi
constexpr int kBlkPx = 4 * 64 * kT;  // pixels per k_count_hi workgroup pass (4 waves)
constexpr int kSelP = 16;          // workgroups per instance in k_refine
constexpr int kPartial = 8;        // doubles per k_refine partial record

struct Ws {
    int32_t* counts;      // [n, hn]  exact inlier counts; zeroed per call
    int32_t* tickets;     // [n]      zeroed (k_refine)
    int32_t* chunk_fg;    // [n, nch]
    int32_t* chunk_kept;  // [n, nch]
    int32_t* meta;        // [n, kMeta]
    float* hyp;           // [n, hn, 2]
    float* hrec;          // [n, hnp, 4]  {fx, fy, E_h, 0} per hypothesis for k_count_hi's scalar loads (hnp = hn rounded up to 64)
    double* partial;      // [n, kSelP, kPartial]   k_refine partial sums
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
    w.tickets = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)n, 256);
    w.zero_bytes = off;
    w.chunk_fg = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)n * nch, 256);
    w.chunk_kept = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)n * nch, 256);
    w.meta = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)n * kMeta, 256);
    w.hyp = (float*)(p + off); off = align_up(off + sizeof(float) * (size_t)n * hn * 2, 256);
    w.hrec = (float*)(p + off); off = align_up(off + sizeof(float) * (size_t)n * (size_t)(cdiv(hn, kWave) * kWave) * 4, 256);
    w.partial = (double*)(p + off); off = align_up(off + sizeof(double) * (size_t)n * kSelP * kPartial, 256);
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

// Also writes the record k_count_hi walks with scalar loads: {fx, fy, E_h, 0} — the point used by the cones and
// their rounding allowance E_h = 2e-6 (|hx| + |hy| + W + H).  Outside the filter's domain (huge / non-finite
// coordinates) E = +inf: every voting pixel is "maybe" and is decided by the exact test.  Padding entries
// (hi >= hn, up to the next multiple of 64) get E = NaN: nothing ever counts.
__global__ __launch_bounds__(256) void k_hypothesis(const float4* __restrict__ px, int HW, int hn, int hnp,
                                                    const int32_t* __restrict__ idxs, uint64_t seed,
                                                    const int32_t* __restrict__ meta, float* __restrict__ hyp,
                                                    float* __restrict__ hrec, float wh, const int32_t* __restrict__ n_dev) {
    if (n_dev && (int)(blockIdx.y) >= *n_dev) return;   // capacity rows past the device-side instance count
    int inst = blockIdx.y;
    int hi = blockIdx.x * blockDim.x + threadIdx.x;
    if (hi >= hnp) return;
    float4* rec = hrec ? reinterpret_cast<float4*>(hrec) + (size_t)inst * hnp + hi : nullptr;
    if (hi >= hn) {
        if (rec) *rec = make_float4(0.f, 0.f, __builtin_nanf(""), 0.f);
        return;
    }
    float x, y;
    make_hypothesis(px + (size_t)inst * HW, meta[inst * kMeta + 1], hn, inst, hi, idxs, seed, x, y);
    hyp[((size_t)inst * hn + hi) * 2] = x;
    hyp[((size_t)inst * hn + hi) * 2 + 1] = y;
    if (rec) {
        float s = fabsf(x) + fabsf(y);
        *rec = s <= 1e18f ? make_float4(x, y, 2e-6f * (s + wh), 0.f)           // false for inf / NaN
                          : make_float4(0.f, 0.f, __builtin_huge_valf(), 0.f);
    }
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

// Exact inlier count of every hypothesis.  grid (ceil(hn/64), S, n), 256 threads.
// Lane g of every wave owns hypothesis h0+g and pixel slots t*64+lane of the wave's T tiles (six affine
// constants + the raw vote, in registers).  For g = 0..63 the hypothesis is broadcast with v_readlane into
// SGPRs, so a (tile, hypothesis) step is 6 FMA + 2 compares per lane, two ballots — no LDS or memory
// traffic inside the loop; the exact test runs only for the lanes of a tile that fall between the cones.
template <int T>
__global__ __launch_bounds__(256) void k_count_hi(const float4* __restrict__ px, int HW, int hn, int hnp,
                                                  float kappa1, float kappa2, float thresh,
                                                  const int32_t* __restrict__ meta, const float* __restrict__ hyp,
                                                  const float* __restrict__ hrec,
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

    // this lane's hypothesis: the true point (hx, hy), read only by the exact test of the band pairs; the cones
    // take (fx, fy, E_h) of hypothesis h0 + g from k_hypothesis' records with SCALAR loads (the address is
    // workgroup-uniform) — three v_readlane per hypothesis cost 30 cycles of vector issue (tools_dev/valu_bench.hip)
    float hx = 0.f, hy = 0.f;
    int hi = h0 + lane;
    if (hi < hn) { hx = hyp[((size_t)inst * hn + hi) * 2]; hy = hyp[((size_t)inst * hn + hi) * 2 + 1]; }
    const float4* HR = reinterpret_cast<const float4*>(hrec) + (size_t)inst * hnp + h0;
    if (tn == 0) return;   // uniform
    __syncthreads();

    int cnt_v = 0;
    int nblk = (tn + kBlk - 1) / kBlk;
    const float qnan = __builtin_nanf("");
    for (int blk = blockIdx.y; blk < nblk; blk += gridDim.y) {
        float ex[T], ey[T], ct[T], cs[T], n1[T];
        float4 q[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            int j = blk * kBlk + (w * T + t) * kWave + lane;
            ex[t] = ey[t] = 0.f;
            ct[t] = cs[t] = qnan;   // NaN compares false: empty slots and zero votes are never "maybe"
            n1[t] = 0.f;
            q[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < tn) {
                q[t] = P[j];
                n1[t] = sqrtf(q[t].z * q[t].z + q[t].w * q[t].w);
                if (!below_eps(n1[t]) && n1[t] <= 3.0e38f) {
                    ex[t] = q[t].z / n1[t]; ey[t] = q[t].w / n1[t];
                    ct[t] = -(q[t].x * ex[t] + q[t].y * ey[t]);      // t = d . e  = ex gx + ey gy + ct
                    cs[t] = -(q[t].x * ey[t] - q[t].y * ex[t]);      // s = d x e  = ey gx - ex gy + cs
                }
            }
        }
#pragma unroll 4
        for (int g = 0; g < kWave; ++g) {
            const float4 rec = HR[g];
            const float gx = rec.x, gy = rec.y, ge = rec.z;
            int c = 0;
            unsigned long long band[T], any = 0;
#pragma unroll
            for (int t = 0; t < T; ++t) {          // branch-free: 6 FMA + 2 compares per lane
                float tt = __builtin_fmaf(ex[t], gx, __builtin_fmaf(ey[t], gy, ct[t]));
                float ss = fabsf(__builtin_fmaf(ey[t], gx, __builtin_fmaf(-ex[t], gy, cs[t])));
                bool maybe = ss <= __builtin_fmaf(kappa1, tt, ge);
                bool sure = ss <= __builtin_fmaf(kappa2, tt, -ge);
                c += __popcll(__builtin_amdgcn_ballot_w64(sure));
                band[t] = __builtin_amdgcn_ballot_w64(maybe && !sure);
                any |= band[t];
            }
            if (any) {      // wave-uniform, rare: the reference's arithmetic for the pairs between the cones
                float rx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hx), g));
                float ry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hy), g));
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    if (!band[t]) continue;
                    bool in = ((band[t] >> lane) & 1ull) &&
                              pair_is_inlier(q[t].x, q[t].y, q[t].z, q[t].w, n1[t], rx, ry, thresh);
                    c += __popcll(__builtin_amdgcn_ballot_w64(in));
                }
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

// ---- k_refine ------------------------------------------------------------------

// (max count, lowest index) over counts[0..hn).  Result broadcast to the block through s_int[0..7].
__device__ __forceinline__ void block_argmax(const int32_t* counts, int hn, int* s_int, int& bc, int& bi) {
    bc = -1; bi = 0x7fffffff;
    for (int h = threadIdx.x; h < hn; h += blockDim.x) {
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
                                                const int32_t* __restrict__ counts_all,
                                                double* __restrict__ partial_all, int32_t* __restrict__ tickets,
                                                float* __restrict__ out_xy, const int32_t* __restrict__ n_dev) {
    if (n_dev && (int)(blockIdx.y) >= *n_dev) return;   // capacity rows past the device-side instance count
    __shared__ double s_sum[4][6];
    __shared__ int s_flag;
    __shared__ int s_int[8];
    int inst = blockIdx.y;
    int tn = meta[inst * kMeta + 1];
    if (tn == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            out_xy[inst * 2] = 0.0f; out_xy[inst * 2 + 1] = 0.0f;
            meta[inst * kMeta + 2] = -1; meta[inst * kMeta + 3] = 0; meta[inst * kMeta + 4] = 0;
        }
        return;
    }
    const float4* P = px + (size_t)inst * HW;
    double* partial = partial_all + (size_t)inst * kSelP * kPartial;
    // winner = (largest exact count, lowest index) as torch.max (RV/ransac_voting_gpu.py:567); every workgroup
    // finds it for itself.  No hypothesis with an inlier: all_win_pts stays (0,0) (:571-574).
    int wcnt, widx;
    block_argmax(counts_all + (size_t)inst * hn, hn, s_int, wcnt, widx);
    if (wcnt <= 0) { widx = -1; wcnt = 0; }
    if (blockIdx.x == 0 && threadIdx.x == 0) { meta[inst * kMeta + 2] = widx; meta[inst * kMeta + 3] = wcnt; }
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
    const int hnp = cdiv(hn, kWave) * kWave;
    hipLaunchKernelGGL(k_hypothesis, dim3(cdiv(hnp, 256), n), dim3(256), 0, s, w.px, HW, hn, hnp, idxs, seed, w.meta, w.hyp,
                       w.hrec, (float)(W + H), n_dev);

    // the cones need th' = th - 1e-6 > 0; otherwise every pair takes the reference's arithmetic
    bool fast = inlier_thresh > 2e-6f && inlier_thresh < 3.0e38f;
    if (fast) {
        double th1 = (double)inlier_thresh - 1e-6, th2 = (double)inlier_thresh + 1e-6;
        double k1 = 1.0 - th1 * th1, k2 = 1.0 - th2 * th2;
        float kappa1 = (float)((k1 > 0.0 ? sqrt(k1) : 0.0) / th1) * (1.0f + 1e-6f);        // wider
        float kappa2 = (th2 < 1.0 && k2 > 0.0) ? (float)(sqrt(k2) / th2) * (1.0f - 1e-6f) : 0.0f;   // narrower (0: no "sure")
        int split = cdiv(max_num < HW ? max_num + max_num / 8 + 64 : HW, kBlkPx);
        split = split < 1 ? 1 : (split > 64 ? 64 : split);
        hipLaunchKernelGGL(k_count_hi<kT>, dim3(cdiv(hn, kWave), split, n), dim3(256), 0, s, w.px, HW, hn, hnp, kappa1,
                           kappa2, inlier_thresh, w.meta, w.hyp, w.hrec, w.counts, n_dev);
    } else {
        int hb = cdiv(hn, 256);
        int split = 2048 / (n * hb);
        split = split < 8 ? 8 : (split > 128 ? 128 : split);
        hipLaunchKernelGGL(k_count_exact, dim3(hb, split, n), dim3(256), 0, s, w.px, HW, hn, inlier_thresh, w.meta,
                           w.hyp, w.counts, n_dev);
    }
    hipLaunchKernelGGL(k_refine, dim3(kSelP, n), dim3(256), 0, s, w.px, HW, hn, inlier_thresh, w.meta, w.hyp,
                       w.counts, w.partial, w.tickets, out_xy, n_dev);
    if (out_tn || out_win_idx || out_win_count || out_inl_count)
        hipLaunchKernelGGL(k_export_meta, dim3(cdiv(n, 256)), dim3(256), 0, s, w.meta, n, out_tn, out_win_idx,
                           out_win_count, out_inl_count);
    if (out_hyp) {
        e = hipMemcpyAsync(out_hyp, w.hyp, sizeof(float) * (size_t)n * hn * 2, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    }
    if (out_counts) {
        e = hipMemcpyAsync(out_counts, w.counts, sizeof(int32_t) * (size_t)n * hn,
                           hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    }
    return check_launch();
}
