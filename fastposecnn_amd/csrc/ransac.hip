// ransac.hip — PVNet-derived RANSAC hough voting for gfx950.
//
//  * fpc_generate_hypothesis / fpc_voting_for_hypothesis: B1-compatible kernels
//    (reference: RV/src/ransac_voting_kernel.cu:11-167).
//  * fpc_ransac_voting_v3: the whole of ransac_voting_layer_v3 (RV/ransac_voting_gpu.py:518-607) for a batch of
//    instances in FOUR stateless launches, without a host round trip, without the hn x tn inlier matrix and without
//    a compacted pixel list on the critical path:
//
//      k_vote_scan    streams the mask planes once (float4 per lane) into a 1-bit-per-pixel image: per 64 pixels a
//                     u64 word, its in-chunk exclusive count, per 4096-pixel chunk the foreground count.
//                                                                                   (HBM: n x H*W*4 bytes read)
//      k_vote_plan    one 1024-thread workgroup per instance: chunk prefix, the > max_num thinning (:541-545: the
//                     kept pixels replace the bit image, so everything downstream sees the thinned instance), then
//                     the hn hypotheses (:552,559): pair sampling, pixel look-up by rank in the bit image (two
//                     binary searches + select), the vote gathered from the caller's strided planes, 2-line
//                     intersection exactly as .cu:28-45.  Writes the points as SoA rows the count kernel reads
//                     with scalar loads, per 64 hypotheses the rounding allowance E_g of the filter, and zeroes the
//                     instance's count row.
//      k_vote_count   task = (instance, 256-pixel block, slice of the hypotheses); lanes own pixels (looked up by
//                     rank, votes gathered once: the only read of the vote planes), the hypotheses arrive in SGPRs.
//                     Per (pixel, hypothesis): 4 FMA + 1 compare, ballot + s_bcnt1, one v_writelane per
//                     hypothesis — an UPPER BOUND U_h of the inlier count (cone widened by the reference's own
//                     rounding, see below).  A block's counts are combined in LDS and added to the instance's row
//                     with one integer atomic per (block, hypothesis) that has any; slice 0 also leaves the block's
//                     pixels (raw and as filter constants) as two float4 lists for the refinement.
//      k_vote_refine  one 1024-thread workgroup per instance: candidates in order (U desc, index asc), four per pass,
//                     are counted EXACTLY — two cones decide almost every pixel ("surely an inlier" / "surely not"),
//                     the pixels between them take the reference's own arithmetic (.cu:106-125) — until no
//                     remaining U can beat or tie-break the best exact count: that is torch.max's winner (:567, first
//                     maximal index).  The winner's inliers give the fp64 normal equations and the closed-form 2x2
//                     solve (b_inv, :503-516, :583-599).
//
// Why the cones are sound: the reference accepts a pair when fl(cos) > th, where fl(cos) carries at most
// 8 ulp(1) < 1e-6 of rounding.  So an accepted pair has true cos >= th' = th - 1e-6, i.e. |s| <= kappa' t with
// t = d.e, s = d x e (e the unit vote, d = h - p), kappa' = sqrt(1-th'^2)/th' ("maybe"), and a pair with true
// cos >= th'' = th + 1e-6, i.e. |s| <= kappa'' t, is accepted for sure.  t and s are affine in the hypothesis (two
// FMAs each against per-pixel constants); the evaluation's own rounding is at most (3.6e-7 + 4.8e-7 kappa) M with
// M = |hx| + |hy| + W + H (six roundings at magnitude <= M on the s side, eight at kappa M on the t side):
//     accepted  =>  |s| <= kappa' t + E        (computed values);        |s| <= kappa'' t - E  =>  accepted
// for any E >= that bound.  k_vote_count folds kappa' and E into the constants:
// kappa' t + E = (kappa' ex) gx + (kappa' ey) gy + (kappa' ct + E_g), E_g = max over the 64 hypotheses of a group of
// 1e-6 (1 + kappa') M; k_vote_refine classifies with 2e-6 (1 + kappa') M per hypothesis.  Hypotheses with huge or
// non-finite coordinates are outside the filter's domain: their U is the pixel count and every pixel takes the
// reference's arithmetic.  Thresholds <= 2e-6 have no cone: every pair takes the reference's arithmetic (U exact).
//
// One RANSAC round: the reference's rounds re-evaluate identical samples.
#include <algorithm>

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

constexpr int kChunkPx = 4096;       // pixels per k_vote_scan task
constexpr int kChunkWords = 64;      // 64-pixel words per chunk
constexpr int kBlockPx = 256;        // pixels per k_vote_count task (one per lane of a 256-thread workgroup)
constexpr int kPlanI = 8;            // i32 per instance: fg, tn, nblocks
constexpr int kMaxHn = 8192;         // k_vote_refine keeps U in LDS (32 KB)

struct Ws {
    int32_t* plan;        // [n, kPlanI]
    int32_t* chunk_fg;    // [n, nch]
    int32_t* chunk_pre;   // [n, nch + 1]   exclusive prefix of chunk_fg (after thinning: of the kept counts)
    uint32_t* word_pre;   // [n, nwords]    exclusive count of the word inside its chunk
    uint64_t* bits;       // [n, nwords]    1 bit per pixel (after k_vote_plan: per KEPT pixel)
    float* hx;            // [n, hnp]       hypothesis points, SoA (hnp = hn rounded up to 64)
    float* hy;            // [n, hnp]
    float* eg;            // [n, hnp / 64]  E_g per group of 64 hypotheses
    float* hyp;           // [n, hn, 2]     the same points as the reference's [hn,1,2] tensor
    int32_t* upper;       // [n, hnp]       U_h (exact counts in exact mode); zeroed by k_vote_plan
    float4* list;         // [n, HW]        {x, y, dx, dy} of pixel rank j (written by k_vote_count, slice 0)
    float4* clist;        // [n, HW]        {ey, -ex, cs, ct}: the pixel's filter constants (NaN cs: never an inlier)
    int nch, nwords, hnp;
    size_t total;
};

static Ws carve(void* base, int n, int H, int W, int hn) {
    Ws w;
    size_t HW = (size_t)H * W;
    w.nch = cdiv((int)HW, kChunkPx);
    w.nwords = w.nch * kChunkWords;
    w.hnp = cdiv(hn, kWave) * kWave;
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* q = p + off; off = align_up(off + bytes, 256); return q; };
    w.plan = (int32_t*)take(sizeof(int32_t) * (size_t)n * kPlanI);
    w.chunk_fg = (int32_t*)take(sizeof(int32_t) * (size_t)n * w.nch);
    w.chunk_pre = (int32_t*)take(sizeof(int32_t) * (size_t)n * (w.nch + 1));
    w.word_pre = (uint32_t*)take(sizeof(uint32_t) * (size_t)n * w.nwords);
    w.bits = (uint64_t*)take(sizeof(uint64_t) * (size_t)n * w.nwords);
    w.hx = (float*)take(sizeof(float) * (size_t)n * w.hnp);
    w.hy = (float*)take(sizeof(float) * (size_t)n * w.hnp);
    w.eg = (float*)take(sizeof(float) * (size_t)n * (w.hnp / kWave));
    w.hyp = (float*)take(sizeof(float) * (size_t)n * hn * 2);
    w.upper = (int32_t*)take(sizeof(int32_t) * (size_t)n * w.hnp);
    w.list = (float4*)take(sizeof(float4) * (size_t)n * HW);
    w.clist = (float4*)take(sizeof(float4) * (size_t)n * HW);
    w.total = off;
    return w;
}

__device__ __forceinline__ int active_instances(int n, const int32_t* __restrict__ n_dev) {
    if (!n_dev) return n;
    int m = *n_dev;
    return m < n ? (m < 0 ? 0 : m) : n;
}

// 16 bytes holding one 4-bit field each (low nibble) -> 64 bits, field i at bits [4i, 4i+4)
__device__ __forceinline__ uint64_t pack_nibbles8(uint64_t x) {
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return x;
}
__device__ __forceinline__ uint64_t pack_nibbles(uint4 raw) {
    const uint64_t lo = (uint64_t)raw.x | ((uint64_t)raw.y << 32), hi = (uint64_t)raw.z | ((uint64_t)raw.w << 32);
    return pack_nibbles8(lo) | (pack_nibbles8(hi) << 32);
}

// exclusive prefix of v over the 64 lanes; `total` = the wave's sum (all lanes)
__device__ __forceinline__ int wave_excl_scan(int v, int& total) {
    int lane = threadIdx.x & (kWave - 1);
    int incl = v;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        int t = __shfl_up(incl, o, kWave);
        if (lane >= o) incl += t;
    }
    total = __shfl(incl, kWave - 1, kWave);
    return incl - v;
}

// ---- k_vote_scan -------------------------------------------------------------------
// grid-stride over (instance, chunk) tasks; 256 threads; a chunk = 4096 pixels = 4 float4 per lane.
template <bool VEC4>
__global__ __launch_bounds__(256) void k_vote_scan(const float* __restrict__ mask, int HW, int nch, int n,
                                                   const int32_t* __restrict__ n_dev, uint64_t* __restrict__ bits,
                                                   uint32_t* __restrict__ word_pre, int32_t* __restrict__ chunk_fg) {
    __shared__ __attribute__((aligned(16))) uint8_t s_nib[kChunkPx / 4];
    const int total = active_instances(n, n_dev) * nch;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int inst = t / nch, c = t - inst * nch;
        const float* m = mask + (size_t)inst * HW;
        unsigned nb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int fi = k * 256 + threadIdx.x;           // float4 index inside the chunk
            const int p = c * kChunkPx + fi * 4;
            nb[k] = 0;
            if (VEC4) {                                     // HW % 4 == 0 and a 16-byte aligned plane
                if (p < HW) {
                    const float4 v = *reinterpret_cast<const float4*>(m + p);
                    nb[k] = (v.x != 0.0f ? 1u : 0u) | (v.y != 0.0f ? 2u : 0u) | (v.z != 0.0f ? 4u : 0u) | (v.w != 0.0f ? 8u : 0u);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (p + q < HW && m[p + q] != 0.0f) nb[k] |= 1u << q;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) s_nib[k * 256 + threadIdx.x] = (uint8_t)nb[k];
        __syncthreads();
        if (threadIdx.x < kChunkWords) {
            const uint64_t word = pack_nibbles(*reinterpret_cast<const uint4*>(s_nib + 16 * threadIdx.x));
            const int cnt = __popcll(word);
            int tot;
            const int ex = wave_excl_scan(cnt, tot);
            const size_t wi = (size_t)inst * nch * kChunkWords + (size_t)c * kChunkWords + threadIdx.x;
            bits[wi] = word;
            word_pre[wi] = (uint32_t)ex;
            if (threadIdx.x == 0) chunk_fg[(size_t)inst * nch + c] = tot;
        }
        __syncthreads();
    }
}

// ---- pixel look-up by rank ---------------------------------------------------------
struct Lut {
    const int32_t* chunk_pre;   // [nch + 1] of this instance (global, or a copy in LDS)
    const uint32_t* word_pre;   // [nwords]
    const uint64_t* bits;       // [nwords]
    int nch;
};

// position of the r-th (0-based) set bit of w; r < popcount(w)
__device__ __forceinline__ int select64(uint64_t w, int r) {
    int pos = 0;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const int c = __popcll((w >> pos) & ((1ull << s) - 1ull));
        if (r >= c) { r -= c; pos += s; }
    }
    return pos;
}

// linear pixel index of the j-th foreground (kept) pixel in raster order; 0 <= j < chunk_pre[nch]
__device__ __forceinline__ int lookup_pixel(const Lut& L, int j) {
    int lo = 0, hi = L.nch;                       // chunk_pre[lo] <= j < chunk_pre[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (L.chunk_pre[mid] <= j) lo = mid; else hi = mid;
    }
    int r = j - L.chunk_pre[lo];
    const uint32_t* wp = L.word_pre + (size_t)lo * kChunkWords;
    int wl = 0, wh = kChunkWords;                 // wp[wl] <= r, and r < wp[wh] where wh < 64
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int mid = (wl + wh) >> 1;
        if ((int)wp[mid] <= r) wl = mid; else wh = mid;
    }
    r -= (int)wp[wl];
    const int w = lo * kChunkWords + wl;
    return w * 64 + select64(L.bits[w], r);
}

__device__ __forceinline__ float4 gather_pixel(const float* __restrict__ v, int64_t vs_h, int64_t vs_w, int64_t vs_c,
                                               int W, int p) {
    const int y = p / W, x = p - y * W;
    const int64_t o = (int64_t)y * vs_h + (int64_t)x * vs_w;
    return make_float4((float)x, (float)y, v[o], v[o + vs_c]);
}

// ---- k_vote_plan -------------------------------------------------------------------
// Exclusive scan of arr[0..cnt) into out[0..cnt] (and s_out, when given), out[cnt] = total, by the whole workgroup.
__device__ __forceinline__ int block_scan_chunks(const int32_t* __restrict__ arr, int32_t* __restrict__ out,
                                                 int* s_out, int cnt, int* s_w /* >= 17 ints */) {
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave, nw = blockDim.x / kWave;
    int carry = 0;
    for (int base = 0; base < cnt; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const int v = i < cnt ? arr[i] : 0;
        int wt;
        const int ex = wave_excl_scan(v, wt);
        __syncthreads();
        if (lane == 0) s_w[w] = wt;
        __syncthreads();
        int off = carry, tile = 0;
        for (int k = 0; k < nw; ++k) { const int x = s_w[k]; if (k < w) off += x; tile += x; }
        if (i < cnt) { out[i] = off + ex; if (s_out) s_out[i] = off + ex; }
        carry += tile;
    }
    if (threadIdx.x == 0) { out[cnt] = carry; if (s_out) s_out[cnt] = carry; }
    return carry;
}

// dynamic LDS: the chunk prefix of the instance when it fits (lds_table != 0)
__global__ __launch_bounds__(1024) void k_vote_plan(const float* __restrict__ vertex, int64_t vs_n, int64_t vs_h,
                                                    int64_t vs_w, int64_t vs_c, const uint8_t* __restrict__ keep,
                                                    int W, int HW, int nch, int n, const int32_t* __restrict__ n_dev,
                                                    int hn, int hnp, const int32_t* __restrict__ idxs, uint64_t seed,
                                                    int min_num, int max_num, float efac, float wh, int lds_table,
                                                    int32_t* __restrict__ chunk_fg, int32_t* __restrict__ chunk_pre,
                                                    uint32_t* __restrict__ word_pre, uint64_t* __restrict__ bits,
                                                    int32_t* __restrict__ plan, float* __restrict__ hx,
                                                    float* __restrict__ hy, float* __restrict__ eg,
                                                    float* __restrict__ hyp, int32_t* __restrict__ upper) {
    extern __shared__ __attribute__((aligned(16))) int s_cpre[];     // [nch + 1] when lds_table
    __shared__ int s_w[20];
    const int n_act = active_instances(n, n_dev);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave, nw = blockDim.x / kWave;
    for (int inst = blockIdx.x; inst < n_act; inst += gridDim.x) {
        int32_t* cfg = chunk_fg + (size_t)inst * nch;
        int32_t* cpre = chunk_pre + (size_t)inst * (nch + 1);
        uint32_t* wpre = word_pre + (size_t)inst * nch * kChunkWords;
        uint64_t* bw = bits + (size_t)inst * nch * kChunkWords;
        for (int h = threadIdx.x; h < hnp; h += blockDim.x) upper[(size_t)inst * hnp + h] = 0;
        const int fg = block_scan_chunks(cfg, cpre, lds_table ? s_cpre : nullptr, nch, s_w);
        int tn = fg;
        if (fg > max_num) {
            // RV/ransac_voting_gpu.py:541-545: keep each foreground pixel with probability max_num / fg (injected
            // selection, or the counter-based stream of include/fpc_rng.h).  One wave per chunk, one lane per word.
            for (int c = wv; c < nch; c += nw) {
                const size_t wi = (size_t)c * kChunkWords + lane;
                uint64_t rem = bw[wi], kept = 0;
                while (rem) {
                    const int b = __ffsll((long long)rem) - 1;
                    rem &= rem - 1;
                    const int p = (int)(wi * 64) + b;
                    const bool k = keep ? (keep[(size_t)inst * HW + p] != 0)
                                        : (fpc_rand_keep(seed, (uint32_t)inst, (uint32_t)p, (uint32_t)fg, (uint32_t)max_num) != 0);
                    if (k) kept |= 1ull << b;
                }
                int tot;
                const int ex = wave_excl_scan(__popcll(kept), tot);
                bw[wi] = kept;
                wpre[wi] = (uint32_t)ex;
                if (lane == 0) cfg[c] = tot;
            }
            __syncthreads();
            tn = block_scan_chunks(cfg, cpre, lds_table ? s_cpre : nullptr, nch, s_w);
        }
        if (fg < min_num) tn = 0;      // :536-539
        if (threadIdx.x == 0) {
            plan[inst * kPlanI + 0] = fg;
            plan[inst * kPlanI + 1] = tn;
            plan[inst * kPlanI + 2] = (tn + kBlockPx - 1) / kBlockPx;
        }
        __syncthreads();               // the look-up tables of this instance are complete (same CU: visible)

        Lut L{lds_table ? s_cpre : cpre, wpre, bw, nch};
        const float* v = vertex + (int64_t)inst * vs_n;
        for (int h0 = 0; h0 < hnp; h0 += blockDim.x) {       // uniform trip count; a wave holds one group of 64
            const int hi = h0 + threadIdx.x;
            float x = 0.0f, y = 0.0f, e = 0.0f;
            if (hi < hn) {
                if (tn > 0) {
                    int t0, t1;
                    if (idxs) {
                        t0 = idxs[((size_t)inst * hn + hi) * 2];
                        t1 = idxs[((size_t)inst * hn + hi) * 2 + 1];
                    } else {
                        t0 = fpc_rand_index(seed, (uint32_t)inst, (uint32_t)hi, 0u, (uint32_t)tn);
                        t1 = fpc_rand_index(seed, (uint32_t)inst, (uint32_t)hi, 1u, (uint32_t)tn);
                    }
                    if (t0 >= 0 && t0 < tn && t1 >= 0 && t1 < tn) {      // the reference reads out of bounds here
                        const int p0 = lookup_pixel(L, t0), p1 = lookup_pixel(L, t1);
                        const float4 a = gather_pixel(v, vs_h, vs_w, vs_c, W, p0);
                        const float4 b = gather_pixel(v, vs_h, vs_w, vs_c, W, p1);
                        intersect(a, b, x, y);
                    }
                }
                hyp[((size_t)inst * hn + hi) * 2] = x;
                hyp[((size_t)inst * hn + hi) * 2 + 1] = y;
                const float s = fabsf(x) + fabsf(y);
                if (s <= 1e18f) e = efac * (s + wh);          // false for inf / NaN: outside the filter's domain
            }
            if (hi < hnp) {
                hx[(size_t)inst * hnp + hi] = x;
                hy[(size_t)inst * hnp + hi] = y;
                float m = e;
#pragma unroll
                for (int o = kWave / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, kWave));
                if (lane == 0) eg[(size_t)inst * (hnp / kWave) + hi / kWave] = m;
            }
        }
        __syncthreads();               // s_w / s_cpre are reused by the next instance
    }
}

// ---- pair classification shared by k_vote_count<kModeFiltered> and k_vote_refine ---------------------------------
// cst = {a_s = ey, b_s = -ex, c_s, ct} (c_s NaN: the pixel never votes).  Returns, per lane, whether the reference
// accepts (pixel, hypothesis): two FMA pairs and two compares decide unless the pair lies between the cones; those
// lanes (wave-uniform branch, rare) run the reference's own arithmetic on the raw pixel q = {x, y, dx, dy}.
struct Cones { float kappa1, kappa2; };

template <typename LoadQ>
__device__ __forceinline__ bool classify_pair(const float4 cst, const Cones k, float gx, float gy, float E, bool wild,
                                              bool valid, float thresh, LoadQ load_q) {
    const float ss = fabsf(__builtin_fmaf(cst.x, gx, __builtin_fmaf(cst.y, gy, cst.z)));
    const float tt = __builtin_fmaf(-cst.y, gx, __builtin_fmaf(cst.x, gy, cst.w));
    bool sure = ss <= __builtin_fmaf(k.kappa2, tt, -E);
    bool band = !sure && (ss <= __builtin_fmaf(k.kappa1, tt, E));
    if (wild) { sure = false; band = valid; }                   // wave-uniform: outside the filter's domain
    if (__builtin_amdgcn_ballot_w64(band)) {
        if (band) {
            const float4 q = load_q();
            const float n1 = sqrtf(q.z * q.z + q.w * q.w);
            sure = pair_is_inlier(q.x, q.y, q.z, q.w, n1, gx, gy, thresh);
        }
    }
    return sure;
}

// ---- k_vote_count ------------------------------------------------------------------
// Task t -> (instance, pixel block, hypothesis slice).  The slices of one block differ by 8 in t, i.e. they run
// on one XCD under round-robin dispatch (speed only).  grid-stride; 256 threads.
// dynamic LDS: [gps * 64] counts of the slice, then the instance's chunk prefix [nch + 1] when lds_table.
enum { kModeUpper = 0, kModeExact = 1, kModeFiltered = 2 };

template <int MODE>
__global__ __launch_bounds__(256) void k_vote_count(const float* __restrict__ vertex, int64_t vs_n, int64_t vs_h,
                                                    int64_t vs_w, int64_t vs_c, int W, int HW, int nch, int n,
                                                    const int32_t* __restrict__ n_dev, int hn, int hnp, int nb_grid,
                                                    int S, int gps /* groups of 64 hypotheses per slice */,
                                                    float kappa1, float kappa2, float efac_ref, float wh, float thresh,
                                                    int lds_table, const int32_t* __restrict__ chunk_pre,
                                                    const uint32_t* __restrict__ word_pre,
                                                    const uint64_t* __restrict__ bits, const int32_t* __restrict__ plan,
                                                    const float* __restrict__ hx, const float* __restrict__ hy,
                                                    const float* __restrict__ eg, int32_t* __restrict__ counts,
                                                    int cstride, float4* __restrict__ list, float4* __restrict__ clist) {
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];
    int* s_cnt = s_dyn;                        // [gps * 64]
    int* s_cpre = s_dyn + gps * kWave;         // [nch + 1]
    const int n_act = active_instances(n, n_dev);
    const long long units = (long long)n_act * nb_grid;
    const long long total = (units + 7) / 8 * 8 * S;
    const int lane = threadIdx.x & (kWave - 1);
    const int ngroups = hnp / kWave;
    const Cones cones{kappa1, kappa2};
    int cached_inst = -1;
    for (long long t = blockIdx.x; t < total; t += gridDim.x) {
        const long long grp = t / (8 * S);
        const int rem = (int)(t - grp * (8 * S));
        const int s = rem >> 3;
        const long long u = grp * 8 + (rem & 7);
        if (u >= units) continue;
        const int inst = (int)(u / nb_grid), b0 = (int)(u - (long long)inst * nb_grid);
        const int tn = plan[inst * kPlanI + 1];
        const int g_lo = s * gps, g_hi = min(ngroups, g_lo + gps);
        if (g_lo >= g_hi || b0 * kBlockPx >= tn) continue;          // uniform
        if (lds_table && cached_inst != inst) {                     // uniform
            __syncthreads();
            for (int i = threadIdx.x; i <= nch; i += blockDim.x) s_cpre[i] = chunk_pre[(size_t)inst * (nch + 1) + i];
            cached_inst = inst;
            __syncthreads();
        }
        const Lut L{lds_table ? s_cpre : chunk_pre + (size_t)inst * (nch + 1), word_pre + (size_t)inst * nch * kChunkWords,
                    bits + (size_t)inst * nch * kChunkWords, nch};
        // an instance normally has at most nb_grid blocks; one that keeps more pixels (an injected selection) wraps around
        for (int b = b0; b * kBlockPx < tn; b += nb_grid) {
            // this lane's pixel
            const int j = b * kBlockPx + threadIdx.x;
            const bool valid = j < tn;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            float n1 = 0.0f;
            const float qnan = __builtin_nanf("");
            float4 cst = make_float4(0.f, 0.f, qnan, qnan);         // NaN: the slot never counts
            if (valid) {
                q = gather_pixel(vertex + (int64_t)inst * vs_n, vs_h, vs_w, vs_c, W, lookup_pixel(L, j));
                n1 = sqrtf(q.z * q.z + q.w * q.w);
                if (MODE != kModeExact && !below_eps(n1) && n1 <= 3.0e38f) {
                    const float ex = q.z / n1, ey = q.w / n1;
                    cst.x = ey; cst.y = -ex;
                    cst.z = -(q.x * ey - q.y * ex);                 // s = d x e = ey gx - ex gy + cs
                    cst.w = -(q.x * ex + q.y * ey);                 // t = d . e = ex gx + ey gy + ct
                }
                if (s == 0 && list) {
                    list[(size_t)inst * HW + j] = q;
                    if (MODE != kModeExact) clist[(size_t)inst * HW + j] = cst;
                }
            }
            // folded constants of the upper-bound test: |ss| <= (kappa ex) gx + (kappa ey) gy + (kappa ct + E_g)
            const float a_s = cst.x, b_s = cst.y, c_s = cst.z;
            const float a_t = kappa1 * -cst.y, b_t = kappa1 * cst.x, c_t0 = kappa1 * cst.w;
            for (int i = threadIdx.x; i < (g_hi - g_lo) * kWave; i += blockDim.x) s_cnt[i] = 0;
            __syncthreads();

            for (int G = g_lo; G < g_hi; ++G) {
                const float* HX = static_cast<const float*>(__builtin_assume_aligned(hx + (size_t)inst * hnp + (size_t)G * kWave, 256));
                const float* HY = static_cast<const float*>(__builtin_assume_aligned(hy + (size_t)inst * hnp + (size_t)G * kWave, 256));
                const float c_t = c_t0 + eg[(size_t)inst * ngroups + G];
                int cntv = 0;
                // one (64-pixel tile, hypothesis) step: scalar-loaded point, 4 FMA + 1 compare, ballot + s_bcnt1, and the
                // count dropped into lane g of cntv by v_writelane (no builtin for it in ROCm 7.2's clang; the lane select is
                // an immediate, the count an SGPR written by SALU: no wait states required)
#define FPC_VOTE_STEP(g)                                                                                               \
                {                                                                                                      \
                    const float gx = HX[(g)], gy = HY[(g)];                                                            \
                    bool in;                                                                                           \
                    if (MODE == kModeExact) {                                                                          \
                        in = valid && pair_is_inlier(q.x, q.y, q.z, q.w, n1, gx, gy, thresh);                          \
                    } else if (MODE == kModeFiltered) {                                                                \
                        const float sabs = fabsf(gx) + fabsf(gy);                                                      \
                        const bool wild = !(sabs <= 1e18f);                                                            \
                        in = classify_pair(cst, cones, gx, gy, efac_ref * (sabs + wh), wild, valid, thresh,            \
                                           [&]() { return q; });                                                       \
                    } else {                                                                                           \
                        const float ss = __builtin_fmaf(a_s, gx, __builtin_fmaf(b_s, gy, c_s));                        \
                        const float th = __builtin_fmaf(a_t, gx, __builtin_fmaf(b_t, gy, c_t));                        \
                        in = fabsf(ss) <= th;                                                                          \
                    }                                                                                                  \
                    const int c = __popcll(__builtin_amdgcn_ballot_w64(in));                                           \
                    asm("v_writelane_b32 %0, %1, %2" : "+v"(cntv) : "s"(c), "n"(g));                                   \
                }
#define FPC_VOTE_STEP4(g) FPC_VOTE_STEP(g) FPC_VOTE_STEP((g) + 1) FPC_VOTE_STEP((g) + 2) FPC_VOTE_STEP((g) + 3)
#define FPC_VOTE_STEP16(g) FPC_VOTE_STEP4(g) FPC_VOTE_STEP4((g) + 4) FPC_VOTE_STEP4((g) + 8) FPC_VOTE_STEP4((g) + 12)
                FPC_VOTE_STEP16(0) FPC_VOTE_STEP16(16) FPC_VOTE_STEP16(32) FPC_VOTE_STEP16(48)
#undef FPC_VOTE_STEP16
#undef FPC_VOTE_STEP4
#undef FPC_VOTE_STEP
                atomicAdd(&s_cnt[(G - g_lo) * kWave + lane], cntv);   // LDS: the four waves' tiles of this block
            }
            __syncthreads();
            // one integer atomic per (block, hypothesis) with any count: 256-byte wave rows, order-independent result
            for (int i = threadIdx.x; i < (g_hi - g_lo) * kWave; i += blockDim.x) {
                const int h = g_lo * kWave + i, v = s_cnt[i];
                if (v && h < hn) atomicAdd(&counts[(size_t)inst * cstride + h], v);
            }
            __syncthreads();
        }
    }
}

// ---- k_vote_refine -----------------------------------------------------------------
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

constexpr int kRefThreads = 1024;
constexpr int kRefWaves = kRefThreads / kWave;
constexpr int kRefK = 4;             // candidates counted exactly per pass over the pixels

// block-wide (max value, lowest index) over s_u[0..hn); entries < 0 are consumed.  Result in every thread.
__device__ __forceinline__ void block_argmax_lds(const int* s_u, int hn, int* s_red, int& bc, int& bi) {
    bc = -1; bi = 0x7fffffff;
    for (int h = threadIdx.x; h < hn; h += blockDim.x) {
        const int c = s_u[h];
        if (c > bc) { bc = c; bi = h; }                               // ascending h: first maximum kept
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const int oc = __shfl_xor(bc, o, kWave), oi = __shfl_xor(bi, o, kWave);
        if (oc > bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    }
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    __syncthreads();
    if (lane == 0) { s_red[w] = bc; s_red[kRefWaves + w] = bi; }
    __syncthreads();
    bc = s_red[0]; bi = s_red[kRefWaves];
    for (int i = 1; i < kRefWaves; ++i) {
        const int oc = s_red[i], oi = s_red[kRefWaves + i];
        if (oc > bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    }
}

// FAST: the two-cone classification (threshold inside the filter's domain); otherwise the reference's arithmetic only.
template <bool FAST>
__global__ __launch_bounds__(kRefThreads) void k_vote_refine(int HW, int n, const int32_t* __restrict__ n_dev, int hn,
                                                             int hnp, float thresh, float kappa1, float kappa2,
                                                             float efac_ref, float wh, const int32_t* __restrict__ plan,
                                                             const float* __restrict__ hyp,
                                                             const int32_t* __restrict__ upper,
                                                             const float4* __restrict__ list,
                                                             const float4* __restrict__ clist,
                                                             float* __restrict__ out_xy, int32_t* __restrict__ out_tn,
                                                             int32_t* __restrict__ out_win_idx,
                                                             int32_t* __restrict__ out_win_count,
                                                             int32_t* __restrict__ out_inl,
                                                             int32_t* __restrict__ out_upper,
                                                             int32_t* __restrict__ out_evals) {
    extern __shared__ __attribute__((aligned(16))) int s_u[];           // [hnp]
    __shared__ int s_red[2 * kRefWaves * kRefK];
    __shared__ double s_sum[kRefWaves][5];
    const int n_act = active_instances(n, n_dev);
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    const Cones cones{kappa1, kappa2};
    for (int inst = blockIdx.x; inst < n_act; inst += gridDim.x) {
        const int tn = plan[inst * kPlanI + 1];
        if (tn == 0) {                                                 // uniform; RV/ransac_voting_gpu.py:536-539
            if (threadIdx.x == 0) {
                out_xy[inst * 2] = 0.0f; out_xy[inst * 2 + 1] = 0.0f;
                if (out_tn) out_tn[inst] = 0;
                if (out_win_idx) out_win_idx[inst] = -1;
                if (out_win_count) out_win_count[inst] = 0;
                if (out_inl) out_inl[inst] = 0;
                if (out_evals) out_evals[inst] = 0;
            }
            if (out_upper) for (int h = threadIdx.x; h < hn; h += blockDim.x) out_upper[(size_t)inst * hn + h] = 0;
            continue;
        }
        const float* hp = hyp + (size_t)inst * hn * 2;
        for (int h = threadIdx.x; h < hnp; h += blockDim.x) {
            int u = -1;
            if (h < hn) {
                u = upper[(size_t)inst * hnp + h];
                if (FAST) {                                            // outside the filter's domain: trivial bound
                    const float sabs = fabsf(hp[2 * h]) + fabsf(hp[2 * h + 1]);
                    if (!(sabs <= 1e18f)) u = tn;
                }
                if (out_upper) out_upper[(size_t)inst * hn + h] = u;
            }
            s_u[h] = u;
        }
        __syncthreads();

        const float4* P = list + (size_t)inst * HW;
        const float4* C = clist + (size_t)inst * HW;
        // exact inlier decision of pixel j for point (gx, gy)
        auto accepts = [&](int j, float gx, float gy, float E, bool wild) -> bool {
            if (FAST) return classify_pair(C[j], cones, gx, gy, E, wild, true, thresh, [&]() { return P[j]; });
            const float4 q = P[j];
            return pair_is_inlier(q.x, q.y, q.z, q.w, sqrtf(q.z * q.z + q.w * q.w), gx, gy, thresh);
        };

        // candidates in order (U desc, index asc), kRefK per pass, counted exactly until none left can win:
        // torch.max's winner (largest count, first index, RV/ransac_voting_gpu.py:567)
        int best_c = -1, best_i = 0x7fffffff, evals = 0;
        for (;;) {
            int cand[kRefK], nc = 0;
#pragma unroll
            for (int k = 0; k < kRefK; ++k) {
                cand[k] = -1;
                if (nc == k) {                                         // uniform
                    int uc, ui;
                    block_argmax_lds(s_u, hn, s_red, uc, ui);
                    if (uc >= 0 && (best_c < 0 || uc > best_c || (uc == best_c && ui < best_i))) {
                        cand[k] = ui; nc = k + 1;
                        __syncthreads();
                        if (threadIdx.x == 0) s_u[ui] = -1;            // consumed
                        __syncthreads();
                    }
                }
            }
            if (nc == 0) break;
            float gx[kRefK], gy[kRefK], E[kRefK];
            bool wild[kRefK];
            int cnt[kRefK];
#pragma unroll
            for (int k = 0; k < kRefK; ++k) {
                const int ci = cand[k] < 0 ? cand[0] : cand[k];
                gx[k] = hp[2 * ci]; gy[k] = hp[2 * ci + 1];
                const float sabs = fabsf(gx[k]) + fabsf(gy[k]);
                wild[k] = !(sabs <= 1e18f);
                E[k] = efac_ref * (sabs + wh);
                cnt[k] = 0;
            }
            // whole waves run the loop (the classification ballots): the tail lanes carry an inert slot
            for (int j0 = w * kWave; j0 < tn; j0 += kRefThreads) {
                const int j = j0 + lane;
                const bool valid = j < tn;
                const int jj = valid ? j : tn - 1;
#pragma unroll
                for (int k = 0; k < kRefK; ++k)
                    if (k < nc) cnt[k] += (accepts(jj, gx[k], gy[k], E[k], wild[k]) && valid) ? 1 : 0;
            }
#pragma unroll
            for (int k = 0; k < kRefK; ++k) {
#pragma unroll
                for (int o = kWave / 2; o > 0; o >>= 1) cnt[k] += __shfl_xor(cnt[k], o, kWave);
            }
            __syncthreads();
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < kRefK; ++k) s_red[k * kRefWaves + w] = cnt[k];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < kRefK; ++k) {
                if (k >= nc) continue;
                int tot = 0;
                for (int i = 0; i < kRefWaves; ++i) tot += s_red[k * kRefWaves + i];
                if (tot > best_c || (tot == best_c && cand[k] < best_i)) { best_c = tot; best_i = cand[k]; }
            }
            evals += nc;
            __syncthreads();
        }

        // vote again with the winner, least squares over its inliers (RV/ransac_voting_gpu.py:583-599).  No hypothesis
        // with an inlier: all_win_pts stays (0,0) (:571-574) and the refinement votes for (0,0).
        float wx = 0.0f, wy = 0.0f;
        if (best_c > 0) { wx = hp[2 * best_i]; wy = hp[2 * best_i + 1]; } else { best_i = -1; best_c = 0; }
        const float wabs = fabsf(wx) + fabsf(wy);
        const bool wwild = !(wabs <= 1e18f);
        const float wE = efac_ref * (wabs + wh);
        int inl = 0;
        double v[5] = {0, 0, 0, 0, 0};                                 // a00, a01, a11, b0, b1
        for (int j0 = w * kWave; j0 < tn; j0 += kRefThreads) {
            const int j = j0 + lane;
            const bool valid = j < tn;
            const int jj = valid ? j : tn - 1;
            if (accepts(jj, wx, wy, wE, wwild) && valid) {
                const float4 q = P[jj];
                const double nx = (double)q.w, ny = -(double)q.z;      // normal = (dy, -dx) :584-586
                const double bb = nx * (double)q.x + ny * (double)q.y;
                ++inl; v[0] += nx * nx; v[1] += nx * ny; v[2] += ny * ny; v[3] += nx * bb; v[4] += ny * bb;
            }
        }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) inl += __shfl_xor(inl, o, kWave);
#pragma unroll
        for (int a = 0; a < 5; ++a) {
            const double r = wave_reduce_add(v[a]);
            if (lane == 0) s_sum[w][a] = r;
        }
        if (lane == 0) s_red[w] = inl;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t[5] = {0, 0, 0, 0, 0};
            int it = 0;
            for (int i = 0; i < kRefWaves; ++i) {                      // fixed order: bit-reproducible
                it += s_red[i];
                for (int a = 0; a < 5; ++a) t[a] += s_sum[i][a];
            }
            double x0, x1;
            solve2_sym(t[0], t[1], t[2], t[3], t[4], x0, x1);
            out_xy[inst * 2] = (float)x0;
            out_xy[inst * 2 + 1] = (float)x1;
            if (out_tn) out_tn[inst] = tn;
            if (out_win_idx) out_win_idx[inst] = best_i;
            if (out_win_count) out_win_count[inst] = best_c;
            if (out_inl) out_inl[inst] = it;
            if (out_evals) out_evals[inst] = evals;
        }
        __syncthreads();
    }
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
                                    int32_t* out_counts, int32_t* out_upper, int32_t* out_evals, void* ws,
                                    size_t ws_bytes, fpc_stream_t stream) {
    if (n < 0 || H < 1 || W < 1 || hn < 1 || hn > kMaxHn || max_num < 1) return FPC_EINVAL;
    if ((int64_t)H * W > (1 << 30)) return FPC_EINVAL;
    if (n == 0) return FPC_OK;
    if (!mask || !vertex || !out_xy || !ws) return FPC_EINVAL;
    if (n > 65535) return FPC_EINVAL;
    if (((uintptr_t)ws & 255) != 0) return FPC_EWORKSPACE;
    Ws w = carve(ws, n, H, W, hn);
    if (ws_bytes < w.total) return FPC_EWORKSPACE;
    clear_hip_error();
    hipStream_t s = (hipStream_t)stream;
    const int HW = H * W;

    // 1. mask planes -> bit image (the only pass over the masks)
    const bool vec4 = (HW % 4 == 0) && (((uintptr_t)mask & 15) == 0);
    const int scan_grid = (int)std::min<long long>((long long)n * w.nch, 4096);
    if (vec4)
        hipLaunchKernelGGL(k_vote_scan<true>, dim3(scan_grid), dim3(256), 0, s, mask, HW, w.nch, n, n_dev, w.bits, w.word_pre, w.chunk_fg);
    else
        hipLaunchKernelGGL(k_vote_scan<false>, dim3(scan_grid), dim3(256), 0, s, mask, HW, w.nch, n, n_dev, w.bits, w.word_pre, w.chunk_fg);

    // the cones need th' = th - 1e-6 > 0; otherwise every pair takes the reference's arithmetic
    const bool fast = inlier_thresh > 2e-6f && inlier_thresh < 3.0e38f;
    float kappa1 = 0.0f, kappa2 = 0.0f;
    if (fast) {
        const double th1 = (double)inlier_thresh - 1e-6, th2 = (double)inlier_thresh + 1e-6;
        const double k1 = 1.0 - th1 * th1, k2 = 1.0 - th2 * th2;
        kappa1 = (float)((k1 > 0.0 ? sqrt(k1) : 0.0) / th1) * (1.0f + 1e-6f);                       // wider
        kappa2 = (th2 < 1.0 && k2 > 0.0) ? (float)(sqrt(k2) / th2) * (1.0f - 1e-6f) : 0.0f;        // narrower (0: no "sure")
    }
    const float efac_cnt = 1e-6f * (1.0f + kappa1), efac_ref = 2e-6f * (1.0f + kappa1);
    const float wh = (float)(W + H);
    const int lds_table = w.nch + 1 <= 4096 ? 1 : 0;              // chunk prefix of an instance in LDS (16 KB)
    const size_t table_lds = lds_table ? (size_t)(w.nch + 1) * sizeof(int) : 0;

    // 2. per instance: prefix, thinning, hypotheses, zeroed count row
    hipLaunchKernelGGL(k_vote_plan, dim3(std::min(n, 2048)), dim3(1024), table_lds, s, vertex, vs_n, vs_h, vs_w, vs_c, keep, W,
                       HW, w.nch, n, n_dev, hn, w.hnp, idxs, seed, min_num, max_num, efac_cnt, wh, lds_table, w.chunk_fg,
                       w.chunk_pre, w.word_pre, w.bits, w.plan, w.hx, w.hy, w.eg, w.hyp, w.upper);

    // 3. upper bounds (exact counts when the threshold has no cone)
    const int nb_launch = std::min(cdiv(HW, kBlockPx), cdiv(std::min(HW, max_num), kBlockPx) + 2);   // blocks per instance in the task grid
    const int ngroups = w.hnp / kWave;
    // hypothesis slices: enough waves to fill the chip when there are few instances (an instance typically fills
    // half of its blocks; with a device-side count the capacity n over-states the instances by ~8x)
    const long long waves_per_slice = std::max<long long>(1, (long long)(n_dev ? std::max(1, n / 8) : n) * nb_launch * 2);
    const int s_needed = (int)std::min<long long>(ngroups, std::max<long long>(1, (6144 + waves_per_slice - 1) / waves_per_slice));
    const int gps = cdiv(ngroups, s_needed);
    const int S = cdiv(ngroups, gps);
    const long long tasks = ((long long)n * nb_launch + 7) / 8 * 8 * S;
    const int count_grid = (int)std::min<long long>(tasks, 8192);            // a multiple of 8 either way
    const size_t count_lds = (size_t)gps * kWave * sizeof(int) + table_lds;
    auto launch_count = [&](int mode, int32_t* counts, int cstride, bool lists) {
#define FPC_LAUNCH_COUNT(M)                                                                                              \
        hipLaunchKernelGGL(k_vote_count<M>, dim3(count_grid), dim3(256), count_lds, s, vertex, vs_n, vs_h, vs_w, vs_c, W, HW,  \
                           w.nch, n, n_dev, hn, w.hnp, nb_launch, S, gps, kappa1, kappa2, efac_ref, wh, inlier_thresh, lds_table, \
                           w.chunk_pre, w.word_pre, w.bits, w.plan, w.hx, w.hy, w.eg, counts, cstride,                    \
                           lists ? w.list : nullptr, w.clist)
        if (mode == kModeUpper) FPC_LAUNCH_COUNT(kModeUpper);
        else if (mode == kModeExact) FPC_LAUNCH_COUNT(kModeExact);
        else FPC_LAUNCH_COUNT(kModeFiltered);
#undef FPC_LAUNCH_COUNT
    };
    launch_count(fast ? kModeUpper : kModeExact, w.upper, w.hnp, true);

    // 4. winner + refinement
    const size_t ref_lds = (size_t)w.hnp * sizeof(int);
    if (fast)
        hipLaunchKernelGGL(k_vote_refine<true>, dim3(std::min(n, 2048)), dim3(kRefThreads), ref_lds, s, HW, n, n_dev, hn, w.hnp,
                           inlier_thresh, kappa1, kappa2, efac_ref, wh, w.plan, w.hyp, w.upper, w.list, w.clist, out_xy, out_tn,
                           out_win_idx, out_win_count, out_inl_count, out_upper, out_evals);
    else
        hipLaunchKernelGGL(k_vote_refine<false>, dim3(std::min(n, 2048)), dim3(kRefThreads), ref_lds, s, HW, n, n_dev, hn, w.hnp,
                           inlier_thresh, kappa1, kappa2, efac_ref, wh, w.plan, w.hyp, w.upper, w.list, w.clist, out_xy, out_tn,
                           out_win_idx, out_win_count, out_inl_count, out_upper, out_evals);

    // diagnostics (never on the product path)
    if (out_hyp) {
        hipError_t e = hipMemcpyAsync(out_hyp, w.hyp, sizeof(float) * (size_t)n * hn * 2, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    }
    if (out_counts) {
        // the exact count of EVERY hypothesis, by the same two-cone classification the refinement uses
        // (rows past a device-side instance count are zeroed too)
        hipError_t e = hipMemsetAsync(out_counts, 0, sizeof(int32_t) * (size_t)n * hn, s);
        if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
        launch_count(fast ? kModeFiltered : kModeExact, out_counts, hn, false);
    }
    return check_launch();
}
