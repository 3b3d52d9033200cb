// ransac.hip — PVNet-derived RANSAC hough voting for gfx950.
//
//  * fpc_generate_hypothesis / fpc_voting_for_hypothesis: B1-compatible kernels
//    (reference: RV/src/ransac_voting_kernel.cu:11-167).
//  * fpc_ransac_voting_v3: the whole of ransac_voting_layer_v3 (RV/ransac_voting_gpu.py:518-607) for a batch of
//    instances in THREE launches (+ one 16-byte-granular memset of the arrival counters), without a host round trip and
//    without the hn x tn inlier matrix:
//
//      k_vote_scan    task = (instance, chunk of 4096 pixels): the only pass over the caller's planes.  Mask -> bit words,
//                     in-chunk prefix, chunk count and bounding box; the chunk's foreground pixels are compacted in raster
//                     order into the chunk's own slots of ONE float4 list {x, y, dx, dy} (vote gathered through the caller's
//                     strides).  The workgroup that arrives LAST for an instance (cdna_hip_programming.md Guideline 16:
//                     write-through stores, drained, one agent-scope ticket) runs the instance's PLAN in its tail: chunk
//                     prefix (rank -> slot), the integer origin / radius the filter's coordinates are measured from, the work
//                     units (blocks of <= 512 entries inside one chunk), the hn hypotheses (:552,559; pair sampling,
//                     two-line intersection exactly as .cu:28-45) — each also as a bf16 MFMA B-fragment of the filter —
//                     and a zeroed count row.
//      k_vote_count   one resident round of workgroups over units x hypothesis slices.  EXACT inlier counts.  The two affine
//                     forms of the filter run on the matrix cores in split precision (below); per (entry, hypothesis) the
//                     VALU does one subtraction and one v_alignbit that shifts TWO bits of the margin into a per-lane row;
//                     pairs the filter cannot decide (about 1 in 1000) are queued and take the reference's own arithmetic
//                     (.cu:106-125), 64 at a time.  Integer atomics per (unit, hypothesis).
//      k_vote_final   task = work unit: torch.max's winner (:567, first maximal index), its inliers voted again with the
//                     reference's arithmetic, fp64 normal-equation records; the unit of an instance that arrives last sums
//                     them in unit order and solves the 2x2 system in closed form (b_inv, :503-516, :583-599).
//
// The filter.  The reference accepts (pixel p, vote d, hypothesis g) when fl(cos) > th, where fl(cos) carries at most
// 8 ulp(1) < 1e-6 of rounding.  With e = d / |d|, D = g - p, t = D . e, s = D x e (both affine in g), and
// kappa(c) = sqrt(1 - c^2) / c:   accepted  =>  |s| <= kappa1 t  (kappa1 = kappa(th - 1e-6));
//                                 |s| <= kappa2 t  =>  accepted  (kappa2 = kappa(th + 1e-6)).
// Per hypothesis h the kernel evaluates ONE margin   r = sigma_h (kappa2 t - |s|) - ES_h   with
//   * E_h  = efac M_h >= the evaluation's own error in (kappa2 t - |s|), M_h = |gx - ox| + |gy - oy| + radius measured from
//            the instance's integer origin (error budget: k_vote_count's header), ES_h = bf16_up(sigma_h E_h),
//   * sigma_h = a bf16 value <= 2 / ((kappa1 - kappa2) T_h + 2.05 E_h), T_h >= max |g - p| >= t over the instance.
// Then   r >= 0  =>  |s| <= kappa2 t exactly  =>  accepted;      r < -2  =>  |s| > kappa1 t exactly  =>  rejected;
// and -2 <= r < 0 is undecided.  Bit 31 of r is "r < 0" and bit 30 is "|r| >= 2": v_alignbit(row, r, 30) appends both.
// Hypotheses that are huge or not finite get a fragment whose margin is -1 for every pixel (always undecided), padded
// hypotheses one whose margin is -4; entries that never vote (|d| < 1e-6 .cu:121, non-finite votes, thinned-out pixels,
// padding lanes) get |s| = 1e30.  Thresholds <= 2e-6 have no cone: every hypothesis is treated as "huge".
//
// Split precision.  F = a X + b Y + c S - [ES] with X = sigma (gx - ox), Y = sigma (gy - oy), S = sigma: each f32 factor is
// split EXACTLY into three bf16 pieces (v = v1 + v2 + v3, truncation split); a X keeps the six products
// a1X1 a1X2 a2X1 a2X2 a1X3 a3X1 (dropped: < 2^-23 |a X|), c S is exact (sigma is ONE bf16 piece), so K = 6 + 6 + 3 + 1 = 16:
// ONE v_mfma_f32_32x32x16_bf16 per form per 32 entries x 32 hypotheses, products exact, f32 accumulation.
//
// One RANSAC round: the reference's rounds re-evaluate identical samples.
#include <stdlib.h>

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

// grid (ceil(vn*tn/256), min(hn, 65535)), the hypothesis is uniform per block and walks the grid's y extent:
// consecutive lanes = consecutive pixels (coalesced coords/direct loads and u8 stores).
__global__ void k_b1_vote(const float* __restrict__ direct, const float* __restrict__ coords,
                          const float* __restrict__ hyp, uint8_t* __restrict__ inliers,
                          int tn, int vn, int hn, float thresh) {
    int vti = blockIdx.x * blockDim.x + threadIdx.x;
    if (vti >= vn * tn) return;
    int vi = vti / tn, ti = vti - vi * tn;
    float cx = coords[(size_t)ti * 2], cy = coords[(size_t)ti * 2 + 1];
    float nx = direct[(size_t)ti * vn * 2 + vi * 2], ny = direct[(size_t)ti * vn * 2 + vi * 2 + 1];
    float norm1 = sqrtf(nx * nx + ny * ny);
    for (int hi = blockIdx.y; hi < hn; hi += gridDim.y) {
        float hx = hyp[hi * vn * 2 + vi * 2], hy = hyp[hi * vn * 2 + vi * 2 + 1];
        if (pair_is_inlier(cx, cy, nx, ny, norm1, hx, hy, thresh)) inliers[((size_t)hi * vn + vi) * tn + ti] = 1;
    }
}


// ----------------------------------------------------------------------------
// fused v3

constexpr int kChunkPx = 4096;       // pixels per k_vote_scan task; a chunk owns list slots [c * 4096, c * 4096 + its count)
constexpr int kChunkWords = 64;      // 64-pixel words per chunk
constexpr int kUnitEntries = 512;    // list entries per work unit (4 waves x 2 groups of 64), always inside one chunk
constexpr int kUnitsPerChunk = kChunkPx / kUnitEntries;
constexpr int kHypTile = 32;         // hypotheses per MFMA tile
constexpr int kMaxSliceTiles = 64;   // hypothesis tiles per k_vote_count task at most (LDS count rows)
constexpr int kPlanI = 12;           // i32 per instance: fg, tn, thin, origin x, origin y, radius, units, votes
constexpr int kMaxHn = 65536;
constexpr int kRec = 6;              // doubles per refinement record: inliers, a00, a01, a11, b0, b1
constexpr int kBandQ = 320;          // queued undecided pairs per wave (one step adds at most 4 x 64)
constexpr float kNeverS = 1.0e30f;   // |s| of an entry that never votes

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long __attribute__((address_space(1))) gu64;
typedef unsigned __attribute__((address_space(1))) gu32;

// everything the three kernels share (passed by value)
struct VoteParams {
    // caller
    const float* mask; const float* vertex; int64_t vs_n, vs_h, vs_w, vs_c;
    int n; const int32_t* n_dev; int W, HW, hn;
    const int32_t* idxs; const uint8_t* keep; uint64_t seed; float thresh; int min_num, max_num;
    float* out_xy; int32_t* out_tn; int32_t* out_win_idx; int32_t* out_win_count; int32_t* out_inl; double* out_refine;
    // derived
    int nch, ntiles, hnp, nux, lds_table, want_tn, all_wild, task_target;
    size_t ls;                        // list slots per instance = nch * kChunkPx
    float kappa2, dkappa, efac;
    // workspace
    int32_t* ctrl;        // zeroed per call: [0] work units, [4, 4 + n) scan arrivals, [4 + n, 4 + 2n) final arrivals
    int32_t* plan;        // [n, kPlanI]
    int32_t* chunk_fg;    // [n, nch]       foreground count per chunk
    int32_t* chunk_box;   // [n, nch, 4]    x min / max, y min / max of the chunk's foreground pixels
    int32_t* chunk_pre;   // [n, nch + 1]   exclusive prefix of the counts (only used when the table does not fit LDS)
    int32_t* unit_pre;    // [n, nch + 1]   exclusive prefix of the work units per chunk                 (same condition)
    int32_t* kept_pre;    // [n, nch + 1]   the same over the KEPT entries (thinned instance with injected idxs / out_tn only)
    uint32_t* kept_wpre;  // [n, nch * 64]  kept entries before each 64-entry group inside its chunk     (same case)
    uint64_t* kept_bits;  // [n, nch * 64]  keep decisions of each 64-entry group                        (same case)
    float* hyp;           // [n, hn, 2]     hypothesis points as the reference's [hn,1,2] tensor
    u32x4* hypB;          // [n, ntiles, 64] their MFMA B fragments (lane = column + 32 * k-half)
    int32_t* counts;      // [n, hnp]       exact inlier count of every hypothesis; zeroed by the plan
    double* partial;      // [n, nux, kRec] k_vote_final per-unit records
    float4* list;         // [n, ls]        {x, y, dx, dy} of the foreground pixels, compacted per chunk (k_vote_scan)
    int4* units;          // [n * nux]      {instance, chunk, block | ordinal << 3, chunk count}
};

struct Ws {
    VoteParams p;
    size_t ctrl_bytes, total;
};

static Ws carve(void* base, int n, int H, int W, int hn) {
    Ws w;
    VoteParams& p = w.p;
    const size_t HW = (size_t)H * W;
    p.nch = cdiv((int)HW, kChunkPx);
    p.ntiles = cdiv(hn, kHypTile);
    p.hnp = p.ntiles * kHypTile;
    p.nux = p.nch * kUnitsPerChunk;
    p.ls = (size_t)p.nch * kChunkPx;
    char* b = (char*)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* q = b + off; off = align_up(off + bytes, 256); return q; };
    w.ctrl_bytes = align_up(sizeof(int32_t) * (4 + 2 * (size_t)n), 16);
    p.ctrl = (int32_t*)take(w.ctrl_bytes);
    p.plan = (int32_t*)take(sizeof(int32_t) * (size_t)n * kPlanI);
    p.chunk_fg = (int32_t*)take(sizeof(int32_t) * (size_t)n * p.nch);
    p.chunk_box = (int32_t*)take(sizeof(int32_t) * (size_t)n * p.nch * 4);
    p.chunk_pre = (int32_t*)take(sizeof(int32_t) * (size_t)n * (p.nch + 1));
    p.unit_pre = (int32_t*)take(sizeof(int32_t) * (size_t)n * (p.nch + 1));
    p.kept_pre = (int32_t*)take(sizeof(int32_t) * (size_t)n * (p.nch + 1));
    p.kept_wpre = (uint32_t*)take(sizeof(uint32_t) * (size_t)n * p.nch * kChunkWords);
    p.kept_bits = (uint64_t*)take(sizeof(uint64_t) * (size_t)n * p.nch * kChunkWords);
    p.hyp = (float*)take(sizeof(float) * (size_t)n * hn * 2);
    p.hypB = (u32x4*)take(sizeof(u32x4) * (size_t)n * p.ntiles * kWave);
    p.counts = (int32_t*)take(sizeof(int32_t) * (size_t)n * p.hnp);
    p.partial = (double*)take(sizeof(double) * (size_t)n * p.nux * kRec);
    p.list = (float4*)take(sizeof(float4) * (size_t)n * p.ls);
    p.units = (int4*)take(sizeof(int4) * (size_t)n * p.nux);
    w.total = off;
    return w;
}

__device__ __forceinline__ int active_instances(int n, const int32_t* __restrict__ n_dev) {
    if (!n_dev) return n;
    int m = *n_dev;
    return m < n ? (m < 0 ? 0 : m) : n;
}

// ---- write-through (sc1) accessors of the words that cross workgroups inside k_vote_scan / k_vote_final ----------------
__device__ __forceinline__ void store_wt(int32_t* p, int v) {
    __hip_atomic_store((gu32*)p, (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int load_wt(const int32_t* p) {
    return (int)__hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long load_wt64(const void* p) {
    return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_wt64(void* p, unsigned long long v) {
    __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16-byte write-through store of a list entry (the compiler does not see its vmcnt: the caller drains with s_waitcnt)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt_entry(float4* p, float4 v) {
    const f32x4 r = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(r) : "memory");
}
// a list entry another workgroup of this launch stored: two 8-byte sc1 loads (L1 bypassed)
__device__ __forceinline__ float4 load_wt_entry(const float4* p) {
    const unsigned long long a = load_wt64(p), b = load_wt64((const char*)p + 8);
    return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b),
                       __uint_as_float((unsigned)(b >> 32)));
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

// chunk holding rank e: largest c with cpre[c] <= e (cpre has nch + 1 entries, cpre[nch] > e)
__device__ __forceinline__ int rank_chunk(const int32_t* cpre, int nch, int e) {
    int lo = 0, hi = nch;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cpre[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

// Is pixel (x, y) of a THINNED instance kept (RV/ransac_voting_gpu.py:541-545)?
__device__ __forceinline__ bool pixel_kept(float x, float y, int W, int HW, int inst, int fg, int max_num, uint64_t seed,
                                           const uint8_t* __restrict__ keep) {
    const int p = (int)y * W + (int)x;
    return keep ? (keep[(size_t)inst * HW + p] != 0)
                : (fpc_rand_keep(seed, (uint32_t)inst, (uint32_t)p, (uint32_t)fg, (uint32_t)max_num) != 0);
}

// ---- split precision ------------------------------------------------------------------------------------------------
// v == p1 + p2 + p3 exactly; each piece has its 16 low bits clear (a bf16 value held in an f32)
__device__ __forceinline__ void split3(float v, float& p1, float& p2, float& p3) {
    p1 = __uint_as_float(__float_as_uint(v) & 0xffff0000u);
    const float r = v - p1;
    p2 = __uint_as_float(__float_as_uint(r) & 0xffff0000u);
    p3 = r - p2;
}
// two pieces -> one register: element 2j (low half) = lo, element 2j + 1 = hi
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
__device__ __forceinline__ float bf16_down(float v) { return __uint_as_float(__float_as_uint(v) & 0xffff0000u); }   // v >= 0
__device__ __forceinline__ float bf16_up(float v) {                                                                 // v >= 0
    const unsigned u = __float_as_uint(v);
    return __uint_as_float((u & 0xffffu) ? (u & 0xffff0000u) + 0x10000u : u);
}
// K slots of a form F = a X + b Y + c S + sg ES:   A (entry side)        B (hypothesis side)
//   0..5   a1 a1 a2 a2 a1 a3                        x  X1 X2 X1 X2 X3 X1
//   6, 7   c1 c2                                    x  S  S
//   8..13  b1 b1 b2 b2 b1 b3                        x  Y1 Y2 Y1 Y2 Y3 Y1
//   14     c3                                       x  S
//   15     sg                                       x  ES
// lanes 0-31 of a fragment hold slots 0-7 of row / column (lane & 31), lanes 32-63 slots 8-15.
__device__ __forceinline__ void a_fragment(float a, float b, float c, float sg, u32x4& lo, u32x4& hi) {
    float a1, a2, a3, b1, b2, b3, c1, c2, c3;
    split3(a, a1, a2, a3); split3(b, b1, b2, b3); split3(c, c1, c2, c3);
    lo = u32x4{pack2(a1, a1), pack2(a2, a2), pack2(a1, a3), pack2(c1, c2)};
    hi = u32x4{pack2(b1, b1), pack2(b2, b2), pack2(b1, b3), pack2(c3, sg)};
}
__device__ __forceinline__ void b_fragment(float X, float Y, float S, float ES, u32x4& lo, u32x4& hi) {
    float x1, x2, x3, y1, y2, y3;
    split3(X, x1, x2, x3); split3(Y, y1, y2, y3);
    lo = u32x4{pack2(x1, x2), pack2(x1, x2), pack2(x3, x1), pack2(S, S)};
    hi = u32x4{pack2(y1, y2), pack2(y1, y2), pack2(y3, y1), pack2(S, ES)};
}


// ---- the plan of one instance (tail of k_vote_scan) ------------------------------------------------------------------
// Exclusive scan of f(i), i in [0, cnt), into out[0..cnt] (out[cnt] = total) by the whole workgroup; out may be LDS or
// global, and f(i) may read out[i] (every thread reads its element before any thread of the tile writes).
template <typename F>
__device__ __forceinline__ int block_scan(F f, int32_t* out, int cnt, int* s_w /* >= 17 ints */) {
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave, nw = blockDim.x / kWave;
    int carry = 0;
    for (int base = 0; base < cnt; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const int v = i < cnt ? f(i) : 0;
        int wt;
        const int ex = wave_excl_scan(v, wt);
        __syncthreads();
        if (lane == 0) s_w[w] = wt;
        __syncthreads();
        int off = carry, tile = 0;
        for (int k = 0; k < nw; ++k) { const int x = s_w[k]; if (k < w) off += x; tile += x; }
        if (i < cnt) out[i] = off + ex;
        carry += tile;
    }
    if (threadIdx.x == 0) out[cnt] = carry;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // a global table: written before the barrier releases readers
    __syncthreads();
    return carry;
}

// Runs in the workgroup whose arrival completed instance `inst`: every chunk's count, box and list slots are in memory
// (write-through stores, drained before the arrival; read here with sc1 loads only).  s_tab: [3][nch + 1] ints of LDS when
// p.lds_table, else the tables live in the workspace.
// Out of line, and it reads the parameter block from the kernel-argument segment (constant address space: scalar loads)
// through a pointer the kernel hands it: the scan loop keeps its registers and no copy of the block is made for the call.
// (__builtin_amdgcn_kernarg_segment_ptr() is only meaningful inside the kernel function itself.)
typedef const VoteParams __attribute__((address_space(4)))* KParams;
__device__ __forceinline__ KParams kernel_params() {
#if defined(__HIP_DEVICE_COMPILE__)
    return (KParams)__builtin_amdgcn_kernarg_segment_ptr();               // the kernels' ONLY argument, at offset 0
#else
    return nullptr;                                                       // host pass: never called
#endif
}

__device__ __attribute__((noinline)) void plan_instance(KParams kp, int inst, int* s_tab, int* s_w, int* s_misc /* >= 8 ints */) {
    const auto& p = *kp;
    const int nch = p.nch, W = p.W, HW = p.HW, hn = p.hn;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave, nw = blockDim.x / kWave;
    const int32_t* cfg = p.chunk_fg + (size_t)inst * nch;
    int32_t* cpre = p.lds_table ? s_tab : p.chunk_pre + (size_t)inst * (nch + 1);
    int32_t* upre = p.lds_table ? s_tab + (nch + 1) : p.unit_pre + (size_t)inst * (nch + 1);
    int32_t* kpre = p.lds_table ? s_tab + 2 * (nch + 1) : p.kept_pre + (size_t)inst * (nch + 1);
    const float4* E = p.list + (size_t)inst * p.ls;

    for (int h = threadIdx.x; h < p.hnp; h += blockDim.x) p.counts[(size_t)inst * p.hnp + h] = 0;
    if (threadIdx.x < 4) s_misc[threadIdx.x] = (threadIdx.x & 1) ? -1 : 0x7fffffff;
    // bounding box of the instance -> the origin the filter's coordinates are measured from and the radius
    // max |x - ox| + |y - oy| of its pixels (both only scale the rounding allowance: any values are sound)
    int b0 = 0x7fffffff, b1 = -1, b2 = 0x7fffffff, b3 = -1;
    for (int c = threadIdx.x; c < nch; c += blockDim.x) {
        const unsigned long long lo = load_wt64(p.chunk_box + ((size_t)inst * nch + c) * 4);
        const unsigned long long hi = load_wt64(p.chunk_box + ((size_t)inst * nch + c) * 4 + 2);
        b0 = min(b0, (int)(unsigned)lo); b1 = max(b1, (int)(unsigned)(lo >> 32));
        b2 = min(b2, (int)(unsigned)hi); b3 = max(b3, (int)(unsigned)(hi >> 32));
    }
    const int fg = block_scan([&](int c) { return load_wt(cfg + c); }, cpre, nch, s_w);
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        b0 = min(b0, __shfl_xor(b0, o, kWave)); b1 = max(b1, __shfl_xor(b1, o, kWave));
        b2 = min(b2, __shfl_xor(b2, o, kWave)); b3 = max(b3, __shfl_xor(b3, o, kWave));
    }
    if (lane == 0 && b1 >= 0) { atomicMin(&s_misc[0], b0); atomicMax(&s_misc[1], b1); atomicMin(&s_misc[2], b2); atomicMax(&s_misc[3], b3); }
    __syncthreads();
    const int x0 = s_misc[0], x1 = s_misc[1], y0 = s_misc[2], y1 = s_misc[3];
    int ox = 0, oy = 0, rad = W + HW / W;
    if (x1 >= 0) {
        ox = (x0 + x1) / 2; oy = (y0 + y1) / 2;
        rad = max(x1 - ox, ox - x0) + max(y1 - oy, oy - y0);
    }
    const float fox = (float)ox, foy = (float)oy, frad = (float)rad;
    const bool thin = fg > p.max_num;
    // RV/ransac_voting_gpu.py:541-545.  The list keeps every foreground pixel; k_vote_count / k_vote_final re-derive each
    // entry's keep decision, and the built-in sampler draws over all foreground ranks and rejects thinned-out ones
    // (include/fpc_rng.h).  Only injected pair indices (they address the KEPT pixels by rank) and the out_tn diagnostic
    // need the kept image: one wave per chunk, one lane per entry.
    const bool tables = thin && (p.idxs != nullptr || p.want_tn);
    const uint32_t* kw = p.kept_wpre + (size_t)inst * nch * kChunkWords;
    const uint64_t* kb = p.kept_bits + (size_t)inst * nch * kChunkWords;
    int tn = thin ? p.max_num : fg;
    if (tables) {
        uint32_t* kww = p.kept_wpre + (size_t)inst * nch * kChunkWords;
        uint64_t* kbw = p.kept_bits + (size_t)inst * nch * kChunkWords;
        for (int c = wv; c < nch; c += nw) {
            const int cnt = cpre[c + 1] - cpre[c];
            int run = 0;
            for (int j = 0; j * kWave < cnt; ++j) {
                const int e = j * kWave + lane;
                bool k = false;
                if (e < cnt) {
                    const unsigned long long xy = load_wt64(E + (size_t)c * kChunkPx + e);
                    k = pixel_kept(__uint_as_float((unsigned)xy), __uint_as_float((unsigned)(xy >> 32)), W, HW, inst, fg,
                                   p.max_num, p.seed, p.keep);
                }
                const uint64_t m = __builtin_amdgcn_ballot_w64(k);
                if (lane == 0) {
                    store_wt64(kbw + (size_t)c * kChunkWords + j, m);
                    store_wt((int32_t*)kww + (size_t)c * kChunkWords + j, run);
                }
                run += __popcll(m);
            }
            if (lane == 0) kpre[c] = run;          // kept entries of the chunk; scanned in place below
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        tn = block_scan([&](int c) { return kpre[c]; }, kpre, nch, s_w);
    }
    const bool votes = fg >= p.min_num && tn > 0;     // :536-539
    // work units: blocks of <= 512 entries inside one chunk, in chunk order; the ordinal indexes k_vote_final's records
    const int nunits = block_scan([&](int c) { return (cpre[c + 1] - cpre[c] + kUnitEntries - 1) / kUnitEntries; }, upre, nch, s_w);
    if (threadIdx.x == 0) {
        int32_t* pl = p.plan + (size_t)inst * kPlanI;
        pl[0] = fg; pl[1] = votes ? tn : 0; pl[2] = thin ? 1 : 0; pl[3] = ox; pl[4] = oy; pl[5] = rad;
        pl[6] = votes ? nunits : 0; pl[7] = votes ? 1 : 0;
        s_misc[4] = (votes && nunits) ? atomicAdd(p.ctrl, nunits) : 0;
    }
    __syncthreads();
    if (!votes) {                                     // uniform: no unit; k_vote_final writes the zeros
        for (int i = threadIdx.x; i < 2 * hn; i += blockDim.x) p.hyp[(size_t)inst * hn * 2 + i] = 0.0f;   // the out_hyp diagnostic
        return;
    }
    const int ubase = s_misc[4];
    for (int c = threadIdx.x; c < nch; c += blockDim.x) {
        const int cnt = cpre[c + 1] - cpre[c], u0 = upre[c];
        for (int k = 0; k * kUnitEntries < cnt; ++k) p.units[ubase + u0 + k] = make_int4(inst, c, k | ((u0 + k) << 3), cnt);
    }

    // list slot of the t-th foreground pixel (raster order) / of the t-th KEPT one
    auto rank_slot = [&](int t) -> int {
        const int c = rank_chunk(cpre, nch, t);
        return c * kChunkPx + (t - cpre[c]);
    };
    auto kept_slot = [&](int t) -> int {
        const int c = rank_chunk(kpre, nch, t);
        const int r = t - kpre[c];
        int lo = 0, hi = (cpre[c + 1] - cpre[c] + kWave - 1) / kWave;      // groups of the chunk
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (load_wt((const int32_t*)kw + (size_t)c * kChunkWords + mid) <= r) lo = mid; else hi = mid;
        }
        const int before = load_wt((const int32_t*)kw + (size_t)c * kChunkWords + lo);
        return c * kChunkPx + lo * kWave + select64(load_wt64(kb + (size_t)c * kChunkWords + lo), r - before);
    };
    // built-in sampler (include/fpc_rng.h): uniform over the kept pixels by rejection
    auto draw_slot = [&](int hi, int which) -> int {
        int slot = 0;
        for (int a = 0; a < FPC_SAMPLE_MAX_TRIES; ++a) {
            slot = rank_slot(fpc_rand_index(p.seed, (uint32_t)inst, (uint32_t)hi, (uint32_t)(which + 2 * a), (uint32_t)fg));
            if (!thin) break;
            const unsigned long long xy = load_wt64(E + slot);
            if (pixel_kept(__uint_as_float((unsigned)xy), __uint_as_float((unsigned)(xy >> 32)), W, HW, inst, fg, p.max_num,
                           p.seed, p.keep))
                break;
        }
        return slot;
    };
    const float bx0 = (float)x0, bx1 = (float)x1, by0 = (float)y0, by1 = (float)y1;
    for (int h0 = 0; h0 < p.hnp; h0 += blockDim.x) {
        const int hi = h0 + threadIdx.x;
        if (hi >= p.hnp) break;
        // padded hypothesis: margin -4 for every entry (never counted, never undecided)
        float X = 0.0f, Y = 0.0f, S = 0.0f, ES = 4.0f;
        if (hi < hn) {
            float x = 0.0f, y = 0.0f;
            int s0 = -1, s1 = -1;
            if (p.idxs) {
                const int t0 = p.idxs[((size_t)inst * hn + hi) * 2], t1 = p.idxs[((size_t)inst * hn + hi) * 2 + 1];
                if (t0 >= 0 && t0 < tn && t1 >= 0 && t1 < tn) {        // the reference reads out of bounds here
                    s0 = thin ? kept_slot(t0) : rank_slot(t0);
                    s1 = thin ? kept_slot(t1) : rank_slot(t1);
                }
            } else {
                s0 = draw_slot(hi, 0);
                s1 = draw_slot(hi, 1);
            }
            if (s0 >= 0) intersect(load_wt_entry(E + s0), load_wt_entry(E + s1), x, y);
            p.hyp[((size_t)inst * hn + hi) * 2] = x;
            p.hyp[((size_t)inst * hn + hi) * 2 + 1] = y;
            const bool wild = p.all_wild || !(fabsf(x) + fabsf(y) <= 1e18f);      // inf / NaN / huge: outside the filter's domain
            if (wild) {
                ES = 1.0f;                                                           // margin -1: every pair undecided
            } else {
                const float xs = x - fox, ys = y - foy;
                const float M = fabsf(xs) + fabsf(ys) + frad;
                const float Eh = p.efac * M;
                const float dxm = fmaxf(fabsf(x - bx0), fabsf(x - bx1)), dym = fmaxf(fabsf(y - by0), fabsf(y - by1));
                const float T = sqrtf(dxm * dxm + dym * dym) * 1.000002f + 1e-3f;   // >= |g - p| for every pixel of the instance
                const float G = (p.dkappa * T + 2.05f * Eh) * 1.000002f;
                S = bf16_down(2.0f / fmaxf(G, 1e-3f));
                ES = bf16_up(S * Eh * 1.0001f);
                X = S * xs; Y = S * ys;
            }
        }
        u32x4 lo, hi4;
        b_fragment(X, Y, S, ES, lo, hi4);
        u32x4* B = p.hypB + ((size_t)inst * p.ntiles + hi / kHypTile) * kWave;
        B[hi % kHypTile] = lo;
        B[hi % kHypTile + kHypTile] = hi4;
    }
}

// ---- k_vote_scan -------------------------------------------------------------------
// grid-stride over (instance, chunk) tasks; 256 threads; a chunk = 4096 pixels = 4 float4 of the mask per lane.
// dynamic LDS: [3][nch + 1] ints when p.lds_table.
template <bool VEC4, bool VGATHER4>
__global__ __launch_bounds__(256) void k_vote_scan(const VoteParams p) {
    extern __shared__ __attribute__((aligned(16))) int s_tab[];
    __shared__ __attribute__((aligned(16))) uint8_t s_nib[kChunkPx / 4];
    __shared__ uint64_t s_word[kChunkWords];
    __shared__ int s_wpre[kChunkWords];
    __shared__ int s_box[4];
    __shared__ int s_w[20];
    __shared__ int s_misc[8];
    __shared__ int s_last;
    const int W = p.W, HW = p.HW, nch = p.nch;
    const int total = active_instances(p.n, p.n_dev) * nch;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int inst = t / nch, c = t - inst * nch;
        const float* m = p.mask + (size_t)inst * HW;
        unsigned nb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int fi = k * 256 + threadIdx.x;           // float4 index inside the chunk
            const int px = c * kChunkPx + fi * 4;
            nb[k] = 0;
            if (VEC4) {                                     // HW % 4 == 0 and a 16-byte aligned plane
                if (px < HW) {
                    const float4 v = *reinterpret_cast<const float4*>(m + px);
                    nb[k] = (v.x != 0.0f ? 1u : 0u) | (v.y != 0.0f ? 2u : 0u) | (v.z != 0.0f ? 4u : 0u) | (v.w != 0.0f ? 8u : 0u);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (px + q < HW && m[px + q] != 0.0f) nb[k] |= 1u << q;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) s_nib[k * 256 + threadIdx.x] = (uint8_t)nb[k];
        __syncthreads();
        if (threadIdx.x < kChunkWords) {
            const uint64_t word = pack_nibbles(*reinterpret_cast<const uint4*>(s_nib + 16 * threadIdx.x));
            int tot;
            const int ex = wave_excl_scan(__popcll(word), tot);
            s_word[threadIdx.x] = word;
            s_wpre[threadIdx.x] = ex;
            if (threadIdx.x == 0) {
                store_wt(p.chunk_fg + (size_t)inst * nch + c, tot);
                s_box[0] = 0x7fffffff; s_box[1] = -1; s_box[2] = 0x7fffffff; s_box[3] = -1;
            }
        }
        __syncthreads();
        int bx0 = 0x7fffffff, bx1 = -1, by0 = 0x7fffffff, by1 = -1;
        const float* v = p.vertex + (int64_t)inst * p.vs_n;
        float4* L = p.list + (size_t)inst * p.ls + (size_t)c * kChunkPx;
        // the votes of this lane's four 4-pixel groups: on the x-contiguous, 16-byte aligned layout (the reference's
        // permuted view of two planes) two float4 loads per group, predicated on the group having a foreground pixel and
        // issued together (one memory latency for all eight); any other layout gathers pixel by pixel
        float4 vx[4], vy[4];
        if (VGATHER4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                vx[k] = vy[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (nb[k]) {
                    const int px = c * kChunkPx + (k * 256 + threadIdx.x) * 4;
                    const int y = px / W, x = px - y * W;             // W % 4 == 0: the group stays in one row
                    const float* a = v + (int64_t)y * p.vs_h + x;
                    vx[k] = *reinterpret_cast<const float4*>(a);
                    vy[k] = *reinterpret_cast<const float4*>(a + p.vs_c);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!nb[k]) continue;
            const int fi = k * 256 + threadIdx.x;
            const int wq = fi >> 4, bit0 = (fi & 15) * 4;
            const uint64_t word = s_word[wq];
            const int wpre = s_wpre[wq];
            const int p0 = c * kChunkPx + fi * 4;
            const int y0 = p0 / W, x0 = p0 - y0 * W;
            const float gx4[4] = {vx[k].x, vx[k].y, vx[k].z, vx[k].w}, gy4[4] = {vy[k].x, vy[k].y, vy[k].z, vy[k].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!((nb[k] >> q) & 1u)) continue;
                const int bit = bit0 + q;
                const int pos = wpre + __popcll(word & ((1ull << bit) - 1ull));
                float4 e;
                if (VGATHER4) {
                    e = make_float4((float)(x0 + q), (float)y0, gx4[q], gy4[q]);
                } else {
                    const int px = p0 + q;
                    const int y = px / W, x = px - y * W;
                    const int64_t o = (int64_t)y * p.vs_h + (int64_t)x * p.vs_w;
                    e = make_float4((float)x, (float)y, v[o], v[o + p.vs_c]);
                }
                store_wt_entry(L + pos, e);
                const int xi = (int)e.x, yi = (int)e.y;
                bx0 = min(bx0, xi); bx1 = max(bx1, xi); by0 = min(by0, yi); by1 = max(by1, yi);
            }
        }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {                  // per wave first: same-address LDS atomics serialise
            bx0 = min(bx0, __shfl_xor(bx0, o, kWave)); bx1 = max(bx1, __shfl_xor(bx1, o, kWave));
            by0 = min(by0, __shfl_xor(by0, o, kWave)); by1 = max(by1, __shfl_xor(by1, o, kWave));
        }
        if ((threadIdx.x & (kWave - 1)) == 0 && bx1 >= 0) {
            atomicMin(&s_box[0], bx0); atomicMax(&s_box[1], bx1); atomicMin(&s_box[2], by0); atomicMax(&s_box[3], by1);
        }
        __syncthreads();
        if (threadIdx.x < 2)
            store_wt64(p.chunk_box + ((size_t)inst * nch + c) * 4 + 2 * threadIdx.x,
                       (unsigned long long)(unsigned)s_box[2 * threadIdx.x] | ((unsigned long long)(unsigned)s_box[2 * threadIdx.x + 1] << 32));
        // hand-off (Guideline 16, R1 with an arrival counter): every storing wave drains its write-through stores, the
        // workgroup meets, ONE lane adds the arrival; the workgroup whose add completes the instance runs its plan
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tk = __hip_atomic_fetch_add(p.ctrl + 4 + inst, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (tk == nch - 1);
        }
        __syncthreads();
        if (s_last) plan_instance(kernel_params(), inst, s_tab, s_w, s_misc);   // uniform
        __syncthreads();
    }
}

// ---- k_vote_count ------------------------------------------------------------------
// EXACT inlier count of every hypothesis.  Task -> (work unit, slice of hypothesis tiles); the slicing is chosen on the
// device from the unit count so that the tasks fill one resident round of workgroups.  256 threads; wave w owns the
// unit's 64-entry groups w and w + 4.  Per group the wave builds the A fragments of the two forms (prologue, once per
// task); per hypothesis tile it loads ONE B fragment (16 bytes per lane) and issues two MFMAs per 32 entries; the
// result registers hold, per lane, ONE hypothesis (column lane & 31) against 16 entries (rows), so counts stay lane-local:
//     r = F_t - |F_s| ;  row = (row << 2) | (r >> 30) ;  after 16: neg += popc(row & 0xAAAAAAAA), undecided = odd bit set & even clear.
// Error budget of r (units of the unscaled margin, M = |gx - ox| + |gy - oy| + radius, |e| = 1):
//     unit vote by v_rsq_f32 (1 ulp) and two products ............................ 4e-7 M   (s and t forms alike)
//     gx - ox, sigma (gx - ox): two roundings; c_s / c_t: three at <= radius ........ 3e-7 M
//     kappa2 e, kappa2 c_t: one rounding each ....................................... 1.2e-7 kappa M
//     dropped piece products ......................................................... 1.2e-7 M
//     f32 accumulation of 16 exact products inside the MFMA (any order, any rounding mode) ... <= 16 x 1.2e-7 M
//  => |r_computed - r_exact| / sigma <= 2.8e-6 (1 + kappa2) M  <  E = efac M  with efac = 3.2e-6 (1 + kappa1);
//     measured worst over 2 M random pairs: 4.2e-7 M (tools_dev/mfma_vote_probe.hip).
// dynamic LDS: [gps * 32] counts of the slice.

// the pairs the filter could not decide: code = hypothesis | entry-in-unit << 16; evaluated 64 at a time, one per lane
__device__ __attribute__((noinline)) void band_flush(int nq, const int* __restrict__ queue, const float4* __restrict__ U,
                                                     int nvalid, bool thin, int inst, int fg, KParams kp) {
    const auto& p = *kp;
    const int lane = threadIdx.x & (kWave - 1), hn = p.hn;
    const float* hyp = p.hyp + (size_t)inst * hn * 2;
    int32_t* cnt_row = p.counts + (size_t)inst * p.hnp;
    for (int base = 0; base < nq; base += kWave) {                           // uniform
        if (base + lane >= nq) continue;
        const int e = queue[base + lane];
        const int h = e & 0xffff, ent = e >> 16;
        if (ent >= nvalid || h >= hn) continue;
        const float4 q = U[ent];
        if (thin && !pixel_kept(q.x, q.y, p.W, p.HW, inst, fg, p.max_num, p.seed, p.keep)) continue;
        const float gx = hyp[2 * h], gy = hyp[2 * h + 1];
        if (pair_is_inlier(q.x, q.y, q.z, q.w, sqrtf(q.z * q.z + q.w * q.w), gx, gy, p.thresh)) atomicAdd(cnt_row + h, 1);
    }
}

struct GroupFrags { u32x4 s[2], t[2]; };          // A fragments of one 64-entry group: forms s / t, row tiles 0 / 1

// the A fragments of this lane's entry, exchanged so that tile 0 = entries 0-31 and tile 1 = entries 32-63 of the group
__device__ __forceinline__ void build_group(GroupFrags& g, bool valid, float4 q, float fox, float foy, float kappa2) {
    float a_s = 0.f, b_s = 0.f, c_s = kNeverS, a_t = 0.f, b_t = 0.f, c_t = 0.f;
    const float n2 = q.z * q.z + q.w * q.w;
    // .cu:121 skips a vote with |d| < 1e-6 (compared in double): n1 <= 1e-6f in f32 (common.hpp); near that bound the
    // correctly rounded sqrt decides.  A non-finite or overflowing |d|^2 never votes either (NaN / 0 cosine).
    bool votes = valid && n2 <= 3.0e38f && n2 >= 4.0e-12f;
    if (__builtin_amdgcn_ballot_w64(valid && n2 < 4.0e-12f)) votes = votes || (valid && n2 < 4.0e-12f && !below_eps(sqrtf(n2)));
    if (votes) {
        const float inv = __builtin_amdgcn_rsqf(n2);
        const float ex = q.z * inv, ey = q.w * inv;
        const float xs = q.x - fox, ys = q.y - foy;
        a_s = ey; b_s = -ex; c_s = -(xs * ey - ys * ex);
        a_t = kappa2 * ex; b_t = kappa2 * ey; c_t = kappa2 * -(xs * ex + ys * ey);
    }
    u32x4 slo, shi, tlo, thi;
    a_fragment(a_s, b_s, c_s, 0.0f, slo, shi);
    a_fragment(a_t, b_t, c_t, -1.0f, tlo, thi);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // lanes 32-63 of `lo` <-> lanes 0-31 of `hi`: lo' = slots 0-7 | 8-15 of entries 0-31, hi' = the same of entries 32-63
        const auto a = __builtin_amdgcn_permlane32_swap(slo[r], shi[r], false, false);
        g.s[0][r] = a[0]; g.s[1][r] = a[1];
        const auto b = __builtin_amdgcn_permlane32_swap(tlo[r], thi[r], false, false);
        g.t[0][r] = b[0]; g.t[1][r] = b[1];
    }
}

// one row tile (32 entries) against one hypothesis tile: two MFMAs, 16 x (v_sub, v_alignbit); returns the 2-bit rows
__device__ __forceinline__ unsigned tile_rows(const u32x4& As, const u32x4& At, const bf16x8 B) {
    f32x16 Fs = {0}, Ft = {0};
    Fs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, As), B, Fs, 0, 0, 0);
    Ft = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, At), B, Ft, 0, 0, 0);
    unsigned row = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) row = __builtin_amdgcn_alignbit(row, __float_as_uint(Ft[i] - fabsf(Fs[i])), 30);
    return row;
}

template <int WAVES /* waves per SIMD the register allocation aims at */>
__global__ __launch_bounds__(256, WAVES) void k_vote_count(const VoteParams p) {
    extern __shared__ __attribute__((aligned(16))) int s_cnt[];      // [gps * 32]
    __shared__ int s_bandq[4][kBandQ];
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const int nu = *p.ctrl;
    const int ntiles = p.ntiles;
    // units x S slices ~ the task count the launch was sized for: one round of equal tasks over the chip
    const int S0 = max(1, min(ntiles, p.task_target / max(nu, 1)));
    const int gps = min(kMaxSliceTiles, (ntiles + S0 - 1) / S0);
    const int S = (ntiles + gps - 1) / gps;
    const long long total = (long long)nu * S;
    int* bq = s_bandq[wv];
    for (long long t = blockIdx.x; t < total; t += gridDim.x) {
        const int u = (int)(t / S), s = (int)(t - (long long)u * S);
        const int4 ub = p.units[u];
        const int inst = ub.x, c = ub.y, k = ub.z & 7, cnt = ub.w;
        const int nvalid = min(kUnitEntries, cnt - k * kUnitEntries);
        const float4* U = p.list + (size_t)inst * p.ls + (size_t)c * kChunkPx + (size_t)k * kUnitEntries;
        const int4 pl0 = *reinterpret_cast<const int4*>(p.plan + (size_t)inst * kPlanI);           // fg, tn, thin, ox
        const int oy = p.plan[(size_t)inst * kPlanI + 4];
        const int fg = pl0.x;
        const bool thin = pl0.z != 0;
        const float fox = (float)pl0.w, foy = (float)oy;
        const int T0 = s * gps, T1 = min(ntiles, T0 + gps);
        // this wave's groups: w and w + 4 of the unit's eight
        const int ng = (wv * kWave < nvalid ? 1 : 0) + ((wv + 4) * kWave < nvalid ? 1 : 0);
        GroupFrags G[2];
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int e = (wv + 4 * gi) * kWave + lane;
            bool valid = e < nvalid;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid) q = U[e];
            if (valid && thin) valid = pixel_kept(q.x, q.y, p.W, p.HW, inst, fg, p.max_num, p.seed, p.keep);
            build_group(G[gi], valid, q, fox, foy, p.kappa2);
        }
        for (int i = threadIdx.x; i < (T1 - T0) * kHypTile; i += blockDim.x) s_cnt[i] = 0;
        __syncthreads();
        if (ng > 0) {                                                       // uniform per wave
            const u32x4* Bp = p.hypB + ((size_t)inst * ntiles + T0) * kWave + lane;
            int qn = 0;
            u32x4 Bn = *Bp;
            for (int T = T0; T < T1; ++T) {
                const bf16x8 B = __builtin_bit_cast(bf16x8, Bn);
                if (T + 1 < T1) Bn = Bp[(size_t)(T + 1 - T0) * kWave];
                unsigned bm[4] = {0u, 0u, 0u, 0u};
                int neg = 0;
#pragma unroll
                for (int gi = 0; gi < 2; ++gi) {
                    if (gi < ng) {
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) {
                            const unsigned row = tile_rows(G[gi].s[rt], G[gi].t[rt], B);
                            neg += __popc(row & 0xAAAAAAAAu);
                            bm[gi * 2 + rt] = (row >> 1) & ~row & 0x55555555u;
                        }
                    }
                }
                // lane-local: column (lane & 31) of tile T against 16 rows x 2 tiles x ng groups
                atomicAdd(&s_cnt[(T - T0) * kHypTile + (lane & 31)], ng * 32 - neg);
                // undecided pairs -> the wave's queue (usually a handful per step)
                while (__builtin_amdgcn_ballot_w64((bm[0] | bm[1] | bm[2] | bm[3]) != 0u)) {        // uniform
                    const bool has = (bm[0] | bm[1] | bm[2] | bm[3]) != 0u;
                    int j = 0;
                    unsigned m = bm[0];
                    if (!m) { j = 1; m = bm[1]; }
                    if (!m) { j = 2; m = bm[2]; }
                    if (!m) { j = 3; m = bm[3]; }
                    const int bit = has ? __ffs((int)m) - 1 : 0;
                    const unsigned cl = m & (m - 1u);
                    if (j == 0) bm[0] = cl; else if (j == 1) bm[1] = cl; else if (j == 2) bm[2] = cl; else bm[3] = cl;
                    const int i = 15 - (bit >> 1);                           // register index: the last one shifted in is bit 0
                    const int row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                    const int ent = ((wv + 4 * (j >> 1)) * kWave) + (j & 1) * 32 + row;
                    const int code = (T * kHypTile + (lane & 31)) | (ent << 16);
                    const unsigned long long mk = __builtin_amdgcn_ballot_w64(has);
                    if (qn + __popcll(mk) > kBandQ) {
                        band_flush(qn, bq, U, nvalid, thin, inst, fg, kernel_params());
                        qn = 0;
                    }
                    if (has) bq[qn + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0))] = code;
                    qn += __popcll(mk);
                }
            }
            if (qn) band_flush(qn, bq, U, nvalid, thin, inst, fg, kernel_params());
        }
        __syncthreads();
        // one integer atomic per (unit, hypothesis) with any count: order-independent result
        for (int i = threadIdx.x; i < (T1 - T0) * kHypTile; i += blockDim.x) {
            const int h = T0 * kHypTile + i, cv = s_cnt[i];
            if (cv && h < p.hn) atomicAdd(&p.counts[(size_t)inst * p.hnp + h], cv);
        }
        __syncthreads();
    }
}

// ---- k_vote_final ------------------------------------------------------------------
// b_inv (RV/ransac_voting_gpu.py:503-516): inverse when regular, pseudo-inverse when singular.  torch.solve raises only
// on an exactly singular LU; here the pseudo-inverse also takes over for det <= 1e-12 tr^2 (conditioning beyond fp64's
// reach for the normal equations of f32 votes) — documented in include/fpc.h.
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

constexpr int kFinWaves = 4;         // 256-thread workgroups

// Winner (largest count, lowest index: torch.max, RV/ransac_voting_gpu.py:567), its inliers voted again (:583-589) with
// the reference's arithmetic, the fp64 normal equations and the 2x2 solve (:592-599).  Task = one work unit; the unit of
// an instance whose arrival ticket comes last combines the instance's records (Guideline 16, counter form: records stored
// write-through (sc1), the storing wave drained, one agent-scope add per workgroup; the last arriver reads them back with
// sc1 loads, in ordinal order: bit-reproducible).
__global__ __launch_bounds__(256) void k_vote_final(const VoteParams p) {
    __shared__ int s_red[2 * kFinWaves];
    __shared__ int s_last;
    __shared__ double s_part[kFinWaves][kRec];
    const int n_act = active_instances(p.n, p.n_dev);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const int hn = p.hn;
    // instances that do not vote (fewer than min_num pixels) have no unit: zeros (RV/ransac_voting_gpu.py:536-539)
    for (int inst = blockIdx.x * blockDim.x + threadIdx.x; inst < n_act; inst += gridDim.x * blockDim.x)
        if (p.plan[(size_t)inst * kPlanI + 6] == 0) {
            p.out_xy[inst * 2] = 0.0f; p.out_xy[inst * 2 + 1] = 0.0f;
            if (p.out_tn) p.out_tn[inst] = p.plan[(size_t)inst * kPlanI + 1];
            if (p.out_win_idx) p.out_win_idx[inst] = -1;
            if (p.out_win_count) p.out_win_count[inst] = 0;
            if (p.out_inl) p.out_inl[inst] = 0;
            if (p.out_refine)
                for (int i = 0; i < 8; ++i) p.out_refine[(size_t)inst * 8 + i] = 0.0;
        }
    const int nu = *p.ctrl;
    for (int t = blockIdx.x; t < nu; t += gridDim.x) {
        const int4 ub = p.units[t];
        const int inst = ub.x, c = ub.y, k = ub.z & 7, ord = ub.z >> 3, cnt = ub.w;
        const int nvalid = min(kUnitEntries, cnt - k * kUnitEntries);
        const int32_t* pl = p.plan + (size_t)inst * kPlanI;
        const int fg = pl[0], tn = pl[1], nrec = pl[6];
        const bool thin = pl[2] != 0;
        // winner: every task of the instance finds the same one
        int wc = -1, wi = 0x7fffffff;
        for (int h = threadIdx.x; h < hn; h += blockDim.x) {
            const int cv = p.counts[(size_t)inst * p.hnp + h];
            if (cv > wc) { wc = cv; wi = h; }                            // ascending h: first maximum kept
        }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            const int oc = __shfl_xor(wc, o, kWave), oi = __shfl_xor(wi, o, kWave);
            if (oc > wc || (oc == wc && oi < wi)) { wc = oc; wi = oi; }
        }
        __syncthreads();                                               // LDS of the previous task is free
        if (lane == 0) { s_red[wv] = wc; s_red[kFinWaves + wv] = wi; }
        __syncthreads();
        wc = s_red[0]; wi = s_red[kFinWaves];
#pragma unroll
        for (int i = 1; i < kFinWaves; ++i) {
            const int oc = s_red[i], oi = s_red[kFinWaves + i];
            if (oc > wc || (oc == wc && oi < wi)) { wc = oc; wi = oi; }
        }
        // no hypothesis with an inlier: all_win_pts stays (0,0) (:571-574) and the refinement votes for (0,0)
        const float* hp = p.hyp + (size_t)inst * hn * 2;
        float wx = 0.0f, wy = 0.0f;
        if (wc > 0) { wx = hp[2 * wi]; wy = hp[2 * wi + 1]; } else { wi = -1; wc = 0; }

        const float4* U = p.list + (size_t)inst * p.ls + (size_t)c * kChunkPx + (size_t)k * kUnitEntries;
        double v[kRec] = {0, 0, 0, 0, 0, 0};                            // inliers, a00, a01, a11, b0, b1
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int e = (wv + 4 * gi) * kWave + lane;
            bool valid = e < nvalid;
            const float4 q = valid ? U[e] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid && thin) valid = pixel_kept(q.x, q.y, p.W, p.HW, inst, fg, p.max_num, p.seed, p.keep);
            if (valid && pair_is_inlier(q.x, q.y, q.z, q.w, sqrtf(q.z * q.z + q.w * q.w), wx, wy, p.thresh)) {
                const double nx = (double)q.w, ny = -(double)q.z;      // normal = (dy, -dx) :584-586
                const double bb = nx * (double)q.x + ny * (double)q.y;
                v[0] += 1.0; v[1] += nx * nx; v[2] += nx * ny; v[3] += ny * ny; v[4] += nx * bb; v[5] += ny * bb;
            }
        }
#pragma unroll
        for (int a = 0; a < kRec; ++a) {
            const double r = wave_reduce_add(v[a]);
            if (lane == 0) s_part[wv][a] = r;
        }
        __syncthreads();
        if (threadIdx.x < kRec) {
            const double r = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
            store_wt64(p.partial + ((size_t)inst * p.nux + ord) * kRec + threadIdx.x, __builtin_bit_cast(unsigned long long, r));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the storing wave drains its sc1 stores
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tk = __hip_atomic_fetch_add(p.ctrl + 4 + p.n + inst, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (tk == nrec - 1);
        }
        __syncthreads();
        if (!s_last) continue;                                         // uniform

        // last arriver of the instance: the records in ordinal order (independent sc1 loads, four in flight per lane)
        if (wv == 0) {
            double tot[kRec] = {0, 0, 0, 0, 0, 0};
            // lane = (record slot r8 = lane / 8, value a = lane % 8): eight records per sweep, then a fixed-order lane tree
            const int a = lane & 7, r8 = lane >> 3;
            double acc = 0.0;
            for (int b = r8; b < nrec; b += 32) {
                unsigned long long x[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    x[i] = (a < kRec && b + 8 * i < nrec) ? load_wt64(p.partial + ((size_t)inst * p.nux + b + 8 * i) * kRec + a) : 0ull;
#pragma unroll
                for (int i = 0; i < 4; ++i) acc += __builtin_bit_cast(double, x[i]);
            }
            acc += __shfl_xor(acc, 8, kWave); acc += __shfl_xor(acc, 16, kWave); acc += __shfl_xor(acc, 32, kWave);
#pragma unroll
            for (int i = 0; i < kRec; ++i) tot[i] = __shfl(acc, i, kWave);
            if (lane == 0) {
                double x0, x1;
                solve2_sym(tot[1], tot[2], tot[3], tot[4], tot[5], x0, x1);
                p.out_xy[inst * 2] = (float)x0;
                p.out_xy[inst * 2 + 1] = (float)x1;
                if (p.out_tn) p.out_tn[inst] = tn;
                if (p.out_win_idx) p.out_win_idx[inst] = wi;
                if (p.out_win_count) p.out_win_count[inst] = wc;
                if (p.out_inl) p.out_inl[inst] = (int)tot[0];
                if (p.out_refine) {      // what the refinement's backward needs (fpc_vote_refine_backward)
                    double* r = p.out_refine + (size_t)inst * 8;
                    r[0] = (double)wx; r[1] = (double)wy; r[2] = tot[1]; r[3] = tot[2]; r[4] = tot[3]; r[5] = tot[4];
                    r[6] = tot[5]; r[7] = tot[0];
                }
            }
        }
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
    hipLaunchKernelGGL(k_b1_vote, dim3(cdiv(vn * tn, 256), std::min(hn, 65535)), dim3(256), 0, (hipStream_t)stream, direct,
                       coords, hyp, inliers, tn, vn, hn, inlier_thresh);
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
                                    int32_t* out_counts, double* out_refine, void* ws, size_t ws_bytes,
                                    fpc_stream_t stream) {
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
    VoteParams& p = w.p;
    p.mask = mask; p.vertex = vertex; p.vs_n = vs_n; p.vs_h = vs_h; p.vs_w = vs_w; p.vs_c = vs_c;
    p.n = n; p.n_dev = n_dev; p.W = W; p.HW = HW; p.hn = hn;
    p.idxs = idxs; p.keep = keep; p.seed = seed; p.thresh = inlier_thresh; p.min_num = min_num; p.max_num = max_num;
    p.out_xy = out_xy; p.out_tn = out_tn; p.out_win_idx = out_win_idx; p.out_win_count = out_win_count;
    p.out_inl = out_inl_count; p.out_refine = out_refine;
    p.want_tn = out_tn ? 1 : 0;

    // the cones need th' = th - 1e-6 > 0; otherwise every pair takes the reference's arithmetic
    const bool fast = inlier_thresh > 2e-6f && inlier_thresh < 3.0e38f;
    float kappa1 = 0.0f, kappa2 = 0.0f;
    if (fast) {
        const double th1 = (double)inlier_thresh - 1e-6, th2 = (double)inlier_thresh + 1e-6;
        const double k1 = 1.0 - th1 * th1, k2 = 1.0 - th2 * th2;
        kappa1 = (float)((k1 > 0.0 ? sqrt(k1) : 0.0) / th1) * (1.0f + 1e-6f);                       // wider
        kappa2 = (th2 < 1.0 && k2 > 0.0) ? (float)(sqrt(k2) / th2) * (1.0f - 1e-6f) : 0.0f;        // narrower (0: no "sure")
    }
    p.all_wild = fast ? 0 : 1;
    p.kappa2 = kappa2;
    p.dkappa = (kappa1 - kappa2) * (1.0f + 1e-6f);
    // rounding allowance of the margin per unit of magnitude M = |gx - ox| + |gy - oy| + radius (k_vote_count's header)
    p.efac = 3.2e-6f * (1.0f + kappa1);
    p.lds_table = p.nch + 1 <= 2048 ? 1 : 0;              // the three chunk tables of an instance in LDS (<= 24 KB)
    const size_t table_lds = p.lds_table ? 3 * (size_t)(p.nch + 1) * sizeof(int) : 0;

    // 0. arrival counters and the unit count
    {
        hipError_t e = hipMemsetAsync(p.ctrl, 0, w.ctrl_bytes, s);
        if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    }
    // 1. mask planes -> per-chunk compacted pixel lists (the only pass over the masks and the vote planes); the plan of an
    //    instance in the tail of its last chunk
    const bool vec4 = (HW % 4 == 0) && (((uintptr_t)mask & 15) == 0);
    const bool vg4 = vec4 && W % 4 == 0 && vs_w == 1 && vs_h % 4 == 0 && vs_n % 4 == 0 && vs_c % 4 == 0 &&
                     (((uintptr_t)vertex & 15) == 0);
    const int scan_grid = (int)std::min<long long>((long long)n * p.nch, 8192);
#define FPC_LAUNCH_SCAN(A, B) hipLaunchKernelGGL((k_vote_scan<A, B>), dim3(scan_grid), dim3(256), table_lds, s, p)
    if (vg4) FPC_LAUNCH_SCAN(true, true); else if (vec4) FPC_LAUNCH_SCAN(true, false); else FPC_LAUNCH_SCAN(false, false);
#undef FPC_LAUNCH_SCAN

    // 2. exact inlier counts of every hypothesis.  One resident round of workgroups (four per CU); the kernel reads how
    //    many units exist and cuts the hypothesis tiles into slices so that the tasks fill that round evenly.
    const long long cap_tasks = (long long)n * p.nux * p.ntiles;
    const int resident = 256 * 4;
    const int count_grid = (int)std::min<long long>(cap_tasks, resident);
    p.task_target = count_grid;
    const size_t count_lds = (size_t)std::min(p.ntiles, kMaxSliceTiles) * kHypTile * sizeof(int);
    hipLaunchKernelGGL((k_vote_count<4>), dim3(count_grid), dim3(256), count_lds, s, p);

    // 3. winner, its inliers, refinement: one task per work unit
    const int fin_grid = (int)std::min<long long>(std::max<long long>((long long)n * p.nux, 1), 2048);
    hipLaunchKernelGGL(k_vote_final, dim3(fin_grid), dim3(256), 0, s, p);

    // diagnostics (never on the product path): copies of the hypotheses and of the count rows
    if (out_hyp) {
        hipError_t e = hipMemcpyAsync(out_hyp, p.hyp, sizeof(float) * (size_t)n * hn * 2, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    }
    if (out_counts) {
        hipError_t e = hipMemcpy2DAsync(out_counts, sizeof(int32_t) * (size_t)hn, p.counts, sizeof(int32_t) * (size_t)p.hnp,
                                        sizeof(int32_t) * (size_t)hn, (size_t)n, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    }
    return check_launch();
}
