// vote.hpp — what the kernels of the fused hough vote share (csrc/ransac.hip: scan, plan, final, host; csrc/vote_count.hip:
// the MFMA count kernel, a translation unit of its own because only it is built with -amdgpu-mfma-vgpr-form).
#pragma once
#include "common.hpp"

namespace fpc {

constexpr int kChunkPx = 4096;       // pixels per k_vote_scan task; a chunk owns list slots [c * 4096, c * 4096 + its count)
constexpr int kChunkWords = 64;      // 64-pixel words per chunk
constexpr int kUnitEntries = 512;    // foreground ranks per count unit (4 waves x 2 groups of 64): [512 u, 512 u + 512) of an instance
constexpr int kHypTile = 32;         // hypotheses per MFMA tile
constexpr int kMaxSliceTiles = 64;   // hypothesis tiles per k_vote_count task at most (LDS count rows)
constexpr int kPlanI = 8;            // i32 per instance: fg, tn, thin, origin x, origin y, radius, runs, votes
constexpr int kMaxHn = 65536;
constexpr int kRec = 6;              // doubles per refinement record: inliers, a00, a01, a11, b0, b1
constexpr int kBandQ = 128;          // queued undecided-pair records per wave and task (one step adds at most 64)
constexpr float kNeverS = 1.0e30f;   // |s| of an entry that never votes
constexpr int kPInfoI = 8;           // i32 per instance of the progressive count: unit base, units, alive hypotheses, their tiles,
                                     // leader, its full count L, valid entries not yet counted, -
constexpr int kMaxPasses = 4;        // passes of the progressive count at most
constexpr int kProgMaxInst = 1024;   // instances whose per-pass tables fit k_vote_count_prog's LDS
constexpr int kProgMaxUnits = 16384; // count units per instance whose "not yet counted" bits fit k_vote_lead's LDS (8.4 M pixels)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long __attribute__((address_space(1))) gu64;

// everything the four kernels share (passed by value)
struct VoteParams {
    // caller
    const float* mask; const float* vertex; int64_t vs_n, vs_h, vs_w, vs_c;
    int n; const int32_t* n_dev; int W, HW, hn;
    const int32_t* idxs; const uint8_t* keep; uint64_t seed; float thresh; int min_num, max_num;
    float* out_xy; int32_t* out_tn; int32_t* out_win_idx; int32_t* out_win_count; int32_t* out_inl; double* out_refine;
    // optional (fpc_ransac_voting_v3_pose): the RT assembly of each instance appended to its voted centre by k_vote_final
    const float* pose_q; const float* pose_z; const float* pose_kinv; float* pose_R; float* pose_T; float* pose_RT;
    // derived
    int nch, ntiles, hnp, nux, nrx, lds_table, want_tn, all_wild, task_target;
    int run_entries;                  // foreground ranks per refinement run (k_vote_final task)
    int npass;                        // progressive count: passes (0 / 1 = the exhaustive k_vote_count)
    int pcum[kMaxPasses + 1];         // ... pass p takes the unit positions [pass_begin(p), pass_begin(p + 1)) of an instance, in 16ths
    size_t ls;                        // list slots per instance = nch * kChunkPx
    float kappa2, dkappa, efac;
    // workspace
    int32_t* ctrl;        // [0] count units, [1] refinement runs (zeroed by k_vote_scan, appended to by k_vote_plan)
    int32_t* tickets;     // [n]            k_vote_final arrivals; zeroed by k_vote_plan
    int32_t* plan;        // [n, kPlanI]
    int32_t* chunk_fg;    // [n, nch]       foreground count per chunk
    int32_t* chunk_box;   // [n, nch, 4]    x min / max, y min / max of the chunk's foreground pixels
    int32_t* chunk_pre;   // [n, nch + 1]   exclusive prefix of the counts: rank r of an instance sits in chunk c with
                          //                chunk_pre[c] <= r < chunk_pre[c + 1], at list slot c * 4096 + r - chunk_pre[c]
    int32_t* kept_pre;    // [n, nch + 1]   prefix over the KEPT entries (thinned instance with injected idxs / out_tn only)
    uint32_t* kept_wpre;  // [n, nch * 64]  kept entries before each 64-entry group inside its chunk     (same case)
    uint64_t* kept_bits;  // [n, nch * 64]  keep decisions of each 64-entry group                        (same case)
    float* hyp;           // [n, hn, 2]     hypothesis points as the reference's [hn,1,2] tensor
    u32x4* hypB;          // [n, ntiles, 64] their MFMA B fragments (lane = column + 32 * k-half)
    int32_t* counts;      // [n, hnp]       exact inlier count of every hypothesis; zeroed by k_vote_plan
    double* partial;      // [n, nrx, kRec] k_vote_final per-run records
    float4* list;         // [n, ls]        {x, y, dx, dy} of the foreground pixels, compacted per chunk (k_vote_scan)
    int4* units;          // [n * nux, 2]   {instance | thin << 16 | (entries - 1) << 17, block u, chunk c of rank 512 u, ox | oy << 16},
                          //                {list slot of rank 512 u, ranks of the unit inside chunk c, fg, -}
    int4* runs;           // [n * nrx]      {instance, run r (= its record ordinal), chunk of rank r * run_entries, fg | thin << 31}
    int32_t* pinfo;       // [n, kPInfoI]   progressive count: see kPInfoI (written by k_vote_plan, updated by k_vote_lead)
    int32_t* hmap;        // [2, n, hnp]    slot -> hypothesis of the alive set (ping-pong between passes; pass 0: identity, not stored)
    u32x4* hypC;          // [n, ntiles, 64] B fragments of the alive hypotheses, compacted in slot order (k_vote_lead)
#ifdef FPC_STAMP_VOTE
    unsigned long long* dbg;      // [kMaxPasses, 1024, 4] per workgroup of k_vote_count_prog: start, end, segments, XCC id (diagnostic build)
#endif
    unsigned long long* stamps;   // [4, 32]  s_memrealtime (100 MHz) at the phases of workgroup 0 of each kernel: written only by a
                                  //          diagnostic build (-DFPC_STAMP_VOTE, tools_dev/vote_stamps.py); never read by a kernel
};

#ifdef FPC_STAMP_VOTE
#define FPC_STAMP(kernel, slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) p.stamps[(kernel) * 32 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FPC_STAMP(kernel, slot) do { } while (0)
#endif

struct Ws {
    VoteParams p;
    size_t total;
};

// 8192: at B = 32 the 370 runs are one resident round of 512-thread workgroups (4096: two rounds, 38 us; whole instances:
// the 35 000-pixel one alone takes 25 us; measured); a handful of instances are cut finer so that more CUs share them
#ifndef FPC_RUN_ENTRIES
#define FPC_RUN_ENTRIES 8192
#endif
inline int run_entries_for(int n) { return n <= 16 ? 2048 : FPC_RUN_ENTRIES; }

inline Ws carve(void* base, int n, int H, int W, int hn) {
    Ws w;
    VoteParams& p = w.p;
    const size_t HW = (size_t)H * W;
    p.nch = cdiv((int)HW, kChunkPx);
    p.ntiles = cdiv(hn, kHypTile);
    p.hnp = p.ntiles * kHypTile;
    p.nux = cdiv((int)HW, kUnitEntries);
    p.run_entries = run_entries_for(n);
    p.nrx = cdiv((int)HW, p.run_entries);
    p.ls = (size_t)p.nch * kChunkPx;
    char* b = (char*)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* q = b + off; off = align_up(off + bytes, 256); return q; };
    p.ctrl = (int32_t*)take(sizeof(int32_t) * 4);
    p.tickets = (int32_t*)take(sizeof(int32_t) * (size_t)n);
    p.plan = (int32_t*)take(sizeof(int32_t) * (size_t)n * kPlanI);
    p.chunk_fg = (int32_t*)take(sizeof(int32_t) * (size_t)n * p.nch);
    p.chunk_box = (int32_t*)take(sizeof(int32_t) * (size_t)n * p.nch * 4);
    p.chunk_pre = (int32_t*)take(sizeof(int32_t) * (size_t)n * (p.nch + 1));
    p.kept_pre = (int32_t*)take(sizeof(int32_t) * (size_t)n * (p.nch + 1));
    p.kept_wpre = (uint32_t*)take(sizeof(uint32_t) * (size_t)n * p.nch * kChunkWords);
    p.kept_bits = (uint64_t*)take(sizeof(uint64_t) * (size_t)n * p.nch * kChunkWords);
    p.hyp = (float*)take(sizeof(float) * (size_t)n * hn * 2);
    p.hypB = (u32x4*)take(sizeof(u32x4) * (size_t)n * p.ntiles * kWave);
    p.counts = (int32_t*)take(sizeof(int32_t) * (size_t)n * p.hnp);
    p.partial = (double*)take(sizeof(double) * (size_t)n * p.nrx * kRec);
    p.list = (float4*)take(sizeof(float4) * (size_t)n * p.ls);
    p.units = (int4*)take(sizeof(int4) * 2 * (size_t)n * p.nux);
    p.runs = (int4*)take(sizeof(int4) * (size_t)n * p.nrx);
    p.pinfo = (int32_t*)take(sizeof(int32_t) * (size_t)n * kPInfoI);
    p.hmap = (int32_t*)take(sizeof(int32_t) * 2 * (size_t)n * p.hnp);
    p.hypC = (u32x4*)take(sizeof(u32x4) * (size_t)n * p.ntiles * kWave);
#ifdef FPC_STAMP_VOTE
    p.dbg = (unsigned long long*)take(sizeof(unsigned long long) * kMaxPasses * 1024 * 4);
#endif
    p.stamps = (unsigned long long*)take(sizeof(unsigned long long) * 4 * 32);
    w.total = off;
    return w;
}

__device__ __forceinline__ int active_instances(int n, const int32_t* __restrict__ n_dev) {
    if (!n_dev) return n;
    int m = *n_dev;
    return m < n ? (m < 0 ? 0 : m) : n;
}

// write-through (sc1) accessors of the records that cross workgroups inside k_vote_final
__device__ __forceinline__ unsigned long long load_wt64(const void* p) {
    return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_wt64(void* p, unsigned long long v) {
    __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// list slot of foreground rank r, walking the instance's chunk prefix forward from chunk c (cpre[c] <= r): a count unit or
// a refinement run starts in a known chunk and its ranks sit in that chunk or the next few
__device__ __forceinline__ int rank_slot_from(const int32_t* __restrict__ cpre, int& c, int r) {
    while (cpre[c + 1] <= r) ++c;
    return c * kChunkPx + (r - cpre[c]);
}

// ---- progressive count: which units a pass takes ---------------------------------------------------------------------
// An instance's unit records are stored in a PERMUTED order (position j holds unit (j * stride) mod nunits, stride ~ 0.618
// nunits and coprime to it: every block of consecutive positions spreads over the whole instance); pass p counts the
// positions [pass_begin(p), pass_begin(p + 1)).
__host__ __device__ __forceinline__ int unit_stride(int nunits) {
    if (nunits < 3) return 1;
    int s = (int)((long long)nunits * 618 / 1000);
    if (s < 1) s = 1;
    for (;; ++s) {                                   // nunits - 1 is coprime to nunits: terminates below nunits
        int a = s, b = nunits;
        while (b) { const int t = a % b; a = b; b = t; }
        if (a == 1) return s;
    }
}
__host__ __device__ __forceinline__ int pass_begin(const int* pcum, int npass, int pass, int nunits) {
    if (pass <= 0) return 0;
    if (pass >= npass) return nunits;
    const long long b = ((long long)nunits * pcum[pass] + 15) / 16;
    return b < nunits ? (int)b : nunits;
}

// Exclusive scan of f(i), i in [0, cnt), into out[0..cnt] (out[cnt] = total) by the whole workgroup; out may be LDS or
// global, and f(i) may read out[i] (every thread reads its element before any thread of the tile writes).
template <typename F>
__device__ __forceinline__ int block_scan(F f, int32_t* out, int cnt, int* s_w /* >= blockDim.x / 64 + 1 ints */) {
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
    __syncthreads();                                   // LDS, or global through this CU's own L1 / L2: visible to the block
    return carry;
}

// Is pixel (x, y) of a THINNED instance kept (RV/ransac_voting_gpu.py:541-545)?  KEEP: the caller injected the selection.
// A template parameter, not a test of the pointer: hipcc 7.2 (clang 22) kept the wave-uniform `keep != nullptr` of the plan
// kernel in a VGPR under SGPR pressure, re-expanded it to a lane mask with v_cmp under the partial EXEC of one divergent
// loop and reused that mask under the wider EXEC of the next loop: lanes that had been inactive took the `keep[...]` side
// with a null pointer (memory fault at inst * HW + pixel).  No uniform runtime condition sits inside a divergent loop here.
template <bool KEEP>
__device__ __forceinline__ bool pixel_kept(float x, float y, int W, int HW, int inst, int fg, int max_num, uint64_t seed,
                                           const uint8_t* __restrict__ keep) {
    const int p = (int)y * W + (int)x;
    if constexpr (KEEP) return keep[(size_t)inst * HW + p] != 0;
    else return fpc_rand_keep(seed, (uint32_t)inst, (uint32_t)p, (uint32_t)fg, (uint32_t)max_num) != 0;
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


void launch_vote_count(const VoteParams& p, int grid, size_t lds_bytes, hipStream_t s);   // picks the KEEP variant from p.keep
void launch_vote_count_prog(const VoteParams& p, int pass, int grid, hipStream_t s);         // one pass of the progressive count

}  // namespace fpc
