// ransac.hip — PVNet-derived RANSAC hough voting for gfx950.
//
//  * fpc_generate_hypothesis / fpc_voting_for_hypothesis: B1-compatible kernels
//    (reference: RV/src/ransac_voting_kernel.cu:11-167).
//  * fpc_ransac_voting_v3: the whole of ransac_voting_layer_v3 (RV/ransac_voting_gpu.py:518-607) for a batch of
//    instances in FOUR stateless launches, without a host round trip and without the hn x tn inlier matrix:
//
//      k_vote_scan    task = (instance, chunk of 4096 pixels): the only pass over the caller's planes.  Mask -> bit words,
//                     in-chunk prefix, chunk count and bounding box; the chunk's foreground pixels are compacted in raster
//                     order into the chunk's own slots of ONE float4 list {x, y, dx, dy} (vote gathered through the caller's
//                     strides).                                                  (HBM: n x 12 H W bytes, read once)
//      k_vote_plan    per instance (256 threads; up to four workgroups split the hypothesis tiles): chunk prefix (rank -> slot),
//                     the integer origin / radius the filter's coordinates are measured from, the work units of the count
//                     (512 consecutive foreground ranks) and of the refinement (runs of 8192 ranks), the hn hypotheses (:552,559; pair sampling,
//                     two-line intersection exactly as .cu:28-45) — each also as a bf16 MFMA B-fragment of the filter —
//                     and a zeroed count row.
//      k_vote_count   one resident round of workgroups over units x hypothesis slices.  EXACT inlier counts.  The two affine
//                     forms of the filter run on the matrix cores in split precision (below); per (entry, hypothesis) the
//                     VALU does one subtraction and one v_alignbit that shifts TWO bits of the margin into a per-lane row;
//                     pairs the filter cannot decide (1 in 1000) are queued and take the reference's own arithmetic
//                     (.cu:106-125), 64 at a time.  Integer atomics per (unit, hypothesis).
//      k_vote_final   task = run of chunks: torch.max's winner (:567, first maximal index), its inliers voted again with the
//                     reference's arithmetic, fp64 normal-equation records; the run of an instance that arrives last sums
//                     them in run order and solves the 2x2 system in closed form (b_inv, :503-516, :583-599).
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
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>

#include "vote.hpp"

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
        // div_ieee (common.hpp): the two quotients must not be interleaved by the scheduler
        y = div_ieee(nx1 * (nx0 * cx0 + ny0 * cy0) - nx0 * (nx1 * cx1 + ny1 * cy1), det_y);
        x = div_ieee(ny1 * (nx0 * cx0 + ny0 * cy0) - ny0 * (nx1 * cx1 + ny1 * cy1), det_x);
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

// ---- k_vote_scan -------------------------------------------------------------------
// grid-stride over (instance, chunk) tasks; 256 threads; a chunk = 4096 pixels = 4 float4 of the mask per lane.
// Writes the chunk's count and bounding box, and compacts its foreground pixels (vote gathered from the caller's
// strided planes) into the chunk's own slots of the list.
// A parameter block of its own: with the whole VoteParams the compiler held 98 VGPRs (5 waves per SIMD) for this
// latency-bound streaming kernel, with these sixteen values 52 (8 waves).
struct ScanParams {
    const float* mask; const uint64_t* bits; const float* vertex; int64_t vs_n, vs_h, vs_w, vs_c;
    int n; const int32_t* n_dev; int W, HW, nch;
    size_t ls;
    int32_t* ctrl; int32_t* chunk_fg; int32_t* chunk_box; float4* list;
    unsigned long long* stamps;
};

// BITS: the caller supplies the foreground as bit words (p.bits [n][nch * 64] u64, bit j of word w = pixel 64 w + j, zero
// past H W): the f32 mask plane — two thirds of this kernel's bytes at a 13 % foreground — is not read at all.
template <bool VEC4, bool VGATHER4, bool BITS = false>
__global__ __launch_bounds__(256) void k_vote_scan(const ScanParams p) {
    __shared__ __attribute__((aligned(16))) uint8_t s_nib[kChunkPx / 4];
    __shared__ uint64_t s_word[kChunkWords];
    __shared__ int s_wpre[kChunkWords];
    __shared__ int s_box[4];
    __shared__ int s_tot;
    FPC_STAMP(0, 0);
    if (blockIdx.x == 0 && threadIdx.x < 2) p.ctrl[threadIdx.x] = 0;       // k_vote_plan appends this call's units and runs
    const int W = p.W, HW = p.HW, nch = p.nch;
    const int total = active_instances(p.n, p.n_dev) * nch;
    // the mask of the NEXT task is requested before the current one is processed (VEC4): a task is two dependent memory
    // round trips (mask, then the votes under it) and the grid is a few resident rounds deep
    float4 cur[4], nxt[4];
    auto load_mask = [&](int t, float4 (&mv)[4]) {
        const int inst = t / nch, c = t - inst * nch;
        const float* m = p.mask + (size_t)inst * HW;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int px = c * kChunkPx + (k * 256 + threadIdx.x) * 4;
            mv[k] = px < HW ? *reinterpret_cast<const float4*>(m + px) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // (Tasks handed out by tickets instead of this static round-robin - 16 sharded words, a memset - measured 70 + 5 us
    // against 68 us at B = 32: the kernel moves 330 MB in 65 us, it is at the memory system's rate, not unbalanced.)
    uint64_t curw = 0, nxtw = 0;
    auto load_bits = [&](int t) -> uint64_t {
        const int inst = t / nch, c = t - inst * nch;
        return threadIdx.x < kChunkWords ? p.bits[((size_t)inst * nch + c) * kChunkWords + threadIdx.x] : 0ull;
    };
    if (BITS && (int)blockIdx.x < total) curw = load_bits(blockIdx.x);
    if (!BITS && VEC4 && (int)blockIdx.x < total) load_mask(blockIdx.x, cur);
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int inst = t / nch, c = t - inst * nch;
        const float* m = p.mask + (size_t)inst * HW;
        if (BITS && t + (int)gridDim.x < total) nxtw = load_bits(t + gridDim.x);
        if (!BITS && VEC4 && t + (int)gridDim.x < total) load_mask(t + gridDim.x, nxt);
        unsigned nb[4];
#pragma unroll
        for (int k = 0; k < 4 && !BITS; ++k) {
            const int fi = k * 256 + threadIdx.x;           // float4 index inside the chunk
            const int px = c * kChunkPx + fi * 4;
            nb[k] = 0;
            if (VEC4) {                                     // HW % 4 == 0 and a 16-byte aligned plane
                const float4 v = cur[k];
                nb[k] = (v.x != 0.0f ? 1u : 0u) | (v.y != 0.0f ? 2u : 0u) | (v.z != 0.0f ? 4u : 0u) | (v.w != 0.0f ? 8u : 0u);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (px + q < HW && m[px + q] != 0.0f) nb[k] |= 1u << q;
            }
        }
        if (!BITS) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s_nib[k * 256 + threadIdx.x] = (uint8_t)nb[k];
            __syncthreads();
        }
        if (threadIdx.x < kChunkWords) {
            const uint64_t word = BITS ? curw : pack_nibbles(*reinterpret_cast<const uint4*>(s_nib + 16 * threadIdx.x));
            int tot;
            const int ex = wave_excl_scan(__popcll(word), tot);
            s_word[threadIdx.x] = word;
            s_wpre[threadIdx.x] = ex;
            if (threadIdx.x == 0) {
                p.chunk_fg[(size_t)inst * nch + c] = tot;
                s_box[0] = 0x7fffffff; s_box[1] = -1; s_box[2] = 0x7fffffff; s_box[3] = -1;
                s_tot = tot;
            }
        }
        __syncthreads();
        // a chunk without a foreground pixel (most of an instance's plane: 85 % of the tasks of the 32-frame batch) is done
        // here: an empty box and the next task — no compaction, no box reduction, one barrier instead of three
        if (s_tot == 0) {                                          // uniform
            if (threadIdx.x < 4) p.chunk_box[((size_t)inst * nch + c) * 4 + threadIdx.x] = (threadIdx.x & 1) ? -1 : 0x7fffffff;
            __syncthreads();                                       // s_tot / s_word are rewritten by the next task
            if (BITS) curw = nxtw;
            if (!BITS && VEC4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
            }
            continue;
        }
        if (BITS) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int fi = k * 256 + threadIdx.x;
                nb[k] = (unsigned)(s_word[fi >> 4] >> ((fi & 15) * 4)) & 15u;
            }
        }
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
                L[pos] = e;
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
        if (threadIdx.x < 4) p.chunk_box[((size_t)inst * nch + c) * 4 + threadIdx.x] = s_box[threadIdx.x];
        __syncthreads();
        FPC_STAMP(0, 1);
        if (BITS) curw = nxtw;
        if (!BITS && VEC4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
        }
    }
}

// ---- k_vote_plan -------------------------------------------------------------------
// gridDim.y 256-thread workgroups per instance: each repeats the (cheap) prefix and bounding box and takes a share of the
// hypotheses, part 0 writes the instance's record, units and runs.  dynamic LDS: [2][nch + 1] ints when p.lds_table.
// INJ: the caller injected pair indices or a keep selection, or wants the out_tn diagnostic (tests and goldens): only that
// variant carries the kept-pixel tables, and it runs as ONE part.  KEEP: see pixel_kept.
template <bool INJ, bool KEEP>
__global__ __launch_bounds__(256) void k_vote_plan(const VoteParams p) {
    extern __shared__ __attribute__((aligned(16))) int s_tab[];
    __shared__ int s_w[20];
    __shared__ int s_misc[8];
    const int nch = p.nch, W = p.W, HW = p.HW, hn = p.hn;
    const int n_act = active_instances(p.n, p.n_dev);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave, nw = blockDim.x / kWave;
    const int part = blockIdx.y, parts = gridDim.y;
    const int tiles_pp = (p.ntiles + parts - 1) / parts;                    // hypothesis tiles per part
    const int h_lo = min(p.hnp, part * tiles_pp * kHypTile), h_hi = min(p.hnp, (part + 1) * tiles_pp * kHypTile);
    for (int inst = blockIdx.x; inst < n_act; inst += gridDim.x) {
        const int32_t* cfg = p.chunk_fg + (size_t)inst * nch;
        int32_t* gpre = p.chunk_pre + (size_t)inst * (nch + 1);                    // k_vote_count / k_vote_final read it
        int32_t* cpre = p.lds_table ? s_tab : gpre;
        int32_t* kpre = p.lds_table ? s_tab + (nch + 1) : p.kept_pre + (size_t)inst * (nch + 1);
        const float4* E = p.list + (size_t)inst * p.ls;

        FPC_STAMP(1, 0);
        for (int h = h_lo + threadIdx.x; h < h_hi; h += blockDim.x) p.counts[(size_t)inst * p.hnp + h] = 0;
        if (part == 0 && threadIdx.x == 0) p.tickets[inst] = 0;
        if (threadIdx.x < 4) s_misc[threadIdx.x] = (threadIdx.x & 1) ? -1 : 0x7fffffff;
        // bounding box of the instance -> the origin the filter's coordinates are measured from and the radius
        // max |x - ox| + |y - oy| of its pixels (both only scale the rounding allowance: any values are sound)
        int b0 = 0x7fffffff, b1 = -1, b2 = 0x7fffffff, b3 = -1;
        for (int c = threadIdx.x; c < nch; c += blockDim.x) {
            const int4 bx = *reinterpret_cast<const int4*>(p.chunk_box + ((size_t)inst * nch + c) * 4);
            b0 = min(b0, bx.x); b1 = max(b1, bx.y); b2 = min(b2, bx.z); b3 = max(b3, bx.w);
        }
        const int fg = block_scan([&](int c) { return cfg[c]; }, cpre, nch, s_w);
        FPC_STAMP(1, 1);
        if (p.lds_table && part == 0)
            for (int c = threadIdx.x; c <= nch; c += blockDim.x) gpre[c] = cpre[c];
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            b0 = min(b0, __shfl_xor(b0, o, kWave)); b1 = max(b1, __shfl_xor(b1, o, kWave));
            b2 = min(b2, __shfl_xor(b2, o, kWave)); b3 = max(b3, __shfl_xor(b3, o, kWave));
        }
        if (lane == 0 && b1 >= 0) { atomicMin(&s_misc[0], b0); atomicMax(&s_misc[1], b1); atomicMin(&s_misc[2], b2); atomicMax(&s_misc[3], b3); }
        __syncthreads();
        FPC_STAMP(1, 2);
        const int x0 = s_misc[0], x1 = s_misc[1], y0 = s_misc[2], y1 = s_misc[3];
        int ox = 0, oy = 0, rad = W + HW / W;
        if (x1 >= 0) {
            ox = (x0 + x1) / 2; oy = (y0 + y1) / 2;
            rad = max(x1 - ox, ox - x0) + max(y1 - oy, oy - y0);
        }
        const float fox = (float)ox, foy = (float)oy, frad = (float)rad;
        const bool thin = fg > p.max_num;
        // RV/ransac_voting_gpu.py:541-545.  The list keeps every foreground pixel; k_vote_count / k_vote_final re-derive
        // each entry's keep decision, and the built-in sampler draws over all foreground ranks and rejects thinned-out ones
        // (include/fpc_rng.h).  Only injected pair indices (they address the KEPT pixels by rank) and the out_tn diagnostic
        // need the kept image: one wave per chunk, one lane per entry.
        const bool tables = INJ && thin && (p.idxs != nullptr || p.want_tn);
        uint32_t* kw = p.kept_wpre + (size_t)inst * nch * kChunkWords;
        uint64_t* kb = p.kept_bits + (size_t)inst * nch * kChunkWords;
        int tn = thin ? p.max_num : fg;
        if (INJ && tables) {
            for (int c = wv; c < nch; c += nw) {
                const int cnt = cpre[c + 1] - cpre[c];
                int run = 0;
                for (int j = 0; j * kWave < cnt; ++j) {
                    const int e = j * kWave + lane;
                    bool k = false;
                    if (e < cnt) {
                        const float4 q = E[(size_t)c * kChunkPx + e];
                        k = pixel_kept<KEEP>(q.x, q.y, W, HW, inst, fg, p.max_num, p.seed, p.keep);
                    }
                    const uint64_t m = __builtin_amdgcn_ballot_w64(k);
                    if (lane == 0) { kb[(size_t)c * kChunkWords + j] = m; kw[(size_t)c * kChunkWords + j] = (uint32_t)run; }
                    run += __popcll(m);
                }
                if (lane == 0) kpre[c] = run;          // kept entries of the chunk; scanned in place below
            }
            __syncthreads();
            tn = block_scan([&](int c) { return kpre[c]; }, kpre, nch, s_w);
        }
        const bool votes = fg >= p.min_num && tn > 0;     // :536-539
        // count units: 512 consecutive foreground ranks; refinement runs: p.run_entries (the run number indexes the records)
        const int nunits = votes ? (fg + kUnitEntries - 1) / kUnitEntries : 0;
        const int nruns = votes ? (fg + p.run_entries - 1) / p.run_entries : 0;
        if (part == 0 && threadIdx.x == blockDim.x - kWave) {   // the last wave: its wait for the two list bases delays no hypothesis of a short row
            int32_t* pl = p.plan + (size_t)inst * kPlanI;
            pl[0] = fg; pl[1] = votes ? tn : 0; pl[2] = thin ? 1 : 0; pl[3] = ox; pl[4] = oy; pl[5] = rad;
            pl[6] = nruns; pl[7] = votes ? 1 : 0;
            // the two list bases: requested here, consumed after the hypotheses (their latency hides behind that loop)
            s_misc[4] = nunits ? atomicAdd(p.ctrl, nunits) : 0;
            s_misc[5] = nruns ? atomicAdd(p.ctrl + 1, nruns) : 0;
        }
        if (!votes) {                                     // uniform: no unit, no run; k_vote_final writes the zeros
            if (part == 0 && threadIdx.x < kPInfoI) p.pinfo[(size_t)inst * kPInfoI + threadIdx.x] = 0;
            for (int i = 2 * h_lo + threadIdx.x; i < 2 * min(hn, h_hi); i += blockDim.x) p.hyp[(size_t)inst * hn * 2 + i] = 0.0f;   // the out_hyp diagnostic
            __syncthreads();
            continue;
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
                if ((int)kw[(size_t)c * kChunkWords + mid] <= r) lo = mid; else hi = mid;
            }
            return c * kChunkPx + lo * kWave + select64(kb[(size_t)c * kChunkWords + lo], r - (int)kw[(size_t)c * kChunkWords + lo]);
        };
        // built-in sampler of a thinned instance (include/fpc_rng.h): uniform over the kept pixels by rejection
        auto draw_thin = [&](int hi, int which) -> int {
            int slot = 0;
            for (int a = 0; a < FPC_SAMPLE_MAX_TRIES; ++a) {
                slot = rank_slot(fpc_rand_index(p.seed, (uint32_t)inst, (uint32_t)hi, (uint32_t)(which + 2 * a), (uint32_t)fg));
                const float4 q = E[slot];
                if (pixel_kept<KEEP>(q.x, q.y, W, HW, inst, fg, p.max_num, p.seed, p.keep)) break;
            }
            return slot;
        };
        FPC_STAMP(1, 3);
        const float bx0 = (float)x0, bx1 = (float)x1, by0 = (float)y0, by1 = (float)y1;
        for (int h0 = h_lo; h0 < h_hi; h0 += blockDim.x) {
            const int hi = h0 + threadIdx.x;
            if (hi >= h_hi) break;
            // padded hypothesis: margin -4 for every entry (never counted, never undecided)
            float X = 0.0f, Y = 0.0f, S = 0.0f, ES = 4.0f;
            if (hi < hn) {
                float x = 0.0f, y = 0.0f;
                int s0 = -1, s1 = -1;
                if (INJ && p.idxs) {
                    const int t0 = p.idxs[((size_t)inst * hn + hi) * 2], t1 = p.idxs[((size_t)inst * hn + hi) * 2 + 1];
                    if (t0 >= 0 && t0 < tn && t1 >= 0 && t1 < tn) {        // the reference reads out of bounds here
                        s0 = thin ? kept_slot(t0) : rank_slot(t0);
                        s1 = thin ? kept_slot(t1) : rank_slot(t1);
                    }
                } else if (thin) {
                    s0 = draw_thin(hi, 0);
                    s1 = draw_thin(hi, 1);
                } else {
                    s0 = rank_slot(fpc_rand_index(p.seed, (uint32_t)inst, (uint32_t)hi, 0u, (uint32_t)fg));
                    s1 = rank_slot(fpc_rand_index(p.seed, (uint32_t)inst, (uint32_t)hi, 1u, (uint32_t)fg));
                }
                if (s0 >= 0) intersect(E[s0], E[s1], x, y);
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
        FPC_STAMP(1, 4);
        __syncthreads();               // s_misc[4], [5]: the list bases
        FPC_STAMP(1, 5);
        if (part == 0) {
            const int ubase = s_misc[4], rbase = s_misc[5];
            // the records of an instance in the permuted order the progressive count's passes cut (vote.hpp: unit_stride);
            // the exhaustive count takes them in any order
            const int ustride = unit_stride(nunits);
            for (int j = threadIdx.x; j < nunits; j += blockDim.x) {
                const int u = (int)(((long long)j * ustride) % nunits);
                const int r0 = u * kUnitEntries, c = rank_chunk(cpre, nch, r0);
                const int nvalid = min(kUnitEntries, fg - r0);
                p.units[2 * (ubase + j)] = make_int4(inst | (thin ? 1 << 16 : 0) | ((nvalid - 1) << 17), u, c, (ox & 0xffff) | (oy << 16));
                p.units[2 * (ubase + j) + 1] = make_int4(c * kChunkPx + (r0 - cpre[c]), cpre[c + 1] - r0, fg, 0);
            }
            if (threadIdx.x == 0) {
                int4* pi = reinterpret_cast<int4*>(p.pinfo + (size_t)inst * kPInfoI);
                pi[0] = make_int4(ubase, nunits, hn, p.ntiles);
                pi[1] = make_int4(-1, 0, 0, 0);
            }
            for (int rr = threadIdx.x; rr < nruns; rr += blockDim.x)
                p.runs[rbase + rr] = make_int4(inst, rr, rank_chunk(cpre, nch, rr * p.run_entries), fg | (thin ? (int)0x80000000 : 0));
        }
        FPC_STAMP(1, 6);
        __syncthreads();               // s_tab / s_misc are reused by the next instance
    }
}

// ---- k_vote_lead -------------------------------------------------------------------
// The progressive count (fpc_vote_set_prune; only the WINNER is an output of the vote, RV/ransac_voting_gpu.py:566-574):
// the count units of an instance are visited in passes; between two passes this kernel drops every hypothesis that can no
// longer win, exactly:
//   * leader = the alive hypothesis with the largest count so far (lowest index on ties); its inliers among the entries
//     NOT yet counted are counted here with the reference's arithmetic  =>  L = its exact final count, a lower bound of the
//     winner's count;
//   * rem = the valid entries not yet counted: count(h) + rem is an upper bound of h's final count;
//   * h stays alive iff count(h) + rem > L, or == L and h <= leader (an equal count wins only with the lower index:
//     torch.max's first maximum).  A dropped hypothesis keeps its partial count, which is < L or (== L with a higher index
//     than a hypothesis that reaches L): k_vote_final's arg-max over ALL count rows is unchanged.
// The alive hypotheses are compacted in ascending order: slot -> hypothesis map and their B fragments (hypC), the last
// tile padded with the never-voting fragment.  One 1024-thread workgroup per instance.
// dynamic LDS: the instance's chunk prefix [nch + 1] when p.lds_table.
constexpr int kLeadThreads = 1024;
template <bool KEEP>
__global__ __launch_bounds__(kLeadThreads) void k_vote_lead(const VoteParams p, const int pass) {
    extern __shared__ __attribute__((aligned(16))) int s_cpre[];
    __shared__ int s_w[kLeadThreads / kWave + 4];
    __shared__ int s_red[3][kLeadThreads / kWave];
    __shared__ unsigned s_unseen[kProgMaxUnits / 32];
    const int n_act = active_instances(p.n, p.n_dev);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    constexpr int NW = kLeadThreads / kWave;
    const int hn = p.hn, nch = p.nch;
    FPC_STAMP(1, 8 + 8 * pass);
    for (int inst = blockIdx.x; inst < n_act; inst += gridDim.x) {
        int32_t* pi = p.pinfo + (size_t)inst * kPInfoI;
        const int nunits = pi[1], alive_prev = pi[2];
        const int lo = pass_begin(p.pcum, p.npass, pass, nunits);
        if (nunits == 0 || lo >= nunits) continue;                       // uniform: nothing left to count for this instance
        const int32_t* mapp = pass == 1 ? nullptr : p.hmap + ((size_t)((pass - 1) & 1) * p.n + inst) * p.hnp;
        int32_t* mapn = p.hmap + ((size_t)(pass & 1) * p.n + inst) * p.hnp;
        const int32_t* cnt = p.counts + (size_t)inst * p.hnp;
        const int32_t* gpre = p.chunk_pre + (size_t)inst * (nch + 1);
        const int fg = p.plan[(size_t)inst * kPlanI], thin = p.plan[(size_t)inst * kPlanI + 2];
        __syncthreads();                                                  // LDS of the previous instance is free
        if (p.lds_table)
            for (int c = threadIdx.x; c <= nch; c += blockDim.x) s_cpre[c] = gpre[c];
        // which units have not been counted yet: the positions [lo, nunits) of the permuted order, as a bit per unit
        const int ustride = unit_stride(nunits);
        for (int w = threadIdx.x; w < (nunits + 31) / 32; w += blockDim.x) s_unseen[w] = 0u;
        __syncthreads();
        for (int j = lo + threadIdx.x; j < nunits; j += blockDim.x) {
            const int u = (int)(((long long)j * ustride) % nunits);
            atomicOr(&s_unseen[u >> 5], 1u << (u & 31));
        }
        // 1. the leader of the alive set
        int wc = -1, wi = 0x7fffffff;
        for (int sl = threadIdx.x; sl < alive_prev; sl += blockDim.x) {
            const int h = mapp ? mapp[sl] : sl;
            const int cv = cnt[h];
            if (cv > wc || (cv == wc && h < wi)) { wc = cv; wi = h; }
        }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            const int oc = __shfl_xor(wc, o, kWave), oi = __shfl_xor(wi, o, kWave);
            if (oc > wc || (oc == wc && oi < wi)) { wc = oc; wi = oi; }
        }
        if (lane == 0) { s_red[0][wv] = wc; s_red[1][wv] = wi; }
        __syncthreads();
        wc = s_red[0][0]; wi = s_red[1][0];
        for (int i = 1; i < NW; ++i) {
            const int oc = s_red[0][i], oi = s_red[1][i];
            if (oc > wc || (oc == wc && oi < wi)) { wc = oc; wi = oi; }
        }
        FPC_STAMP(1, 8 + 8 * pass + 1);
        const float gx = p.hyp[((size_t)inst * hn + wi) * 2], gy = p.hyp[((size_t)inst * hn + wi) * 2 + 1];
        // 2. the leader's inliers and the valid entries among the ranks of the units not yet counted
        const int32_t* cpre = p.lds_table ? s_cpre : gpre;
        const float4* Lst = p.list + (size_t)inst * p.ls;
        int lc = 0, rem = 0, c = 0;                                     // c: this lane's chunk cursor (its ranks only grow)
        for (int r0 = threadIdx.x; r0 < fg; r0 += 4 * kLeadThreads) {
            float4 q[4];
            bool on[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int rk = r0 + j * kLeadThreads;
                q[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                on[j] = rk < fg && ((s_unseen[rk / (kUnitEntries * 32)] >> ((rk / kUnitEntries) & 31)) & 1u) != 0u;
                if (on[j]) q[j] = Lst[rank_slot_from(cpre, c, rk)];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bool valid = on[j];
                if (valid && thin) valid = pixel_kept<KEEP>(q[j].x, q[j].y, p.W, p.HW, inst, fg, p.max_num, p.seed, p.keep);
                if (valid) {
                    ++rem;
                    if (pair_is_inlier(q[j].x, q[j].y, q[j].z, q[j].w, sqrtf(q[j].z * q[j].z + q[j].w * q[j].w), gx, gy, p.thresh)) ++lc;
                }
            }
        }
        lc = wave_reduce_add(lc); rem = wave_reduce_add(rem);
        __syncthreads();                                                  // s_red of step 1 has been read
        if (lane == 0) { s_red[0][wv] = lc; s_red[1][wv] = rem; }
        __syncthreads();
        lc = 0; rem = 0;
        for (int i = 0; i < NW; ++i) { lc += s_red[0][i]; rem += s_red[1][i]; }
        const int L = wc + lc;
        FPC_STAMP(1, 8 + 8 * pass + 2);
        // 3. prune and compact, in slot order (ascending hypothesis index)
        int base = 0;
        const u32x4* Bo = p.hypB + (size_t)inst * p.ntiles * kWave;
        u32x4* Bc = p.hypC + (size_t)inst * p.ntiles * kWave;
        for (int s0 = 0; s0 < alive_prev; s0 += blockDim.x) {
            const int sl = s0 + threadIdx.x;
            int h = -1;
            bool keep = false;
            if (sl < alive_prev) {
                h = mapp ? mapp[sl] : sl;
                const int ub = cnt[h] + rem;
                keep = ub > L || (ub == L && h <= wi);
            }
            int wt;
            const int ex = wave_excl_scan(keep ? 1 : 0, wt);
            __syncthreads();
            if (lane == 0) s_w[wv] = wt;
            __syncthreads();
            int off = base, tile = 0;
            for (int k = 0; k < NW; ++k) { const int x = s_w[k]; if (k < wv) off += x; tile += x; }
            if (keep) {
                const int ns = off + ex;
                mapn[ns] = h;
                Bc[(ns / kHypTile) * kWave + ns % kHypTile] = Bo[(h / kHypTile) * kWave + h % kHypTile];
                Bc[(ns / kHypTile) * kWave + ns % kHypTile + kHypTile] = Bo[(h / kHypTile) * kWave + h % kHypTile + kHypTile];
            }
            base += tile;
        }
        const int tiles = (base + kHypTile - 1) / kHypTile;
        {   // the padding of the last tile: margin -4 for every entry (never counted, never undecided)
            u32x4 plo, phi;
            b_fragment(0.0f, 0.0f, 0.0f, 4.0f, plo, phi);
            for (int ns = base + threadIdx.x; ns < tiles * kHypTile; ns += blockDim.x) {
                mapn[ns] = -1;
                Bc[(ns / kHypTile) * kWave + ns % kHypTile] = plo;
                Bc[(ns / kHypTile) * kWave + ns % kHypTile + kHypTile] = phi;
            }
        }
        if (threadIdx.x == 0) { pi[2] = base; pi[3] = tiles; pi[4] = wi; pi[5] = L; pi[6] = rem; }
        FPC_STAMP(1, 8 + 8 * pass + 3);
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

constexpr int kFinWaves = 8;         // 512-thread workgroups: at 120 VGPRs two fit a CU, i.e. 512 resident tasks (1024 threads: one per CU,
                                     // and the 370 runs of the B = 32 batch took two rounds)

// Winner (largest count, lowest index: torch.max, RV/ransac_voting_gpu.py:567), its inliers voted again (:583-589) with
// the reference's arithmetic, the fp64 normal equations and the 2x2 solve (:592-599).  Task = one run of p.run_entries
// consecutive foreground ranks of an instance, 512 lanes, four loads in flight each.  An instance of one run (the usual
// case) is solved by its task; otherwise the run whose arrival ticket comes last combines the instance's records
// (Guideline 16, counter form: records stored write-through (sc1), the storing wave drained, one agent-scope add per
// workgroup; the last arriver reads them back with sc1 loads, in run order: bit-reproducible).
// dynamic LDS: the instance's chunk prefix [nch + 1] when p.lds_table.
template <bool KEEP>
__global__ __launch_bounds__(64 * kFinWaves) void k_vote_final(const VoteParams p) {
    extern __shared__ __attribute__((aligned(16))) int s_cpre[];
    __shared__ int s_red[2 * kFinWaves];
    __shared__ float2 s_pt[kFinWaves];
    __shared__ int s_last;
    __shared__ double s_part[kFinWaves][kRec];
    const int n_act = active_instances(p.n, p.n_dev);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const int hn = p.hn, nch = p.nch;
    // instances that do not vote (fewer than min_num pixels) have no run: zeros (RV/ransac_voting_gpu.py:536-539)
    for (int inst = blockIdx.x * blockDim.x + threadIdx.x; inst < n_act; inst += gridDim.x * blockDim.x)
        if (p.plan[(size_t)inst * kPlanI + 6] == 0) {
            p.out_xy[inst * 2] = 0.0f; p.out_xy[inst * 2 + 1] = 0.0f;
            if (p.pose_RT) pose_rt_one((size_t)inst, 0.0f, 0.0f, p.pose_q, p.pose_z, p.pose_kinv, p.pose_R, p.pose_T, p.pose_RT);
            if (p.out_tn) p.out_tn[inst] = p.plan[(size_t)inst * kPlanI + 1];
            if (p.out_win_idx) p.out_win_idx[inst] = -1;
            if (p.out_win_count) p.out_win_count[inst] = 0;
            if (p.out_inl) p.out_inl[inst] = 0;
            if (p.out_refine)
                for (int i = 0; i < 8; ++i) p.out_refine[(size_t)inst * 8 + i] = 0.0;
        }
    FPC_STAMP(3, 0);
    // the first task's record is requested together with the task count (the grid never exceeds the record array: a
    // slot past the count holds stale data that is not used), and an instance's chunk table together with its counts and
    // hypotheses: two dependent round trips less in a kernel that is a chain of them
    int4 rb = p.runs[blockIdx.x];
    const int nr = p.ctrl[1];
    for (int t = blockIdx.x; t < nr; t += gridDim.x) {
        if (t != (int)blockIdx.x) rb = p.runs[t];
        const int inst = rb.x, run = rb.y, c_lo = rb.z;
        const int fg = rb.w & 0x7fffffff, nrec = (fg + p.run_entries - 1) / p.run_entries;
        const bool thin = rb.w < 0;
        const int32_t* gpre = p.chunk_pre + (size_t)inst * (nch + 1);
        // winner: every task of the instance finds the same one; its point travels with it (no dependent load afterwards)
        int wc = -1, wi = 0x7fffffff;
        float wx = 0.0f, wy = 0.0f;
        const float* hp = p.hyp + (size_t)inst * hn * 2;
        const bool table_in_reg = p.lds_table && nch < (int)blockDim.x;
        int cpre_reg = 0;
        if (table_in_reg && (int)threadIdx.x <= nch) cpre_reg = gpre[threadIdx.x];
        for (int h = threadIdx.x; h < hn; h += blockDim.x) {
            const int cv = p.counts[(size_t)inst * p.hnp + h];
            const float2 g = *reinterpret_cast<const float2*>(hp + 2 * h);
            if (cv > wc) { wc = cv; wi = h; wx = g.x; wy = g.y; }        // ascending h: first maximum kept
        }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            const int oc = __shfl_xor(wc, o, kWave), oi = __shfl_xor(wi, o, kWave);
            const float ox = __shfl_xor(wx, o, kWave), oy = __shfl_xor(wy, o, kWave);
            if (oc > wc || (oc == wc && oi < wi)) { wc = oc; wi = oi; wx = ox; wy = oy; }
        }
        __syncthreads();                                               // LDS of the previous task is free
        if (lane == 0) { s_red[wv] = wc; s_red[kFinWaves + wv] = wi; s_pt[wv] = make_float2(wx, wy); }
        if (table_in_reg) {
            if ((int)threadIdx.x <= nch) s_cpre[threadIdx.x] = cpre_reg;
        } else if (p.lds_table) {
            for (int c = threadIdx.x; c <= nch; c += blockDim.x) s_cpre[c] = gpre[c];
        }
        __syncthreads();
        wc = s_red[0]; wi = s_red[kFinWaves];
        {
            int best = 0;
            for (int i = 1; i < kFinWaves; ++i) {
                const int oc = s_red[i], oi = s_red[kFinWaves + i];
                if (oc > wc || (oc == wc && oi < wi)) { wc = oc; wi = oi; best = i; }
            }
            wx = s_pt[best].x; wy = s_pt[best].y;
        }
        // no hypothesis with an inlier: all_win_pts stays (0,0) (:571-574) and the refinement votes for (0,0)
        if (!(wc > 0)) { wi = -1; wc = 0; wx = 0.0f; wy = 0.0f; }

        FPC_STAMP(3, 1);
        const int32_t* cpre = p.lds_table ? s_cpre : gpre;
        const float4* L = p.list + (size_t)inst * p.ls;
        const int r_end = min(fg, (run + 1) * p.run_entries);
        double v[kRec] = {0, 0, 0, 0, 0, 0};                            // inliers, a00, a01, a11, b0, b1
        int c = c_lo;                                                   // this lane's chunk cursor: its ranks only grow
        for (int r0 = run * p.run_entries + threadIdx.x; r0 < r_end; r0 += 4 * 64 * kFinWaves) {
            float4 q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                               // four independent loads per lane in flight
                const int rk = r0 + j * 64 * kFinWaves;
                q[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rk < r_end) q[j] = L[rank_slot_from(cpre, c, rk)];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bool valid = r0 + j * 64 * kFinWaves < r_end;
                if (valid && thin) valid = pixel_kept<KEEP>(q[j].x, q[j].y, p.W, p.HW, inst, fg, p.max_num, p.seed, p.keep);
                if (valid && pair_is_inlier(q[j].x, q[j].y, q[j].z, q[j].w, sqrtf(q[j].z * q[j].z + q[j].w * q[j].w), wx, wy, p.thresh)) {
                    const double nx = (double)q[j].w, ny = -(double)q[j].z;      // normal = (dy, -dx) :584-586
                    const double bb = nx * (double)q[j].x + ny * (double)q[j].y;
                    v[0] += 1.0; v[1] += nx * nx; v[2] += nx * ny; v[3] += ny * ny; v[4] += nx * bb; v[5] += ny * bb;
                }
            }
        }
        FPC_STAMP(3, 2);
#pragma unroll
        for (int a = 0; a < kRec; ++a) {
            const double r = wave_reduce_add(v[a]);
            if (lane == 0) s_part[wv][a] = r;
        }
        __syncthreads();
        FPC_STAMP(3, 3);
        double tot[kRec] = {0, 0, 0, 0, 0, 0};
        if (wv == 0) {                                                  // wave 0 finishes the task
#pragma unroll
            for (int a = 0; a < kRec; ++a) {
                tot[a] = s_part[0][a];
#pragma unroll
                for (int w = 1; w < kFinWaves; ++w) tot[a] += s_part[w][a];                 // fixed order
            }
        }
        if (nrec > 1) {                                                 // uniform
            if (threadIdx.x < kRec)
                store_wt64(p.partial + ((size_t)inst * p.nrx + run) * kRec + threadIdx.x, __builtin_bit_cast(unsigned long long, tot[threadIdx.x]));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the storing wave drains its sc1 stores
            __syncthreads();
            if (threadIdx.x == 0) {
                const int tk = __hip_atomic_fetch_add(p.tickets + inst, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = (tk == nrec - 1);
            }
            __syncthreads();
            if (!s_last) continue;                                     // uniform
            if (wv != 0) continue;                                     // wave 0 finishes
            // last arriver of the instance: the records in run order (independent sc1 loads)
            // lane = (record slot r8 = lane / 8, value a = lane % 8): eight records per sweep, then a fixed-order lane tree
            const int a = lane & 7, r8 = lane >> 3;
            double acc = 0.0;
            for (int b = r8; b < nrec; b += 32) {
                unsigned long long x[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    x[i] = (a < kRec && b + 8 * i < nrec) ? load_wt64(p.partial + ((size_t)inst * p.nrx + b + 8 * i) * kRec + a) : 0ull;
#pragma unroll
                for (int i = 0; i < 4; ++i) acc += __builtin_bit_cast(double, x[i]);
            }
            acc += __shfl_xor(acc, 8, kWave); acc += __shfl_xor(acc, 16, kWave); acc += __shfl_xor(acc, 32, kWave);
#pragma unroll
            for (int i = 0; i < kRec; ++i) tot[i] = __shfl(acc, i, kWave);
        }
        if (threadIdx.x == 0) {
            double x0, x1;
            solve2_sym(tot[1], tot[2], tot[3], tot[4], tot[5], x0, x1);
            p.out_xy[inst * 2] = (float)x0;
            p.out_xy[inst * 2 + 1] = (float)x1;
            if (p.pose_RT) pose_rt_one((size_t)inst, (float)x0, (float)x1, p.pose_q, p.pose_z, p.pose_kinv, p.pose_R, p.pose_T, p.pose_RT);
            if (p.out_tn) p.out_tn[inst] = p.plan[(size_t)inst * kPlanI + 1];
            if (p.out_win_idx) p.out_win_idx[inst] = wi;
            if (p.out_win_count) p.out_win_count[inst] = wc;
            if (p.out_inl) p.out_inl[inst] = (int)tot[0];
            if (p.out_refine) {      // what the refinement's backward needs (fpc_vote_refine_backward)
                double* r = p.out_refine + (size_t)inst * 8;
                r[0] = (double)wx; r[1] = (double)wy; r[2] = tot[1]; r[3] = tot[2]; r[4] = tot[3]; r[5] = tot[4];
                r[6] = tot[5]; r[7] = tot[0];
            }
        }
        FPC_STAMP(3, 4);
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

// the progressive count's switch and schedule (process-wide; read at every call)
static std::atomic<int> g_prune_mode{0};                   // 0: never (default), 1: whenever the count rows are not an output
static std::atomic<int> g_prune_passes{3};
static std::atomic<int> g_prune_cum[kMaxPasses + 1] = {{0}, {5}, {10}, {16}, {16}};

extern "C" int fpc_vote_set_prune(int mode, int npass, const int32_t* cum16) {
    if (mode < 0 || mode > 1) return FPC_EINVAL;
    if (npass != 0) {
        if (npass < 2 || npass > kMaxPasses || !cum16) return FPC_EINVAL;
        int prev = 0;
        for (int i = 1; i < npass; ++i) {
            if (cum16[i - 1] <= prev || cum16[i - 1] >= 16) return FPC_EINVAL;
            prev = cum16[i - 1];
        }
        for (int i = 1; i < npass; ++i) g_prune_cum[i].store(cum16[i - 1], std::memory_order_relaxed);
        g_prune_passes.store(npass, std::memory_order_relaxed);
    }
    g_prune_mode.store(mode, std::memory_order_relaxed);
    return FPC_OK;
}

extern "C" int fpc_vote_prune_info(const void* ws, size_t ws_bytes, int n, int H, int W, int hn, int32_t* out, fpc_stream_t stream) {
    if (n < 1 || H < 1 || W < 1 || hn < 1 || !ws || !out) return FPC_EINVAL;
    Ws w = carve(const_cast<void*>(ws), n, H, W, hn);
    if (ws_bytes < w.total) return FPC_EWORKSPACE;
    clear_hip_error();
    hipError_t e = hipMemcpyAsync(out, w.p.pinfo, sizeof(int32_t) * (size_t)n * kPInfoI, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    return FPC_OK;
}

extern "C" size_t fpc_ransac_workspace_bytes(int n, int H, int W, int hn) {
    if (n <= 0 || H < 1 || W < 1 || hn < 1) return 256;
    return carve(nullptr, n, H, W, hn).total;
}

extern "C" size_t fpc_mask_bits_words(int H, int W) {
    return (H < 1 || W < 1) ? 0 : (size_t)cdiv(H * W, kChunkPx) * kChunkWords;
}

extern "C" int fpc_ransac_voting_v3(const float* mask, const float* vertex, int64_t vs_n, int64_t vs_h, int64_t vs_w,
                                    int64_t vs_c, int n, const int32_t* n_dev, int H, int W, int hn,
                                    const int32_t* idxs, const uint8_t* keep, uint64_t seed, float inlier_thresh,
                                    int min_num, int max_num, float* out_xy, int32_t* out_tn, int32_t* out_win_idx,
                                    int32_t* out_win_count, int32_t* out_inl_count, float* out_hyp,
                                    int32_t* out_counts, double* out_refine, void* ws, size_t ws_bytes,
                                    fpc_stream_t stream) {
    return fpc_ransac_voting_v3_bits(mask, nullptr, vertex, vs_n, vs_h, vs_w, vs_c, n, n_dev, H, W, hn, idxs, keep, seed, inlier_thresh,
                                     min_num, max_num, out_xy, out_tn, out_win_idx, out_win_count, out_inl_count, out_hyp, out_counts,
                                     out_refine, ws, ws_bytes, stream);
}

extern "C" int fpc_ransac_voting_v3_bits(const float* mask, const uint64_t* mask_bits, const float* vertex, int64_t vs_n,
                                         int64_t vs_h, int64_t vs_w, int64_t vs_c, int n, const int32_t* n_dev, int H, int W, int hn,
                                         const int32_t* idxs, const uint8_t* keep, uint64_t seed, float inlier_thresh,
                                         int min_num, int max_num, float* out_xy, int32_t* out_tn, int32_t* out_win_idx,
                                         int32_t* out_win_count, int32_t* out_inl_count, float* out_hyp,
                                         int32_t* out_counts, double* out_refine, void* ws, size_t ws_bytes,
                                         fpc_stream_t stream) {
    return fpc_ransac_voting_v3_pose(mask, mask_bits, vertex, vs_n, vs_h, vs_w, vs_c, n, n_dev, H, W, hn, idxs, keep, seed, inlier_thresh,
                                     min_num, max_num, out_xy, out_tn, out_win_idx, out_win_count, out_inl_count, out_hyp, out_counts,
                                     out_refine, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes, stream);
}

extern "C" int fpc_ransac_voting_v3_pose(const float* mask, const uint64_t* mask_bits, const float* vertex, int64_t vs_n,
                                         int64_t vs_h, int64_t vs_w, int64_t vs_c, int n, const int32_t* n_dev, int H, int W, int hn,
                                         const int32_t* idxs, const uint8_t* keep, uint64_t seed, float inlier_thresh,
                                         int min_num, int max_num, float* out_xy, int32_t* out_tn, int32_t* out_win_idx,
                                         int32_t* out_win_count, int32_t* out_inl_count, float* out_hyp,
                                         int32_t* out_counts, double* out_refine, const float* pose_q, const float* pose_z,
                                         const float* pose_kinv, float* pose_R, float* pose_T, float* pose_RT, void* ws,
                                         size_t ws_bytes, fpc_stream_t stream) {
    const bool pose = pose_q || pose_z || pose_kinv || pose_R || pose_T || pose_RT;
    if (pose && !(pose_q && pose_z && pose_kinv && pose_R && pose_T && pose_RT)) return FPC_EINVAL;      // all six or none
    if (n < 0 || H < 1 || W < 1 || hn < 1 || hn > kMaxHn || max_num < 1) return FPC_EINVAL;
    if ((int64_t)H * W > (1 << 30) || H > 65535 || W > 65535) return FPC_EINVAL;
    if (n == 0) return FPC_OK;
    if ((!mask && !mask_bits) || !vertex || !out_xy || !ws || ((uintptr_t)mask_bits & 7)) return FPC_EINVAL;
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
    p.pose_q = pose_q; p.pose_z = pose_z; p.pose_kinv = pose_kinv; p.pose_R = pose_R; p.pose_T = pose_T; p.pose_RT = pose_RT;
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
    p.lds_table = p.nch + 1 <= 2048 ? 1 : 0;              // the chunk tables of an instance in LDS (<= 8 KB each)
    // Only the winner is an output unless the caller asks for the count rows: the progressive count (k_vote_lead) then
    // skips the pairs of hypotheses that cannot win.  OFF unless asked for: on the 32-frame / hn = 1000 batch it evaluates
    // 0.55 of the pairs and still takes 326 us against 283 (profiles/r05_vote_prune.md).
    const int prune_mode = g_prune_mode.load(std::memory_order_relaxed);
    const bool prune_fits = !out_counts && n <= kProgMaxInst && p.nux <= kProgMaxUnits;
    const bool prune = prune_fits && prune_mode > 0;
    p.npass = 1;
    for (int i = 0; i <= kMaxPasses; ++i) p.pcum[i] = 16;
    p.pcum[0] = 0;
    if (prune) {
        p.npass = g_prune_passes.load(std::memory_order_relaxed);
        for (int i = 1; i < p.npass; ++i) p.pcum[i] = g_prune_cum[i].load(std::memory_order_relaxed);
    }
    const size_t table_lds = p.lds_table ? (size_t)(p.nch + 1) * sizeof(int) : 0;

#ifdef FPC_VOTE_TRACE     // diagnostic build only (python -c "build(extra=['-DFPC_VOTE_TRACE'])"): name the launch that faults
#define FPC_TRACE(what) do { hipError_t te = hipStreamSynchronize(s); fprintf(stderr, "[fpc vote] %s done: %s\n", what, hipGetErrorString(te)); fflush(stderr); } while (0)
#else
#define FPC_TRACE(what) do { } while (0)
#endif
    // 1. mask planes -> per-chunk compacted pixel lists (the only pass over the masks and the vote planes)
    const bool vec4 = (HW % 4 == 0) && (((uintptr_t)mask & 15) == 0);
    const bool vg4 = (HW % 4 == 0) && W % 4 == 0 && vs_w == 1 && vs_h % 4 == 0 && vs_n % 4 == 0 && vs_c % 4 == 0 &&
                     (((uintptr_t)vertex & 15) == 0);
#ifndef FPC_SCAN_WGS             // (diagnostic builds sweep it: tools_dev/scan_sweep.sh)
#define FPC_SCAN_WGS (256 * 8)   // 32 frames, hn = 128: 768: 153.4 us, 1024: 147.9, 1536: 156.4, 2048: 146.9 per vote (f32 masks)
#endif
    const int scan_grid = (int)std::min<long long>((long long)n * p.nch, FPC_SCAN_WGS);       // resident: the loop prefetches
    const ScanParams sp{mask, mask_bits, vertex, vs_n, vs_h, vs_w, vs_c, n, n_dev, W, HW, p.nch, p.ls, p.ctrl, p.chunk_fg, p.chunk_box, p.list, p.stamps};
#define FPC_LAUNCH_SCAN(A, B) hipLaunchKernelGGL((k_vote_scan<A, B>), dim3(scan_grid), dim3(256), 0, s, sp)
    if (mask_bits) {      // the foreground as bit words: the f32 planes are not read
        if (vg4) hipLaunchKernelGGL((k_vote_scan<false, true, true>), dim3(scan_grid), dim3(256), 0, s, sp);
        else hipLaunchKernelGGL((k_vote_scan<false, false, true>), dim3(scan_grid), dim3(256), 0, s, sp);
    } else if (vg4 && vec4) FPC_LAUNCH_SCAN(true, true); else if (vec4) FPC_LAUNCH_SCAN(true, false); else FPC_LAUNCH_SCAN(false, false);
#undef FPC_LAUNCH_SCAN
    FPC_TRACE("scan");

    // 2. per instance: prefix, origin, units and runs, hypotheses (+ their B fragments), zeroed count row and ticket
    const dim3 plan_grid(std::min(n, 2048), std::min(p.ntiles, 4)), plan_one(std::min(n, 2048), 1);
    if (keep) hipLaunchKernelGGL((k_vote_plan<true, true>), plan_one, dim3(256), 2 * table_lds, s, p);
    else if (idxs || out_tn) hipLaunchKernelGGL((k_vote_plan<true, false>), plan_one, dim3(256), 2 * table_lds, s, p);
    else hipLaunchKernelGGL((k_vote_plan<false, false>), plan_grid, dim3(256), 2 * table_lds, s, p);
    FPC_TRACE("plan");

    // 3. exact inlier counts of every hypothesis.  One resident round of workgroups (four per CU); the kernel reads how
    //    many units exist and cuts the hypothesis tiles into slices so that the tasks fill that round evenly.
    const long long cap_tasks = (long long)n * p.nux * p.ntiles;
    const int resident = 256 * 4;
    const int count_grid = (int)std::min<long long>(cap_tasks, resident);
    p.task_target = count_grid;
    const size_t count_lds = (size_t)std::min(p.ntiles, kMaxSliceTiles) * kHypTile * sizeof(int);
    if (p.npass > 1) {
        // the progressive count: passes over disjoint unit sets, k_vote_lead drops the hypotheses that cannot win in between
        for (int pass = 0; pass < p.npass; ++pass) {
            if (pass > 0) {
                if (keep) hipLaunchKernelGGL(k_vote_lead<true>, dim3(std::min(n, 2048)), dim3(kLeadThreads), table_lds, s, p, pass);
                else hipLaunchKernelGGL(k_vote_lead<false>, dim3(std::min(n, 2048)), dim3(kLeadThreads), table_lds, s, p, pass);
                FPC_TRACE("lead");
            }
            launch_vote_count_prog(p, pass, count_grid, s);
            FPC_TRACE("count pass");
        }
    } else {
        launch_vote_count(p, count_grid, count_lds, s);
        FPC_TRACE("count");
    }

    // 4. winner, its inliers, refinement: one task per run of chunks
    const int fin_grid = (int)std::min<long long>(std::max<long long>((long long)n * p.nrx, 1), 512);
    if (keep) hipLaunchKernelGGL(k_vote_final<true>, dim3(fin_grid), dim3(64 * kFinWaves), table_lds, s, p);
    else hipLaunchKernelGGL(k_vote_final<false>, dim3(fin_grid), dim3(64 * kFinWaves), table_lds, s, p);
    FPC_TRACE("final");

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

