// ransac.hip — PVNet-derived RANSAC hough voting for gfx950.
//
//  * fpc_generate_hypothesis / fpc_voting_for_hypothesis: B1-compatible kernels
//    (reference: RV/src/ransac_voting_kernel.cu:11-167).
//  * fpc_ransac_voting_v3: the whole of ransac_voting_layer_v3 (RV/ransac_voting_gpu.py:518-607) for a batch of
//    instances in FOUR stateless launches, without a host round trip and without the hn x tn inlier matrix:
//
//      k_vote_scan    task = (instance, chunk of 4096 pixels): the only pass over the caller's planes.  Mask -> bit words,
//                     in-chunk prefix, chunk count and bounding box; the chunk's foreground pixels are compacted in order
//                     into its own slots of two float4 lists: {x, y, dx, dy} (vote gathered through the caller's
//                     strides) and the pixel's filter constants.            (HBM: n x 12 H W bytes read: at the roofline)
//      k_vote_plan    one 1024-thread workgroup per instance: chunk prefix (rank -> slot), the > max_num thinning
//                     (:541-545), the integer origin / radius the filter's coordinates are measured from, the hn
//                     hypotheses (:552,559; pair sampling, two-line intersection exactly as .cu:28-45) as SoA rows, the
//                     rounding allowance E_g per 64 of them, a zeroed count row, and the WORK UNITS: one (instance, block of
//                     512 list entries) record per block that has entries, appended to a device-side list.
//      k_vote_count   one resident round of workgroups over units x hypothesis slices (the slicing is chosen on the device
//                     from the unit count).  EXACT inlier counts: per (entry, point) the margins to two cones, their sign
//                     bits shifted into per-lane bit rows; after 64 points a 64 x 64 bit transpose across the wave and
//                     v_bcnt give the counts; the pairs between the cones are queued and take the reference's own
//                     arithmetic (.cu:106-125) 64 at a time.  Integer atomics per (block, hypothesis).
//      k_vote_final   task = work unit: torch.max's winner (:567, first maximal index), its inliers voted again, fp64
//                     normal-equation records; the unit of an instance that arrives last sums them in unit order and
//                     solves the 2x2 system in closed form (b_inv, :503-516, :583-599).
//
// Why the cones are sound: the reference accepts a pair when fl(cos) > th, where fl(cos) carries at most
// 8 ulp(1) < 1e-6 of rounding.  So an accepted pair has true cos >= th' = th - 1e-6, i.e. |s| <= kappa' t with
// t = d.e, s = d x e (e the unit vote, d = h - p), kappa' = sqrt(1-th'^2)/th' ("maybe"), and a pair with true
// cos >= th'' = th + 1e-6, i.e. |s| <= kappa'' t, is accepted for sure.  t and s are affine in the hypothesis (two
// FMAs each against per-pixel constants, all measured from the instance's integer origin so that magnitudes stay
// small); the evaluation's own rounding is at most (3.6e-7 + 4.8e-7 kappa) M with M = |gx - ox| + |gy - oy| + radius
// (six roundings at magnitude <= M on the s side, eight at kappa M on the t side):
//     accepted  =>  |s| <= kappa' t + E        (computed values);        |s| <= kappa'' t - E  =>  accepted
// for any E >= that bound; E_g = max over the 64 hypotheses of a group of 2e-6 (1 + kappa') M.  The compares are taken
// as sign bits of differences (x - y >= 0 exactly when y <= x in IEEE arithmetic).  A group with a huge or non-finite
// point is outside the filter's domain: every pair of it takes the reference's arithmetic.  Thresholds <= 2e-6 have no
// cone: kModeReference.
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

constexpr int kChunkPx = 4096;       // pixels per k_vote_scan task; a chunk owns list slots [c * 4096, c * 4096 + its count)
constexpr int kChunkWords = 64;      // 64-pixel words per chunk
constexpr int kBlockPx = 512;        // list entries per k_vote_count / k_vote_exact task: 4 waves x 2 tiles of 64
constexpr int kPlanI = 8;            // i32 per instance: fg, tn, thinned, origin x, origin y, radius
constexpr int kMaxHn = 65536;
constexpr int kRec = 6;              // doubles per refinement record: inliers, a00, a01, a11, b0, b1

struct Ws {
    int32_t* plan;        // [n, kPlanI]
    int32_t* chunk_fg;    // [n, nch]       foreground count per chunk (k_vote_plan overwrites it with the kept count when thinning)
    int32_t* chunk_pre;   // [n, nch + 1]   exclusive prefix of the foreground counts: rank -> chunk
    uint32_t* word_pre;   // [n, nwords]    exclusive count of the word inside its chunk
    uint64_t* bits;       // [n, nwords]    1 bit per foreground pixel
    int32_t* chunk_preK;  // [n, nch + 1]   the same three over the KEPT pixels of a thinned instance (hypothesis sampling only)
    uint32_t* word_preK;  // [n, nwords]
    uint64_t* bitsK;      // [n, nwords]
    int32_t* chunk_box;   // [n, nch, 4]    x min / max, y min / max of the chunk's foreground pixels
    float* hx;            // [n, hnp]       hypothesis points, SoA (hnp = hn rounded up to 64)
    float* hy;            // [n, hnp]
    float* hxs;           // [n, hnp]       the same minus the instance's origin (centre of its bounding box)
    float* hys;           // [n, hnp]
    float* eg;            // [n, hnp / 64]  E_g per group of 64 hypotheses
    float* hyp;           // [n, hn, 2]     the same points as the reference's [hn,1,2] tensor
    int32_t* upper;       // [n, hnp]       exact inlier count of every hypothesis; zeroed by k_vote_plan
    int32_t* tickets;     // [n]            k_vote_final arrivals; zeroed by k_vote_plan
    double* partial;      // [n, nbx, kRec] k_vote_final per-task records
    float4* list;         // [n, HW]        {x, y, dx, dy} of the foreground pixels, compacted per chunk (k_vote_scan)
    float4* clist;        // [n, HW]        {ey, -ex, cs, ct}: their filter constants (NaN cs: never an inlier)
    int4* units;          // [n * nbx]      {instance, block of 512 list entries, fg | thinned << 31, ox | oy << 16} of every
                          //                block that has entries (k_vote_plan): all a count task needs to start loading
    int32_t* n_units;     // [1]            how many; zeroed by k_vote_scan
    int nch, nwords, hnp, nbx;
    size_t total;
};

static Ws carve(void* base, int n, int H, int W, int hn) {
    Ws w;
    size_t HW = (size_t)H * W;
    w.nch = cdiv((int)HW, kChunkPx);
    w.nwords = w.nch * kChunkWords;
    w.hnp = cdiv(hn, kWave) * kWave;
    w.nbx = cdiv((int)HW, kBlockPx);
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* q = p + off; off = align_up(off + bytes, 256); return q; };
    w.plan = (int32_t*)take(sizeof(int32_t) * (size_t)n * kPlanI);
    w.chunk_fg = (int32_t*)take(sizeof(int32_t) * (size_t)n * w.nch);
    w.chunk_pre = (int32_t*)take(sizeof(int32_t) * (size_t)n * (w.nch + 1));
    w.word_pre = (uint32_t*)take(sizeof(uint32_t) * (size_t)n * w.nwords);
    w.bits = (uint64_t*)take(sizeof(uint64_t) * (size_t)n * w.nwords);
    w.chunk_preK = (int32_t*)take(sizeof(int32_t) * (size_t)n * (w.nch + 1));
    w.word_preK = (uint32_t*)take(sizeof(uint32_t) * (size_t)n * w.nwords);
    w.bitsK = (uint64_t*)take(sizeof(uint64_t) * (size_t)n * w.nwords);
    w.chunk_box = (int32_t*)take(sizeof(int32_t) * (size_t)n * w.nch * 4);
    w.hx = (float*)take(sizeof(float) * (size_t)n * w.hnp);
    w.hy = (float*)take(sizeof(float) * (size_t)n * w.hnp);
    w.hxs = (float*)take(sizeof(float) * (size_t)n * w.hnp);
    w.hys = (float*)take(sizeof(float) * (size_t)n * w.hnp);
    w.eg = (float*)take(sizeof(float) * (size_t)n * (w.hnp / kWave));
    w.hyp = (float*)take(sizeof(float) * (size_t)n * hn * 2);
    w.upper = (int32_t*)take(sizeof(int32_t) * (size_t)n * w.hnp);
    w.tickets = (int32_t*)take(sizeof(int32_t) * (size_t)n);
    w.partial = (double*)take(sizeof(double) * (size_t)n * w.nbx * kRec);
    w.list = (float4*)take(sizeof(float4) * (size_t)n * HW);
    w.clist = (float4*)take(sizeof(float4) * (size_t)n * HW);
    w.units = (int4*)take(sizeof(int4) * (size_t)n * w.nbx);
    w.n_units = (int32_t*)take(sizeof(int32_t));
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

// The filter constants of one pixel q = {x, y, dx, dy}: {a_s = ey, b_s = -ex, cs, ct} with e the unit vote, so that for a
// point g:  s = d x e = a_s gx + b_s gy + cs  and  t = d . e = -b_s gx + a_s gy + ct  (d = g - p).  A vote that the
// reference skips (norm < 1e-6, .cu:121) or that is not finite gets NaN constants: no compare ever accepts it.
// k_vote_scan stores them for the frame origin; the count / exact kernels re-centre them on the instance
// (recentre_constants), which shrinks every magnitude the rounding allowance E is proportional to.
__device__ __forceinline__ float4 pixel_constants(float4 q) {
    const float qnan = __builtin_nanf("");
    const float n1 = sqrtf(q.z * q.z + q.w * q.w);
    float4 c = make_float4(0.f, 0.f, qnan, qnan);
    if (!below_eps(n1) && n1 <= 3.0e38f) {
        const float ex = q.z / n1, ey = q.w / n1;
        c.x = ey; c.y = -ex;
        c.z = -(q.x * ey - q.y * ex);
        c.w = -(q.x * ex + q.y * ey);
    }
    return c;
}

// The same constants for coordinates measured from the integer origin (ox, oy): the unit vote is kept, cs / ct are
// recomputed from the exactly shifted pixel (q.x - ox and q.y - oy are exact: small integers).
__device__ __forceinline__ float4 recentre_constants(float4 cst, float4 q, float ox, float oy) {
    if (cst.z != cst.z) return cst;                                  // NaN: the pixel never votes
    const float xs = q.x - ox, ys = q.y - oy, ey = cst.x, ex = -cst.y;
    return make_float4(cst.x, cst.y, -(xs * ey - ys * ex), -(xs * ex + ys * ey));
}

// ---- k_vote_scan -------------------------------------------------------------------
// grid-stride over (instance, chunk) tasks; 256 threads; a chunk = 4096 pixels = 4 float4 of the mask per lane.
// Writes the chunk's bit words / in-chunk prefix / count, and compacts its foreground pixels (vote gathered from the
// caller's strided planes, filter constants computed once) into the chunk's own slots of the two lists.
template <bool VEC4, bool VGATHER4>
__global__ __launch_bounds__(256) void k_vote_scan(const float* __restrict__ mask, const float* __restrict__ vertex,
                                                   int64_t vs_n, int64_t vs_h, int64_t vs_w, int64_t vs_c, int W, int HW,
                                                   int nch, int n, const int32_t* __restrict__ n_dev,
                                                   uint64_t* __restrict__ bits, uint32_t* __restrict__ word_pre,
                                                   int32_t* __restrict__ chunk_fg, int32_t* __restrict__ chunk_box,
                                                   float4* __restrict__ list, float4* __restrict__ clist,
                                                   int32_t* __restrict__ n_units) {
    __shared__ __attribute__((aligned(16))) uint8_t s_nib[kChunkPx / 4];
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_units = 0;         // k_vote_plan appends this call's work units
    __shared__ uint64_t s_word[kChunkWords];
    __shared__ int s_wpre[kChunkWords];
    __shared__ int s_box[4];
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
            s_word[threadIdx.x] = word;
            s_wpre[threadIdx.x] = ex;
            if (threadIdx.x == 0) {
                chunk_fg[(size_t)inst * nch + c] = tot;
                s_box[0] = 0x7fffffff; s_box[1] = -1; s_box[2] = 0x7fffffff; s_box[3] = -1;
            }
        }
        __syncthreads();
        int bx0 = 0x7fffffff, bx1 = -1, by0 = 0x7fffffff, by1 = -1;
        const float* v = vertex + (int64_t)inst * vs_n;
        const size_t lbase = (size_t)inst * HW + (size_t)c * kChunkPx;
        // the votes of this lane's four 4-pixel groups: on the x-contiguous, 16-byte aligned layout (the reference's
        // permuted view of two planes) two float4 loads per group, predicated on the group having a foreground pixel and
        // issued together (one memory latency for all eight); any other layout gathers pixel by pixel
        float4 vx[4], vy[4];
        if (VGATHER4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                vx[k] = vy[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (nb[k]) {
                    const int p = c * kChunkPx + (k * 256 + threadIdx.x) * 4;
                    const int y = p / W, x = p - y * W;             // W % 4 == 0: the group stays in one row
                    const float* a = v + (int64_t)y * vs_h + x;
                    vx[k] = *reinterpret_cast<const float4*>(a);
                    vy[k] = *reinterpret_cast<const float4*>(a + vs_c);
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
                    const int p = p0 + q;
                    const int y = p / W, x = p - y * W;
                    const int64_t o = (int64_t)y * vs_h + (int64_t)x * vs_w;
                    e = make_float4((float)x, (float)y, v[o], v[o + vs_c]);
                }
                list[lbase + pos] = e;
                clist[lbase + pos] = pixel_constants(e);
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
        if (threadIdx.x < 4) chunk_box[((size_t)inst * nch + c) * 4 + threadIdx.x] = s_box[threadIdx.x];
        __syncthreads();
    }
}

// ---- rank -> list slot -------------------------------------------------------------
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

// list slot of the e-th foreground pixel (raster order) of an instance
__device__ __forceinline__ int entry_slot(const int32_t* cpre, int nch, int e) {
    const int c = rank_chunk(cpre, nch, e);
    return c * kChunkPx + (e - cpre[c]);
}

// Is the list entry q of a THINNED instance kept (RV/ransac_voting_gpu.py:541-545)?
__device__ __forceinline__ bool entry_kept(float4 q, int W, int HW, int inst, int fg, int max_num, uint64_t seed,
                                           const uint8_t* __restrict__ keep) {
    const int p = (int)q.y * W + (int)q.x;
    return keep ? (keep[(size_t)inst * HW + p] != 0)
                : (fpc_rand_keep(seed, (uint32_t)inst, (uint32_t)p, (uint32_t)fg, (uint32_t)max_num) != 0);
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

// One 1024-thread workgroup per instance.  dynamic LDS: two chunk prefixes [nch + 1] when lds_table.
__global__ __launch_bounds__(1024) void k_vote_plan(const uint8_t* __restrict__ keep, int W, int HW, int nch, int n,
                                                    const int32_t* __restrict__ n_dev, int hn, int hnp,
                                                    const int32_t* __restrict__ idxs, uint64_t seed, int min_num,
                                                    int max_num, float efac, int lds_table,
                                                    int32_t* __restrict__ chunk_fg, const int32_t* __restrict__ chunk_box,
                                                    int32_t* __restrict__ chunk_pre,
                                                    const uint32_t* __restrict__ word_pre,
                                                    const uint64_t* __restrict__ bits, int32_t* __restrict__ chunk_preK,
                                                    uint32_t* __restrict__ word_preK, uint64_t* __restrict__ bitsK,
                                                    const float4* __restrict__ list, int32_t* __restrict__ plan,
                                                    float* __restrict__ hx, float* __restrict__ hy,
                                                    float* __restrict__ hxs, float* __restrict__ hys,
                                                    float* __restrict__ eg, float* __restrict__ hyp,
                                                    int32_t* __restrict__ upper, int32_t* __restrict__ tickets,
                                                    int4* __restrict__ units, int32_t* __restrict__ n_units) {
    extern __shared__ __attribute__((aligned(16))) int s_tab[];      // [2][nch + 1] when lds_table
    __shared__ int s_w[20];
    __shared__ int s_ubase;
    __shared__ int s_box[4];
    int* s_cpre = lds_table ? s_tab : nullptr;
    int* s_cpreK = lds_table ? s_tab + (nch + 1) : nullptr;
    const int n_act = active_instances(n, n_dev);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave, nw = blockDim.x / kWave;
    for (int inst = blockIdx.x; inst < n_act; inst += gridDim.x) {
        int32_t* cfg = chunk_fg + (size_t)inst * nch;
        int32_t* cpre = chunk_pre + (size_t)inst * (nch + 1);
        int32_t* cpreK = chunk_preK + (size_t)inst * (nch + 1);
        const uint32_t* wpre = word_pre + (size_t)inst * nch * kChunkWords;
        const uint64_t* bw = bits + (size_t)inst * nch * kChunkWords;
        uint32_t* wpreK = word_preK + (size_t)inst * nch * kChunkWords;
        uint64_t* bwK = bitsK + (size_t)inst * nch * kChunkWords;
        for (int h = threadIdx.x; h < hnp; h += blockDim.x) upper[(size_t)inst * hnp + h] = 0;
        if (threadIdx.x == 0) { tickets[inst] = 0; s_box[0] = 0x7fffffff; s_box[1] = -1; s_box[2] = 0x7fffffff; s_box[3] = -1; }
        // bounding box of the instance -> the origin the filter's coordinates are measured from, and the largest
        // |x - ox| + |y - oy| of its pixels (both only scale the rounding allowance: any values are sound).  Its loads are
        // issued before the chunk scan so that both memory round trips overlap.
        int b0 = 0x7fffffff, b1 = -1, b2 = 0x7fffffff, b3 = -1;
        for (int c = threadIdx.x; c < nch; c += blockDim.x) {
            const int4 bx = *reinterpret_cast<const int4*>(chunk_box + ((size_t)inst * nch + c) * 4);
            b0 = min(b0, bx.x); b1 = max(b1, bx.y); b2 = min(b2, bx.z); b3 = max(b3, bx.w);
        }
        const int fg = block_scan_chunks(cfg, cpre, s_cpre, nch, s_w);
        {
#pragma unroll
            for (int o = kWave / 2; o > 0; o >>= 1) {
                b0 = min(b0, __shfl_xor(b0, o, kWave)); b1 = max(b1, __shfl_xor(b1, o, kWave));
                b2 = min(b2, __shfl_xor(b2, o, kWave)); b3 = max(b3, __shfl_xor(b3, o, kWave));
            }
            if (lane == 0 && b1 >= 0) { atomicMin(&s_box[0], b0); atomicMax(&s_box[1], b1); atomicMin(&s_box[2], b2); atomicMax(&s_box[3], b3); }
            __syncthreads();
        }
        int ox = 0, oy = 0, rad = W + HW / W;
        if (s_box[1] >= 0) {
            ox = (s_box[0] + s_box[1]) / 2; oy = (s_box[2] + s_box[3]) / 2;
            rad = max(s_box[1] - ox, ox - s_box[0]) + max(s_box[3] - oy, oy - s_box[2]);
        }
        const float fox = (float)ox, foy = (float)oy, frad = (float)rad;
        int tn = fg;
        const bool thin = fg > max_num;
        if (thin) {
            // RV/ransac_voting_gpu.py:541-545: keep each foreground pixel with probability max_num / fg (injected
            // selection, or the counter-based stream of include/fpc_rng.h).  One wave per chunk, one lane per word.
            // The kept image only serves the pair sampling below: the lists keep every foreground pixel and the
            // count / exact kernels re-derive each entry's decision.
            for (int c = wv; c < nch; c += nw) {
                const size_t wi = (size_t)c * kChunkWords + lane;
                // lane = word for the loads and the prefix; the keep decisions of one word are taken by the 64 lanes at
                // once (lane = pixel), word by word over the chunk's non-empty words: a dense chunk costs 64 hashes per
                // lane, not 64 x 64 (one lane walking its own word's bits was 35 us for a 30 000-pixel instance)
                const uint64_t word = bw[wi];
                uint64_t kept = 0;
                unsigned long long todo = __builtin_amdgcn_ballot_w64(word != 0);
                while (todo) {                                               // uniform
                    const int w = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)word, w);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(word >> 32), w);
                    const uint64_t ww = ((uint64_t)hi << 32) | lo;
                    bool k = false;
                    if ((ww >> lane) & 1ull) {
                        const int p = (int)(((size_t)c * kChunkWords + w) * 64) + lane;
                        k = keep ? (keep[(size_t)inst * HW + p] != 0)
                                 : (fpc_rand_keep(seed, (uint32_t)inst, (uint32_t)p, (uint32_t)fg, (uint32_t)max_num) != 0);
                    }
                    const uint64_t kw = __builtin_amdgcn_ballot_w64(k);
                    if (lane == w) kept = kw;
                }
                int tot;
                const int ex = wave_excl_scan(__popcll(kept), tot);
                bwK[wi] = kept;
                wpreK[wi] = (uint32_t)ex;
                if (lane == 0) cfg[c] = tot;
            }
            __syncthreads();
            tn = block_scan_chunks(cfg, cpreK, s_cpreK, nch, s_w);
        }
        if (fg < min_num) tn = 0;      // :536-539
        if (threadIdx.x == 0) {
            plan[inst * kPlanI + 0] = fg;
            plan[inst * kPlanI + 1] = tn;
            plan[inst * kPlanI + 2] = thin ? 1 : 0;
            plan[inst * kPlanI + 3] = ox;
            plan[inst * kPlanI + 4] = oy;
            plan[inst * kPlanI + 5] = rad;
            // the work units of k_vote_count: one per block of 512 list entries (no unit for an instance that does not vote)
            const int nb_i = tn > 0 ? (fg + kBlockPx - 1) / kBlockPx : 0;
            s_ubase = nb_i ? atomicAdd(n_units, nb_i) : 0;
        }
        __syncthreads();               // the tables of this instance are complete (same CU: visible)
        for (int b = threadIdx.x; b * kBlockPx < (tn > 0 ? fg : 0); b += blockDim.x)
            units[s_ubase + b] = make_int4(inst, b, fg | (thin ? (int)0x80000000 : 0), (ox & 0xffff) | (oy << 16));

        const int32_t* tab = lds_table ? s_cpre : cpre;
        const int32_t* tabK = lds_table ? s_cpreK : cpreK;
        const float4* E = list + (size_t)inst * HW;
        // list slot of the t-th pixel the pair sampling may draw (the t-th KEPT one when thinned)
        auto sample_slot = [&](int t) -> int {
            if (!thin) return entry_slot(tab, nch, t);
            const int c = rank_chunk(tabK, nch, t);
            int r = t - tabK[c];
            const uint32_t* wp = wpreK + (size_t)c * kChunkWords;
            int wl = 0, wh2 = kChunkWords;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const int mid = (wl + wh2) >> 1;
                if ((int)wp[mid] <= r) wl = mid; else wh2 = mid;
            }
            r -= (int)wp[wl];
            const int w = c * kChunkWords + wl;
            const int bit = select64(bwK[w], r);
            return c * kChunkPx + (int)wpre[w] + __popcll(bw[w] & ((1ull << bit) - 1ull));
        };
        for (int h0 = 0; h0 < hnp; h0 += blockDim.x) {       // uniform trip count; a wave holds one group of 64
            const int hi = h0 + threadIdx.x;
            float x = 0.0f, y = 0.0f, e = 0.0f;
            bool wild = false;
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
                        const int s0 = sample_slot(t0), s1 = sample_slot(t1);
                        intersect(E[s0], E[s1], x, y);
                    }
                }
                hyp[((size_t)inst * hn + hi) * 2] = x;
                hyp[((size_t)inst * hn + hi) * 2 + 1] = y;
                const float s = fabsf(x) + fabsf(y);
                wild = !(s <= 1e18f);                          // inf / NaN / huge: outside the filter's domain
                if (!wild) e = efac * (fabsf(x - fox) + fabsf(y - foy) + frad);
            }
            if (hi < hnp) {
                hx[(size_t)inst * hnp + hi] = x;
                hy[(size_t)inst * hnp + hi] = y;
                hxs[(size_t)inst * hnp + hi] = x - fox;
                hys[(size_t)inst * hnp + hi] = y - foy;
                float m = e;
#pragma unroll
                for (int o = kWave / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, kWave));
                if (__builtin_amdgcn_ballot_w64(wild)) m = -1.0f;     // the whole group takes the reference's arithmetic
                if (lane == 0) eg[(size_t)inst * (hnp / kWave) + hi / kWave] = m;
            }
        }
        __syncthreads();               // s_w / s_tab are reused by the next instance
    }
}

// ---- pair classification shared by k_vote_count<kModeFiltered> and k_vote_exact ----------------------------------
// cst = {a_s = ey, b_s = -ex, c_s, ct} (c_s NaN: the pixel never votes).  Returns, per lane, whether the reference
// accepts (pixel, hypothesis): two FMA pairs and two compares decide unless the pair lies between the cones; those
// lanes (wave-uniform branch, rare) run the reference's own arithmetic on the raw pixel q = {x, y, dx, dy}.
struct Cones { float kappa1, kappa2; };

// (gxs, gys) = the point minus the instance's origin, cst re-centred on it; (gx, gy) = the point itself.
__device__ __forceinline__ bool classify_pair(const float4 cst, const float4 q, const Cones k, float gxs, float gys,
                                              float gx, float gy, float E, bool wild, bool valid, float thresh) {
    const float ss = fabsf(__builtin_fmaf(cst.x, gxs, __builtin_fmaf(cst.y, gys, cst.z)));
    const float tt = __builtin_fmaf(-cst.y, gxs, __builtin_fmaf(cst.x, gys, cst.w));
    bool sure = ss <= __builtin_fmaf(k.kappa2, tt, -E);
    bool band = !sure && (ss <= __builtin_fmaf(k.kappa1, tt, E));
    if (wild) { sure = false; band = valid; }                   // wave-uniform: outside the filter's domain
    if (__builtin_amdgcn_ballot_w64(band)) {
        if (band) {
            const float n1 = sqrtf(q.z * q.z + q.w * q.w);
            sure = pair_is_inlier(q.x, q.y, q.z, q.w, n1, gx, gy, thresh);
        }
    }
    return sure && valid;
}

// The pairs between the cones take the reference's own arithmetic.  The hot loop of k_vote_count only QUEUES them
// (entry = hypothesis g | tile << 6 | lane << 7, in the wave's LDS queue); band_flush evaluates up to 64 queued pairs at
// once, one per lane — the pixel comes from its owner lane and the point from lane g by ds_bpermute — and adds the
// inliers to the group's packed LDS counts.  Out of line: the hot loop stays a run of FMAs, compares and scalar counts,
// with no scalar-memory wait and no per-event sqrt / division.
constexpr int kBandQ = 256;          // queue entries per wave: one step of the hot loop adds at most 4 x 64

__device__ __forceinline__ float lane_fetch(float v, int src_lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}

__device__ __attribute__((noinline)) void band_flush(int n, const int* __restrict__ queue, float4 q0, float4 q1, float hxv,
                                                     float hyv, float thresh, int* __restrict__ s_cnt_group) {
    const int lane = threadIdx.x & (kWave - 1);
    for (int base = 0; base < n; base += kWave) {                           // uniform
        const bool on = base + lane < n;
        const int e = on ? queue[base + lane] : 0;
        const int g = e & 63, src = e >> 7;
        const bool t1 = (e >> 6) & 1;
        const float x0 = lane_fetch(q0.x, src), y0 = lane_fetch(q0.y, src), z0 = lane_fetch(q0.z, src), w0 = lane_fetch(q0.w, src);
        const float x1 = lane_fetch(q1.x, src), y1 = lane_fetch(q1.y, src), z1 = lane_fetch(q1.z, src), w1 = lane_fetch(q1.w, src);
        const float gx = lane_fetch(hxv, g), gy = lane_fetch(hyv, g);
        const float qx = t1 ? x1 : x0, qy = t1 ? y1 : y0, qz = t1 ? z1 : z0, qw = t1 ? w1 : w0;
        if (on && pair_is_inlier(qx, qy, qz, qw, sqrtf(qz * qz + qw * qw), gx, gy, thresh))
            atomicAdd(&s_cnt_group[g >> 1], 1 << ((g & 1) * 16));
    }
}

// the lanes of `m` append (their code | lane << 7) to the wave's queue
__device__ __forceinline__ void band_push(unsigned long long m, int code, int lane, int* __restrict__ bq, int& qn) {
    if ((m >> lane) & 1ull)
        bq[qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] = code | (lane << 7);
    qn += __popcll(m);
}

// In-register 64 x 64 bit-matrix transpose across the wave (row = lane, column = bit of hi:lo): six block-swap stages.
__device__ __forceinline__ void transpose64(unsigned& lo, unsigned& hi, int lane) {
    {   // 32 x 32 blocks: lanes < 32 give their high word and take the partner's low word
        const unsigned recv = (unsigned)__shfl_xor((int)((lane & 32) ? lo : hi), 32, kWave);
        if (lane & 32) lo = recv; else hi = recv;
    }
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int m = 16 >> st;
        const unsigned cm = m == 16 ? 0xFFFF0000u : m == 8 ? 0xFF00FF00u : m == 4 ? 0xF0F0F0F0u : m == 2 ? 0xCCCCCCCCu : 0xAAAAAAAAu;
        const bool up = (lane & m) != 0;                     // keeps its high columns, takes the partner's into the low ones
        const unsigned keep = up ? cm : ~cm;
        const unsigned ylo = (unsigned)__shfl_xor((int)lo, m, kWave), yhi = (unsigned)__shfl_xor((int)hi, m, kWave);
        lo = (lo & keep) | ((up ? (ylo >> m) : (ylo << m)) & ~keep);
        hi = (hi & keep) | ((up ? (yhi >> m) : (yhi << m)) & ~keep);
    }
}

struct TilePair {                    // one lane's two list entries: filter constants, |vote|, raw pixel
    float a_s[2], b_s[2], c_s[2], a_t[2], b_t[2], c_t[2], n1[2];
    float4 q[2];
};

// ---- k_vote_count ------------------------------------------------------------------
// EXACT inlier count of every hypothesis.  Task t -> (instance, block of 512 list entries, hypothesis slice).  The
// slices of one block differ by 8 in t, i.e. they run on one XCD under round-robin dispatch (speed only).
// grid-stride; 256 threads; a wave owns two 64-entry tiles; lanes hold six constants per entry, the hypotheses arrive
// in SGPRs (scalar loads).  Per (entry, hypothesis): 5 FMA + 2 compares decide "surely an inlier" / "surely not";
// the pairs between the cones (wave-uniform branch, a few per 10^4) run the reference's own arithmetic.
// dynamic LDS: [gps * 32] packed counts of the slice (two hypotheses per word), then the chunk prefix [nch + 1] when lds_table.
enum { kModeCones = 0, kModeReference = 1 };

template <int MODE, int WAVES /* waves per SIMD the register allocation aims at */>
__global__ __launch_bounds__(256, WAVES) void k_vote_count(int W, int HW, int nch, const int4* __restrict__ units,
                                                    const int32_t* __restrict__ n_units, int hn, int hnp,
                                                    int task_target /* tasks the launch wants: slices are cut to reach it */,
                                                    int s_fixed /* > 0: that many slices (tuning aid) */, float kappa1,
                                                    float kappa2, float thresh, int max_num, uint64_t seed,
                                                    const uint8_t* __restrict__ keep, int lds_table,
                                                    const int32_t* __restrict__ chunk_pre,
                                                    const int32_t* __restrict__ plan, const float* __restrict__ hx,
                                                    const float* __restrict__ hy, const float* __restrict__ hxs,
                                                    const float* __restrict__ hys, const float* __restrict__ eg,
                                                    int32_t* __restrict__ counts, const float4* __restrict__ list,
                                                    const float4* __restrict__ clist, int dbg) {
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];
    const int ngroups = hnp / kWave;
    int* s_cnt = s_dyn;                              // [ngroups * 32] (a slice uses its first gps * 32)
    int* s_cpre = s_dyn + ngroups * (kWave / 2);     // [nch + 1]
    __shared__ int s_bandq[4][kBandQ];               // per wave: queued band pairs of the current group
    // The units (blocks that have entries) are known on the device only; each is cut into S slices of gps groups of 64
    // hypotheses so that units x S ~ the task count the launch was sized for: one round of equal tasks over the chip.
    const int nu = *n_units;
    const int S0 = s_fixed > 0 ? s_fixed : max(1, task_target / max(nu, 1));
    const int gps = (ngroups + min(S0, ngroups) - 1) / min(S0, ngroups);
    const int S = (ngroups + gps - 1) / gps;
    const long long total = ((long long)nu + 7) / 8 * 8 * S;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    // sure  <=>  |s| <= kappa2 t - E  =  r (kappa1 t + E) - (1 + r) E   with r = kappa2 / kappa1: one FMA on the other bound
    const float ratio = kappa1 > 0.0f ? kappa2 / kappa1 : 0.0f;
    int cached_inst = -1;
    for (long long t = blockIdx.x; t < total; t += gridDim.x) {
        const long long grp = t / (8 * S);
        const int rem = (int)(t - grp * (8 * S));
        const int s = rem >> 3;
        const int u = (int)(grp * 8 + (rem & 7));
        if (u >= nu) continue;
        const int4 ub = units[u];
        const int inst = ub.x, b = ub.y;
        const int fg = ub.z & 0x7fffffff;
        const bool thin = ub.z < 0;
        const float fox = (float)(ub.w & 0xffff), foy = (float)(ub.w >> 16);
        const int nent = fg;                                        // list entries of the instance (all foreground pixels)
        const int g_lo = s * gps, g_hi = min(ngroups, g_lo + gps);
        if (g_lo >= g_hi) continue;                                 // uniform
        if (lds_table && cached_inst != inst) {                     // uniform
            __syncthreads();
            for (int i = threadIdx.x; i <= nch; i += blockDim.x) s_cpre[i] = chunk_pre[(size_t)inst * (nch + 1) + i];
            cached_inst = inst;
            __syncthreads();
        }
        const int32_t* tab = lds_table ? s_cpre : chunk_pre + (size_t)inst * (nch + 1);
        {
            TilePair tp;                                            // this lane's two entries
            float c_t0[2];
            bool valid[2];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const int e = b * kBlockPx + (wv * 2 + tl) * kWave + lane;
                valid[tl] = e < nent;
                // a slot that never counts: |s| = +inf, so both margins are -inf (sign set, never NaN)
                const float4 never = make_float4(0.f, 0.f, __builtin_inff(), 0.f);
                float4 cst = never;
                tp.q[tl] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (valid[tl]) {
                    const size_t slot = (size_t)inst * HW + entry_slot(tab, nch, e);
                    tp.q[tl] = list[slot];
                    if (MODE == kModeCones) cst = recentre_constants(clist[slot], tp.q[tl], fox, foy);
                    if (cst.z != cst.z) cst = never;                 // a vote the reference skips (pixel_constants)
                    if (thin && !entry_kept(tp.q[tl], W, HW, inst, fg, max_num, seed, keep)) {
                        valid[tl] = false;
                        cst = never;
                    }
                }
                // folded constants:  s = a_s gx + b_s gy + c_s ;  kappa1 t + E_g = a_t gx + b_t gy + (c_t0 + E_g)
                tp.a_s[tl] = cst.x; tp.b_s[tl] = cst.y; tp.c_s[tl] = cst.z;
                tp.a_t[tl] = kappa1 * -cst.y; tp.b_t[tl] = kappa1 * cst.x; c_t0[tl] = kappa1 * cst.w;
                tp.n1[tl] = sqrtf(tp.q[tl].z * tp.q[tl].z + tp.q[tl].w * tp.q[tl].w);
            }
            for (int i = threadIdx.x; i < (g_hi - g_lo) * (kWave / 2); i += blockDim.x) s_cnt[i] = 0;
            __syncthreads();

            for (int G = g_lo; G < g_hi; ++G) {
                if (dbg == 1) break;                 // tuning aid (FPC_COUNT_DBG=1): prologue / epilogue only, results wrong
                // HX / HY: the points themselves (the reference's arithmetic); HXS / HYS: minus the origin (the cones)
                const float* HX = static_cast<const float*>(__builtin_assume_aligned(hx + (size_t)inst * hnp + (size_t)G * kWave, 256));
                const float* HY = static_cast<const float*>(__builtin_assume_aligned(hy + (size_t)inst * hnp + (size_t)G * kWave, 256));
                const float* HXS = static_cast<const float*>(__builtin_assume_aligned(hxs + (size_t)inst * hnp + (size_t)G * kWave, 256));
                const float* HYS = static_cast<const float*>(__builtin_assume_aligned(hys + (size_t)inst * hnp + (size_t)G * kWave, 256));
                // E_g < 0 marks a group with a hypothesis outside the filter's domain: its pairs all take the reference's
                // arithmetic (the cones are switched off by NaN bounds)
                const float eg_g = eg[(size_t)inst * ngroups + G];
                int cntv = 0;
                if (MODE != kModeCones || !(eg_g >= 0.0f)) {                        // uniform: every pair by the reference's arithmetic
                    for (int g = 0; g < kWave; ++g) {
                        const float gx = HX[g], gy = HY[g];
                        int c = 0;
#pragma unroll
                        for (int tl = 0; tl < 2; ++tl)
                            c += __popcll(__builtin_amdgcn_ballot_w64(
                                valid[tl] && pair_is_inlier(tp.q[tl].x, tp.q[tl].y, tp.q[tl].z, tp.q[tl].w, tp.n1[tl], gx, gy, thresh)));
                        if (lane == (g >> 1)) cntv |= c << ((g & 1) * 16);
                    }
                    if (lane < kWave / 2) atomicAdd(&s_cnt[(G - g_lo) * (kWave / 2) + lane], cntv);
                    continue;
                }
                tp.c_t[0] = c_t0[0] + eg_g; tp.c_t[1] = c_t0[1] + eg_g;
                const float e2 = (1.0f + ratio) * eg_g;
                // lane g holds point g of the group: minus the origin for the cones (v_readlane -> SGPR operands of the
                // FMAs: no scalar-memory wait inside the loop), as it is for band_flush
                const float hxv = HX[lane], hyv = HY[lane];
                int* bq = s_bandq[wv];
                int* cnt_g = s_cnt + (G - g_lo) * (kWave / 2);
                // One (hypothesis, two 64-entry tiles) step is plain VALU work only: per tile the margins to the outer cone
                // (d = u1 - |s|) and to the inner one (d2 = r u1 - e2 - |s|), whose SIGN bits are shifted into per-lane
                // bit rows (v_alignbit: row = (row << 1) | sign).  x - y >= 0 exactly when y <= x, so the signs are the
                // reference-safe compares of the two-cone filter; no v_cmp -> SGPR -> s_bcnt chain, no branch.  Ten f32
                // lane-operations per (entry, point): measured, the loop runs at the SIMD's plain-f32 rate (4 cycles per
                // wave64 instruction; v_pk_fma_f32 on the two tiles at once costs twice that, i.e. gains nothing).
                unsigned rowO[2][2] = {{0u, 0u}, {0u, 0u}}, rowS[2][2] = {{0u, 0u}, {0u, 0u}};
#define FPC_VOTE_HYP(g)                                                                                                \
                {                                                                                                      \
                    const float gxs = HXS[(g)], gys = HYS[(g)];      /* scalar loads, batched by the compiler */        \
                    _Pragma("unroll") for (int tl = 0; tl < 2; ++tl) {                                                 \
                        const float ss = fabsf(__builtin_fmaf(tp.a_s[tl], gxs, __builtin_fmaf(tp.b_s[tl], gys, tp.c_s[tl]))); \
                        const float u1 = __builtin_fmaf(tp.a_t[tl], gxs, __builtin_fmaf(tp.b_t[tl], gys, tp.c_t[tl])); \
                        const float d = u1 - ss, d2 = __builtin_fmaf(ratio, u1, -e2) - ss;                             \
                        rowO[tl][(g) >> 5] = __builtin_amdgcn_alignbit(rowO[tl][(g) >> 5], __float_as_uint(d), 31);    \
                        rowS[tl][(g) >> 5] = __builtin_amdgcn_alignbit(rowS[tl][(g) >> 5], __float_as_uint(d2), 31);   \
                    }                                                                                                  \
                }
#define FPC_VOTE_4(g) FPC_VOTE_HYP(g) FPC_VOTE_HYP((g) + 1) FPC_VOTE_HYP((g) + 2) FPC_VOTE_HYP((g) + 3)
#define FPC_VOTE_16(g) FPC_VOTE_4(g) FPC_VOTE_4((g) + 4) FPC_VOTE_4((g) + 8) FPC_VOTE_4((g) + 12)
                FPC_VOTE_16(0) FPC_VOTE_16(16) FPC_VOTE_16(32) FPC_VOTE_16(48)
#undef FPC_VOTE_16
#undef FPC_VOTE_4
#undef FPC_VOTE_HYP
                // bit g of a row <-> point g: sure = inside the inner cone, band = between the cones (rare)
                unsigned long long sure[2], band[2];
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    const unsigned o0 = ~__builtin_bitreverse32(rowO[tl][0]), o1 = ~__builtin_bitreverse32(rowO[tl][1]);
                    const unsigned s0 = ~__builtin_bitreverse32(rowS[tl][0]), s1 = ~__builtin_bitreverse32(rowS[tl][1]);
                    sure[tl] = ((unsigned long long)s1 << 32) | s0;
                    band[tl] = ((unsigned long long)(o1 & ~s1) << 32) | (o0 & ~s0);
                }
                // the pairs between the cones are queued (one entry per pair), then evaluated 64 at a time by band_flush
                int qn = 0;
                while (__builtin_amdgcn_ballot_w64((band[0] | band[1]) != 0ull)) {                   // uniform; usually not entered
                    const bool has = (band[0] | band[1]) != 0ull;
                    int code = 0;
                    if (band[0]) { code = __ffsll((long long)band[0]) - 1; band[0] &= band[0] - 1; }
                    else if (band[1]) { code = (__ffsll((long long)band[1]) - 1) | 64; band[1] &= band[1] - 1; }
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(has);
                    if (qn + __popcll(m) > kBandQ) {
                        band_flush(qn, bq, tp.q[0], tp.q[1], hxv, hyv, thresh, cnt_g);
                        qn = 0;
                    }
                    band_push(m, code, lane, bq, qn);
                }
                if (qn) band_flush(qn, bq, tp.q[0], tp.q[1], hxv, hyv, thresh, cnt_g);
                // 64 x 64 bit transposes across the wave: lane g then holds the sure bits of point g over the tile's entries
                int cnt = 0;
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    unsigned lo = (unsigned)sure[tl], hi = (unsigned)(sure[tl] >> 32);
                    transpose64(lo, hi, lane);
                    cnt += __popc(lo) + __popc(hi);
                }
                atomicAdd(&cnt_g[lane >> 1], cnt << ((lane & 1) * 16));                              // fields <= 512: no carry
            }
            __syncthreads();
            // one integer atomic per (block, hypothesis) with any count: order-independent result
            for (int i = threadIdx.x; i < (g_hi - g_lo) * (kWave / 2); i += blockDim.x) {
                const int h = g_lo * kWave + 2 * i, pk = s_cnt[i];
                const int lo = pk & 0xffff, hi = (int)((unsigned)pk >> 16);
                if (lo && h < hn) atomicAdd(&counts[(size_t)inst * hnp + h], lo);
                if (hi && h + 1 < hn) atomicAdd(&counts[(size_t)inst * hnp + h + 1], hi);
            }
            __syncthreads();
        }
    }
}

// ---- k_vote_final ------------------------------------------------------------------
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

constexpr int kFinWaves = 4;         // 256-thread workgroups

typedef unsigned long long __attribute__((address_space(1))) gu64;

// Winner (largest count, lowest index: torch.max, RV/ransac_voting_gpu.py:567), its inliers voted again (:583-589), the
// fp64 normal equations and the 2x2 solve (:592-599).  Task = one work unit of k_vote_plan's list = (instance, block b0 of
// 512 list entries); the unit of an instance whose arrival ticket comes last combines the instance's records
// (cdna_hip_programming.md Guideline 16, counter form: records stored write-through (sc1), the storing wave drained,
// one agent-scope add per workgroup; the last arriver reads them back with sc1 loads, in task order: bit-reproducible).
// dynamic LDS: the chunk prefix [nch + 1] when lds_table.
template <int MODE>
__global__ __launch_bounds__(256) void k_vote_final(int W, int HW, int nch, int n, const int32_t* __restrict__ n_dev,
                                                    const int4* __restrict__ units, const int32_t* __restrict__ n_units,
                                                    int hn, int hnp, int nbx, float thresh, float kappa1,
                                                    float kappa2, float efac_ref, int max_num, uint64_t seed,
                                                    const uint8_t* __restrict__ keep, int lds_table,
                                                    const int32_t* __restrict__ chunk_pre,
                                                    const int32_t* __restrict__ plan, const float* __restrict__ hyp,
                                                    const int32_t* __restrict__ counts, const float4* __restrict__ list,
                                                    const float4* __restrict__ clist, int32_t* __restrict__ tickets,
                                                    double* __restrict__ partial, float* __restrict__ out_xy,
                                                    int32_t* __restrict__ out_tn, int32_t* __restrict__ out_win_idx,
                                                    int32_t* __restrict__ out_win_count, int32_t* __restrict__ out_inl,
                                                    double* __restrict__ out_refine) {
    extern __shared__ __attribute__((aligned(16))) int s_cpre[];      // [nch + 1]
    __shared__ int s_red[2 * kFinWaves];
    __shared__ int s_last;
    __shared__ double s_part[kFinWaves][kRec];
    const int n_act = active_instances(n, n_dev);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const Cones cones{kappa1, kappa2};
    // instances that do not vote (fewer than min_num pixels) have no unit: zeros (RV/ransac_voting_gpu.py:536-539)
    for (int inst = blockIdx.x * blockDim.x + threadIdx.x; inst < n_act; inst += gridDim.x * blockDim.x)
        if (plan[inst * kPlanI + 1] == 0) {
            out_xy[inst * 2] = 0.0f; out_xy[inst * 2 + 1] = 0.0f;
            if (out_tn) out_tn[inst] = 0;
            if (out_win_idx) out_win_idx[inst] = -1;
            if (out_win_count) out_win_count[inst] = 0;
            if (out_inl) out_inl[inst] = 0;
            if (out_refine)
                for (int i = 0; i < 8; ++i) out_refine[(size_t)inst * 8 + i] = 0.0;
        }
    const int nu = *n_units;
    for (int t = blockIdx.x; t < nu; t += gridDim.x) {
        const int4 ub = units[t];
        const int inst = ub.x, b0 = ub.y;
        const int fg = plan[inst * kPlanI + 0], tn = plan[inst * kPlanI + 1];
        const bool thin = plan[inst * kPlanI + 2] != 0;
        const int nent = fg;
        const int nb_i = (nent + kBlockPx - 1) / kBlockPx;              // units of this instance = arrivals to wait for
        const float fox = (float)plan[inst * kPlanI + 3], foy = (float)plan[inst * kPlanI + 4];
        const float frad = (float)plan[inst * kPlanI + 5];
        // winner: every task of the instance finds the same one
        int wc = -1, wi = 0x7fffffff;
        for (int h = threadIdx.x; h < hn; h += blockDim.x) {
            const int c = counts[(size_t)inst * hnp + h];
            if (c > wc) { wc = c; wi = h; }                            // ascending h: first maximum kept
        }
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            const int oc = __shfl_xor(wc, o, kWave), oi = __shfl_xor(wi, o, kWave);
            if (oc > wc || (oc == wc && oi < wi)) { wc = oc; wi = oi; }
        }
        __syncthreads();                                               // LDS of the previous task is free
        if (lane == 0) { s_red[wv] = wc; s_red[kFinWaves + wv] = wi; }
        if (lds_table)
            for (int i = threadIdx.x; i <= nch; i += blockDim.x) s_cpre[i] = chunk_pre[(size_t)inst * (nch + 1) + i];
        __syncthreads();
        wc = s_red[0]; wi = s_red[kFinWaves];
#pragma unroll
        for (int i = 1; i < kFinWaves; ++i) {
            const int oc = s_red[i], oi = s_red[kFinWaves + i];
            if (oc > wc || (oc == wc && oi < wi)) { wc = oc; wi = oi; }
        }
        // no hypothesis with an inlier: all_win_pts stays (0,0) (:571-574) and the refinement votes for (0,0)
        const float* hp = hyp + (size_t)inst * hn * 2;
        float wx = 0.0f, wy = 0.0f;
        if (wc > 0) { wx = hp[2 * wi]; wy = hp[2 * wi + 1]; } else { wi = -1; wc = 0; }
        const float wxs = wx - fox, wys = wy - foy;
        const bool wwild = !(fabsf(wx) + fabsf(wy) <= 1e18f);
        const float wE = efac_ref * (fabsf(wxs) + fabsf(wys) + frad);
        const int32_t* tab = lds_table ? s_cpre : chunk_pre + (size_t)inst * (nch + 1);

        double v[kRec] = {0, 0, 0, 0, 0, 0};                            // inliers, a00, a01, a11, b0, b1
        {
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const int e = b0 * kBlockPx + (wv * 2 + tl) * kWave + lane;
                bool valid = e < nent;
                const size_t slot = (size_t)inst * HW + (valid ? entry_slot(tab, nch, e) : 0);
                const float4 q = list[slot];
                if (valid && thin) valid = entry_kept(q, W, HW, inst, fg, max_num, seed, keep);
                bool in;
                if (MODE == kModeCones)
                    in = classify_pair(recentre_constants(clist[slot], q, fox, foy), q, cones, wxs, wys, wx, wy, wE, wwild,
                                       valid, thresh);
                else
                    in = valid && pair_is_inlier(q.x, q.y, q.z, q.w, sqrtf(q.z * q.z + q.w * q.w), wx, wy, thresh);
                if (in) {
                    const double nx = (double)q.w, ny = -(double)q.z;  // normal = (dy, -dx) :584-586
                    const double bb = nx * (double)q.x + ny * (double)q.y;
                    v[0] += 1.0; v[1] += nx * nx; v[2] += nx * ny; v[3] += ny * ny; v[4] += nx * bb; v[5] += ny * bb;
                }
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
            __hip_atomic_store((gu64*)(partial + ((size_t)inst * nbx + b0) * kRec + threadIdx.x),
                               __builtin_bit_cast(unsigned long long, r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the storing wave drains its sc1 stores
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tk = __hip_atomic_fetch_add(&tickets[inst], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (tk == nb_i - 1);
        }
        __syncthreads();
        if (!s_last) continue;                                         // uniform

        // last arriver of the instance: the records in task order (independent sc1 loads, four in flight per lane)
        if (wv == 0) {
            const int nrec = nb_i;
            double tot[kRec] = {0, 0, 0, 0, 0, 0};
            // lane = (record slot r8 = lane / 8, value a = lane % 8): eight records per sweep, then a fixed-order lane tree
            const int a = lane & 7, r8 = lane >> 3;
            double acc = 0.0;
            for (int b = r8; b < nrec; b += 32) {
                unsigned long long x[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    x[i] = (a < kRec && b + 8 * i < nrec)
                               ? __hip_atomic_load((gu64*)(partial + ((size_t)inst * nbx + b + 8 * i) * kRec + a), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT)
                               : 0ull;
#pragma unroll
                for (int i = 0; i < 4; ++i) acc += __builtin_bit_cast(double, x[i]);
            }
            acc += __shfl_xor(acc, 8, kWave); acc += __shfl_xor(acc, 16, kWave); acc += __shfl_xor(acc, 32, kWave);
#pragma unroll
            for (int i = 0; i < kRec; ++i) tot[i] = __shfl(acc, i, kWave);
            if (lane == 0) {
                double x0, x1;
                solve2_sym(tot[1], tot[2], tot[3], tot[4], tot[5], x0, x1);
                out_xy[inst * 2] = (float)x0;
                out_xy[inst * 2 + 1] = (float)x1;
                if (out_tn) out_tn[inst] = tn;
                if (out_win_idx) out_win_idx[inst] = wi;
                if (out_win_count) out_win_count[inst] = wc;
                if (out_inl) out_inl[inst] = (int)tot[0];
                if (out_refine) {      // what the refinement's backward needs (fpc_vote_refine_backward)
                    double* r = out_refine + (size_t)inst * 8;
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

    // 1. mask planes -> bit image + per-chunk compacted pixel lists (the only pass over the masks and the vote planes)
    const bool vec4 = (HW % 4 == 0) && (((uintptr_t)mask & 15) == 0);
    const bool vg4 = vec4 && W % 4 == 0 && vs_w == 1 && vs_h % 4 == 0 && vs_n % 4 == 0 && vs_c % 4 == 0 &&
                     (((uintptr_t)vertex & 15) == 0);
    const int scan_grid = (int)std::min<long long>((long long)n * w.nch, 8192);
#define FPC_LAUNCH_SCAN(A, B)                                                                                            \
    hipLaunchKernelGGL((k_vote_scan<A, B>), dim3(scan_grid), dim3(256), 0, s, mask, vertex, vs_n, vs_h, vs_w, vs_c, W, HW,  \
                       w.nch, n, n_dev, w.bits, w.word_pre, w.chunk_fg, w.chunk_box, w.list, w.clist, w.n_units)
    if (vg4) FPC_LAUNCH_SCAN(true, true); else if (vec4) FPC_LAUNCH_SCAN(true, false); else FPC_LAUNCH_SCAN(false, false);
#undef FPC_LAUNCH_SCAN

    // the cones need th' = th - 1e-6 > 0; otherwise every pair takes the reference's arithmetic
    const bool fast = inlier_thresh > 2e-6f && inlier_thresh < 3.0e38f;
    float kappa1 = 0.0f, kappa2 = 0.0f;
    if (fast) {
        const double th1 = (double)inlier_thresh - 1e-6, th2 = (double)inlier_thresh + 1e-6;
        const double k1 = 1.0 - th1 * th1, k2 = 1.0 - th2 * th2;
        kappa1 = (float)((k1 > 0.0 ? sqrt(k1) : 0.0) / th1) * (1.0f + 1e-6f);                       // wider
        kappa2 = (th2 < 1.0 && k2 > 0.0) ? (float)(sqrt(k2) / th2) * (1.0f - 1e-6f) : 0.0f;        // narrower (0: no "sure")
    }
    // rounding allowance of the cones per unit of magnitude M = |gx - ox| + |gy - oy| + radius (header comment)
    const float efac = 2e-6f * (1.0f + kappa1);
    const int lds_table = w.nch + 1 <= 2048 ? 1 : 0;              // chunk prefix of an instance in LDS (8 KB; two in the plan)
    const size_t table_lds = lds_table ? (size_t)(w.nch + 1) * sizeof(int) : 0;

    // 2. per instance: prefix, thinning, origin, hypotheses, zeroed count row and arrival ticket
    static const int plan_threads = getenv("FPC_PLAN_THREADS") ? atoi(getenv("FPC_PLAN_THREADS")) : 1024;      // tuning aid
    hipLaunchKernelGGL(k_vote_plan, dim3(std::min(n, 2048)), dim3(plan_threads), 2 * table_lds, s, keep, W, HW, w.nch, n, n_dev, hn, w.hnp,
                       idxs, seed, min_num, max_num, efac, lds_table, w.chunk_fg, w.chunk_box, w.chunk_pre, w.word_pre, w.bits,
                       w.chunk_preK, w.word_preK, w.bitsK, w.list, w.plan, w.hx, w.hy, w.hxs, w.hys, w.eg, w.hyp, w.upper,
                       w.tickets, w.units, w.n_units);

    // 3. exact inlier counts of every hypothesis.  One resident round of workgroups (five per CU); the kernel reads how many
    // blocks have entries and cuts each into hypothesis slices so that the tasks fill that round evenly.
    const int nb_launch = std::min(w.nbx, cdiv(std::min(HW, max_num), kBlockPx) + 1);   // blocks per instance in the final's task grid
    const int ngroups = w.hnp / kWave;
    static const int count_waves = getenv("FPC_COUNT_WAVES") ? atoi(getenv("FPC_COUNT_WAVES")) : 5;      // tuning aids
    static const int count_slices = getenv("FPC_COUNT_SLICES") ? atoi(getenv("FPC_COUNT_SLICES")) : 0;
    static const int count_rounds = getenv("FPC_COUNT_ROUNDS") ? atoi(getenv("FPC_COUNT_ROUNDS")) : 1;
    static const int count_dbg = getenv("FPC_COUNT_DBG") ? atoi(getenv("FPC_COUNT_DBG")) : 0;
    const long long cap_tasks = ((long long)n * w.nbx + 7) / 8 * 8 * ngroups;
    const int resident = 256 * std::min(std::max(count_waves, 4), 6);                  // workgroups the chip holds at once
    const int count_grid = (int)std::min<long long>(cap_tasks, resident);                 // a multiple of 8 either way
    const int task_target = count_grid * std::max(1, count_rounds);
    const size_t count_lds = (size_t)ngroups * (kWave / 2) * sizeof(int) + table_lds;
#define FPC_LAUNCH_COUNT(M)                                                                                              \
    if (count_waves >= 6) FPC_LAUNCH_COUNT2(M, 6); else if (count_waves == 5) FPC_LAUNCH_COUNT2(M, 5); else FPC_LAUNCH_COUNT2(M, 4)
#define FPC_LAUNCH_COUNT2(M, WV)                                                                                         \
    hipLaunchKernelGGL((k_vote_count<M, WV>), dim3(count_grid), dim3(256), count_lds, s, W, HW, w.nch, w.units, w.n_units, hn, \
                       w.hnp, task_target, count_slices, kappa1, kappa2, inlier_thresh, max_num, seed, keep, lds_table,     \
                       w.chunk_pre, w.plan, w.hx, w.hy, w.hxs, w.hys, w.eg, w.upper, w.list, w.clist, count_dbg)
    if (fast) { FPC_LAUNCH_COUNT(kModeCones); } else { FPC_LAUNCH_COUNT(kModeReference); }
#undef FPC_LAUNCH_COUNT
#undef FPC_LAUNCH_COUNT2

    // 4. winner, its inliers, refinement: one task per work unit
    const int fin_grid = (int)std::min<long long>(std::max<long long>((long long)n * std::min(w.nbx, nb_launch), 1), 2048);
#define FPC_LAUNCH_FINAL(M)                                                                                              \
    hipLaunchKernelGGL(k_vote_final<M>, dim3(fin_grid), dim3(256), table_lds, s, W, HW, w.nch, n, n_dev, w.units, w.n_units, hn, \
                       w.hnp, w.nbx, inlier_thresh, kappa1, kappa2, efac, max_num, seed, keep, lds_table, w.chunk_pre, w.plan, \
                       w.hyp, w.upper, w.list, w.clist, w.tickets, w.partial, out_xy, out_tn, out_win_idx, out_win_count,   \
                       out_inl_count, out_refine)
    if (fast) FPC_LAUNCH_FINAL(kModeCones); else FPC_LAUNCH_FINAL(kModeReference);
#undef FPC_LAUNCH_FINAL

    // diagnostics (never on the product path): copies of the hypotheses and of the count rows
    if (out_hyp) {
        hipError_t e = hipMemcpyAsync(out_hyp, w.hyp, sizeof(float) * (size_t)n * hn * 2, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    }
    if (out_counts) {
        hipError_t e = hipMemcpy2DAsync(out_counts, sizeof(int32_t) * (size_t)hn, w.upper, sizeof(int32_t) * (size_t)w.hnp,
                                        sizeof(int32_t) * (size_t)hn, (size_t)n, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    }
    return check_launch();
}
