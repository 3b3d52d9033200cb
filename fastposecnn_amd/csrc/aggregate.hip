// aggregate.hip — AggregationLayer.forward (F/lib/aggregation_layer.py:61-158)
// from the label plane, without the per-sample one-hot scatter, torch.unique calls and
// [n,A,H,W] gathers of the reference:
//   k_agg_accum     one pass over labels + cat_mask + the 8 averaged planes: per-instance
//                   pixel count, smallest class id, fp64 sums (wave-level pre-reduction when a
//                   wave sees a single label, which is the common case)
//   k_agg_planes_img  the drop-in outputs instance_masks [N,H,W] and masked xy [N,2,H,W] (+ bit words): one label read per
//                   image chunk for all of the image's instances; the workgroup of chunk 0 also takes their means, exp(z),
//                   quaternion re-normalisation, class / sample ids
//   k_agg_fused     (<= 4 frames) both halves in one launch
#include "common.hpp"

namespace fpc {

struct AggWs {
    double* sums;        // [N, 8]   zero-filled per call
    int32_t* cnt;        // [N]      zero-filled
    uint32_t* cls_min;   // [N]      zero-filled; holds 0xFFFFFFFF - (smallest class id seen)
    int32_t* ticket;     // [1]      zero-filled; arrival counter of the fused launch
    int32_t* sample;     // [N]
    size_t zero_bytes, total;
};

static AggWs agg_carve(void* base, int N) {
    AggWs w;
    char* p = (char*)base;
    size_t off = 0;
    w.sums = (double*)(p + off); off = align_up(off + sizeof(double) * 8 * (size_t)N, 256);
    w.cnt = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)N, 256);
    w.cls_min = (uint32_t*)(p + off); off = align_up(off + sizeof(uint32_t) * (size_t)N, 256);
    w.ticket = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t), 256);
    w.zero_bytes = off;
    w.sample = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)N, 256);
    w.total = off;
    return w;
}

constexpr int kAggRows = 8;                   // image rows per wave: a workgroup covers a 64 x 32 pixel tile
__device__ __forceinline__ void agg_flush(int i, const double* v, int n, uint32_t c, int b, double* sums,
                                          int32_t* cnt, uint32_t* cls_min, int32_t* sample) {
#pragma unroll
    for (int a = 0; a < 8; ++a) unsafeAtomicAdd(&sums[(size_t)i * 8 + a], v[a]);
    atomicAdd(&cnt[i], n);
    atomicMax(&cls_min[i], 0xFFFFFFFFu - c);   // stored inverted so that one zero-fill initialises everything
    sample[i] = b;
}

constexpr int kAggSlots = 4;                  // instances a workgroup combines in LDS before it touches global memory

// The global atomics are what this kernel costs: with them stubbed out it runs 6.4 instead of 15.2 us on one frame and 19
// instead of 259 us on 32 (device-scope f64 atomics of 600 waves per frame on 8 N addresses).  So a workgroup first combines
// its waves' sums per instance in LDS (ds_add_f64) and only its <= kAggSlots slot owners go to global memory.
struct AggLds {
    int lab[kAggSlots];            // label + 1 owning the slot, 0 = free
    int cnt[kAggSlots];
    uint32_t cls[kAggSlots];
    double sum[kAggSlots][8];
};

__device__ __forceinline__ void agg_flush_lds(AggLds& s, int i, const double* v, int n, uint32_t c, int b, double* sums,
                                              int32_t* cnt, uint32_t* cls_min, int32_t* sample) {
#pragma unroll
    for (int k = 0; k < kAggSlots; ++k) {
        const int old = atomicCAS(&s.lab[k], 0, i + 1);
        if (old == 0 || old == i + 1) {
#pragma unroll
            for (int a = 0; a < 8; ++a) unsafeAtomicAdd(&s.sum[k][a], v[a]);
            atomicAdd(&s.cnt[k], n);
            atomicMin(&s.cls[k], c);
            return;
        }
    }
    agg_flush(i, v, n, c, b, sums, cnt, cls_min, sample);      // a fifth instance under this workgroup: straight to global memory
}


// One wave = a 64-pixel-wide, kAggRows-tall strip (coalesced 256-byte rows).  Instances are blobs, so
// going DOWN a strip the wave usually stays inside one label: lanes keep private fp64 sums while the
// wave-uniform label is unchanged and the wave reduces (shuffles) + issues ONE set of global atomics
// only when that label changes or the strip ends.  Rows where two instances meet inside the 64 pixels
// fall back to per-lane atomics for the minority label.   grid (ceil(W/64), ceil(H/(4*kAggRows)), B)
// The strip's eight label rows are requested together, then the nine planes of four rows at a time under
// their labels: three dependent memory round trips per wave instead of sixteen (one per row and stage).
__device__ __forceinline__ void agg_accum_block(AggLds& s, int bx, int by, int bz, const int32_t* __restrict__ labels,
                                                const int64_t* __restrict__ cm, const float* __restrict__ quat,
                                                const float* __restrict__ scales, const float* __restrict__ z,
                                                int H, int W, int N, double* __restrict__ sums, int32_t* __restrict__ cnt,
                                                uint32_t* __restrict__ cls_min, int32_t* __restrict__ sample) {
    if (threadIdx.x < kAggSlots) {
        s.lab[threadIdx.x] = 0; s.cnt[threadIdx.x] = 0; s.cls[threadIdx.x] = 0xFFFFFFFFu;
#pragma unroll
        for (int a = 0; a < 8; ++a) s.sum[threadIdx.x][a] = 0.0;
    }
    __syncthreads();
    const int b = bz, HW = H * W;
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    const int x = bx * kWave + lane;
    const int y0 = (by * 4 + w) * kAggRows;
    int lab[kAggRows];
#pragma unroll
    for (int r = 0; r < kAggRows; ++r) {
        const int y = y0 + r;
        int l = (x < W && y < H) ? labels[(size_t)b * HW + (size_t)y * W + x] : 0;
        lab[r] = l > N ? 0 : l;
    }
    unsigned long long any = 0ull;
#pragma unroll
    for (int r = 0; r < kAggRows; ++r) any |= __ballot(lab[r] > 0);
    if (any != 0ull) {                                       // wave-uniform: not a strip of background
    int cur = 0, n = 0;
    uint32_t c = 0xFFFFFFFFu;
    double v[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) v[a] = 0.0;
    auto wave_flush = [&]() {
        double t[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) t[a] = wave_reduce_add(v[a]);
        int nn = wave_reduce_add(n);
        uint32_t cc = c;
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) cc = min(cc, (uint32_t)__shfl_down((int)cc, o, kWave));
        if (lane == 0 && nn > 0) agg_flush_lds(s, cur - 1, t, nn, cc, b, sums, cnt, cls_min, sample);
        n = 0; c = 0xFFFFFFFFu;
#pragma unroll
        for (int a = 0; a < 8; ++a) v[a] = 0.0;
    };
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float q[4][8];
        long long cls[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {                        // the loads of four rows in flight together
            const int r = half * 4 + k;
            cls[k] = 0;
#pragma unroll
            for (int a = 0; a < 8; ++a) q[k][a] = 0.0f;
            if (lab[r] > 0) {
                const size_t p = (size_t)(y0 + r) * W + x, o = (size_t)b * HW + p;
                cls[k] = cm[o];
#pragma unroll
                for (int a = 0; a < 4; ++a) q[k][a] = quat[((size_t)b * 4 + a) * HW + p];
#pragma unroll
                for (int a = 0; a < 3; ++a) q[k][4 + a] = scales[((size_t)b * 3 + a) * HW + p];
                q[k][7] = z[o];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int l = lab[half * 4 + k];
            const bool act = l > 0;
            const unsigned long long m = __ballot(act);
            if (m == 0) continue;                            // wave-uniform
            const int first = __builtin_amdgcn_readlane(l, __ffsll((long long)m) - 1);
            if (first != cur) {                              // wave-uniform decision
                if (cur > 0) wave_flush();
                cur = first;
            }
            if (!act) continue;
            const uint32_t pc = cls[k] != 0 ? (uint32_t)cls[k] : 0xFFFFFFFFu;
            if (l == cur) {
#pragma unroll
                for (int a = 0; a < 8; ++a) v[a] += (double)q[k][a];
                ++n;
                c = min(c, pc);
            } else {
                double qd[8];
#pragma unroll
                for (int a = 0; a < 8; ++a) qd[a] = (double)q[k][a];
                agg_flush_lds(s, l - 1, qd, 1, pc, b, sums, cnt, cls_min, sample);   // a second instance inside these 64 pixels
            }
        }
    }
    if (cur > 0) wave_flush();
    }
    __syncthreads();
    if (threadIdx.x < kAggSlots && s.lab[threadIdx.x] != 0)
        agg_flush(s.lab[threadIdx.x] - 1, s.sum[threadIdx.x], s.cnt[threadIdx.x], s.cls[threadIdx.x], b, sums, cnt, cls_min, sample);
}

__global__ __launch_bounds__(256) void k_agg_accum(const int32_t* __restrict__ labels,
                                                   const int64_t* __restrict__ cm, const float* __restrict__ quat,
                                                   const float* __restrict__ scales, const float* __restrict__ z,
                                                   int H, int W, int N, const int32_t* __restrict__ n_dev,
                                                   double* __restrict__ sums, int32_t* __restrict__ cnt,
                                                   uint32_t* __restrict__ cls_min, int32_t* __restrict__ sample) {
    if (n_dev) N = min(N, *n_dev);
    __shared__ AggLds s;
    agg_accum_block(s, blockIdx.x, blockIdx.y, blockIdx.z, labels, cm, quat, scales, z, H, W, N, sums, cnt, cls_min, sample);
}

// means, exp(z), quaternion re-normalisation, class / sample ids of instance i (one thread)
__device__ __forceinline__ void agg_finalize_one(int i, const double* __restrict__ sums, const int32_t* __restrict__ cnt,
                                                 const uint32_t* __restrict__ cls_min, const int32_t* __restrict__ sample,
                                                 int64_t* __restrict__ class_ids, int64_t* __restrict__ sample_ids,
                                                 float* __restrict__ oq, float* __restrict__ os, float* __restrict__ oz,
                                                 float* __restrict__ stats) {
    double c = (double)cnt[i];
    float q[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) q[a] = (float)div_ieee(sums[(size_t)i * 8 + a], c);
    float nq = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (stats) { stats[(size_t)i * 2] = (float)cnt[i]; stats[(size_t)i * 2 + 1] = nq; }     // for the backward (train.hip)
    if (nq == 0.0f) nq = 1.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a) oq[(size_t)i * 4 + a] = div_ieee(q[a], nq);
#pragma unroll
    for (int a = 0; a < 3; ++a) os[(size_t)i * 3 + a] = (float)div_ieee(sums[(size_t)i * 8 + 4 + a], c);
    oz[i] = expf((float)div_ieee(sums[(size_t)i * 8 + 7], c));
    uint32_t cm = 0xFFFFFFFFu - cls_min[i];
    class_ids[i] = cm == 0xFFFFFFFFu ? 0 : (int64_t)cm;
    sample_ids[i] = cnt[i] > 0 ? (int64_t)sample[i] : -1;
}

// when no plane output is wanted: the per-instance results alone
__global__ void k_agg_finalize(int N, const int32_t* __restrict__ n_dev, const double* __restrict__ sums, const int32_t* __restrict__ cnt,
                               const uint32_t* __restrict__ cls_min, const int32_t* __restrict__ sample,
                               int64_t* __restrict__ class_ids, int64_t* __restrict__ sample_ids,
                               float* __restrict__ oq, float* __restrict__ os, float* __restrict__ oz,
                               float* __restrict__ stats) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) N = min(N, *n_dev);
    if (i >= N) return;
    agg_finalize_one(i, sums, cnt, cls_min, sample, class_ids, sample_ids, oq, os, oz, stats);
}

// grid (ceil(HW/4096), N): a workgroup writes one 4096-pixel chunk of one instance (four 4-pixel groups per lane, their
// label loads issued together); the first block of every instance also finalises it (no launch of its own).
// the planes of chunk `bx` of instance i, which lives in image b
__device__ __forceinline__ void agg_planes_block(int bx, int i, int b, const int32_t* __restrict__ labels, const float* __restrict__ xy,
                                                 int HW, float* __restrict__ inst_masks, float* __restrict__ oxy,
                                                 uint64_t* __restrict__ bits, int nwords) {
    const int32_t* L = labels + (size_t)b * HW;
    const bool vec = (HW & 3) == 0;
    const int lane = threadIdx.x & (kWave - 1);
    int4 lab[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int p0 = bx * 4096 + it * 1024 + threadIdx.x * 4;
        lab[it] = make_int4(0, 0, 0, 0);
        if (vec) {
            if (p0 < HW) lab[it] = *reinterpret_cast<const int4*>(L + p0);
        } else {
            if (p0 < HW) lab[it].x = L[p0];
            if (p0 + 1 < HW) lab[it].y = L[p0 + 1];
            if (p0 + 2 < HW) lab[it].z = L[p0 + 2];
            if (p0 + 3 < HW) lab[it].w = L[p0 + 3];
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int p0 = bx * 4096 + it * 1024 + threadIdx.x * 4;
        const bool f0 = lab[it].x == i + 1, f1 = lab[it].y == i + 1, f2 = lab[it].z == i + 1, f3 = lab[it].w == i + 1;      // labels are >= 1
        if (bits) {
            // the instance's foreground as bit words for the vote's scan (bit j of word w = pixel 64 w + j, zero past HW):
            // four ballots, lanes 0..3 interleave the wave's four words
            const uint64_t b0 = __ballot(f0), b1 = __ballot(f1), b2 = __ballot(f2), b3 = __ballot(f3);
            if (lane < 4) {
                auto spread = [](uint64_t x) {      // bit m of the low 16 -> bit 4 m
                    x &= 0xFFFFull;
                    x = (x | (x << 24)) & 0x000000FF000000FFull;
                    x = (x | (x << 12)) & 0x000F000F000F000Full;
                    x = (x | (x << 6)) & 0x0303030303030303ull;
                    x = (x | (x << 3)) & 0x1111111111111111ull;
                    return x;
                };
                const int sh = 16 * lane;
                const uint64_t word = spread(b0 >> sh) | (spread(b1 >> sh) << 1) | (spread(b2 >> sh) << 2) | (spread(b3 >> sh) << 3);
                const int wi = bx * 64 + it * 16 + (threadIdx.x / kWave) * 4 + lane;
                if (wi < nwords) bits[(size_t)i * nwords + wi] = word;
            }
        }
        if (p0 >= HW) continue;
        if (vec) {
            if (inst_masks)
                *reinterpret_cast<float4*>(inst_masks + (size_t)i * HW + p0) =
                    make_float4(f0 ? 1.f : 0.f, f1 ? 1.f : 0.f, f2 ? 1.f : 0.f, f3 ? 1.f : 0.f);
            if (oxy) {
                const bool any = f0 | f1 | f2 | f3;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (any) {
                        const float4 s = *reinterpret_cast<const float4*>(xy + ((size_t)b * 2 + c) * HW + p0);
                        v = make_float4(f0 ? s.x : 0.f, f1 ? s.y : 0.f, f2 ? s.z : 0.f, f3 ? s.w : 0.f);
                    }
                    *reinterpret_cast<float4*>(oxy + ((size_t)i * 2 + c) * HW + p0) = v;
                }
            }
        } else {
            const bool fl[4] = {f0, f1, f2, f3};
            for (int k = 0; k < 4 && p0 + k < HW; ++k) {
                const int p = p0 + k;
                if (inst_masks) inst_masks[(size_t)i * HW + p] = fl[k] ? 1.f : 0.f;
                if (oxy) {
                    oxy[((size_t)i * 2 + 0) * HW + p] = fl[k] ? xy[((size_t)b * 2 + 0) * HW + p] : 0.f;
                    oxy[((size_t)i * 2 + 1) * HW + p] = fl[k] ? xy[((size_t)b * 2 + 1) * HW + p] : 0.f;
                }
            }
        }
    }
}

// grid (ceil(HW/4096), B): the large-batch form.  A workgroup reads ONE 4096-pixel chunk of an image's label plane and of
// its two vote planes and writes that chunk of EVERY instance of the image (masks, masked vote field, bit words): the labels
// of an image are read once, not once per instance (round 3: 236 of the 944 MB the kernel moved on 32 frames).  The
// instances of image b are a contiguous label range (components are numbered image by image): found from `sample`, which
// the accumulation wrote, by one sweep of the workgroup.  The workgroup of chunk 0 finalises the image's instances.
__global__ __launch_bounds__(256) void k_agg_planes_img(const int32_t* __restrict__ labels, const float* __restrict__ xy,
                                                        const int32_t* __restrict__ sample, int HW, int N,
                                                        const int32_t* __restrict__ n_dev,
                                                        float* __restrict__ inst_masks, float* __restrict__ oxy,
                                                        const double* __restrict__ sums, const int32_t* __restrict__ cnt,
                                                        const uint32_t* __restrict__ cls_min, int64_t* __restrict__ class_ids,
                                                        int64_t* __restrict__ sample_ids, float* __restrict__ oq,
                                                        float* __restrict__ os, float* __restrict__ oz, float* __restrict__ stats,
                                                        uint64_t* __restrict__ bits, int nwords) {
    __shared__ int s_rng[2];
    const int b = blockIdx.y, bx = blockIdx.x;
    const int n = n_dev ? min(N, *n_dev) : N;
    if (threadIdx.x == 0) { s_rng[0] = 0x7fffffff; s_rng[1] = -1; }
    __syncthreads();
    int lo = 0x7fffffff, hi = -1;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        if (sample[i] == b && cnt[i] > 0) { lo = min(lo, i); hi = max(hi, i); }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, kWave)); hi = max(hi, __shfl_xor(hi, o, kWave)); }
    if ((threadIdx.x & (kWave - 1)) == 0 && hi >= 0) { atomicMin(&s_rng[0], lo); atomicMax(&s_rng[1], hi); }
    __syncthreads();
    lo = s_rng[0]; hi = s_rng[1];
    if (hi < 0) return;                                  // an image without instances
    if (bx == 0)
        for (int i = lo + (int)threadIdx.x; i <= hi; i += blockDim.x)
            agg_finalize_one(i, sums, cnt, cls_min, sample, class_ids, sample_ids, oq, os, oz, stats);
    const int32_t* L = labels + (size_t)b * HW;
    const bool vec = (HW & 3) == 0;
    const int lane = threadIdx.x & (kWave - 1);
    int4 lab[4];
    float4 vx[4], vy[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int p0 = bx * 4096 + it * 1024 + threadIdx.x * 4;
        lab[it] = make_int4(0, 0, 0, 0);
        vx[it] = vy[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vec) {
            if (p0 < HW) lab[it] = *reinterpret_cast<const int4*>(L + p0);
        } else {
            if (p0 < HW) lab[it].x = L[p0];
            if (p0 + 1 < HW) lab[it].y = L[p0 + 1];
            if (p0 + 2 < HW) lab[it].z = L[p0 + 2];
            if (p0 + 3 < HW) lab[it].w = L[p0 + 3];
        }
    }
    if (oxy) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int p0 = bx * 4096 + it * 1024 + threadIdx.x * 4;
            if ((lab[it].x | lab[it].y | lab[it].z | lab[it].w) == 0) continue;      // labels are >= 0: a background group
            if (vec) {
                vx[it] = *reinterpret_cast<const float4*>(xy + ((size_t)b * 2 + 0) * HW + p0);
                vy[it] = *reinterpret_cast<const float4*>(xy + ((size_t)b * 2 + 1) * HW + p0);
            } else {
                float ax[4] = {0, 0, 0, 0}, ay[4] = {0, 0, 0, 0};
                for (int k = 0; k < 4 && p0 + k < HW; ++k) { ax[k] = xy[((size_t)b * 2 + 0) * HW + p0 + k]; ay[k] = xy[((size_t)b * 2 + 1) * HW + p0 + k]; }
                vx[it] = make_float4(ax[0], ax[1], ax[2], ax[3]); vy[it] = make_float4(ay[0], ay[1], ay[2], ay[3]);
            }
        }
    }
    for (int i = lo; i <= hi; ++i) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int p0 = bx * 4096 + it * 1024 + threadIdx.x * 4;
            const bool f0 = lab[it].x == i + 1, f1 = lab[it].y == i + 1, f2 = lab[it].z == i + 1, f3 = lab[it].w == i + 1;
            if (bits) {
                // four ballots, lanes 0..3 interleave the wave's four words (as agg_planes_block)
                const uint64_t b0 = __ballot(f0), b1 = __ballot(f1), b2 = __ballot(f2), b3 = __ballot(f3);
                if (lane < 4) {
                    auto spread = [](uint64_t x) {      // bit m of the low 16 -> bit 4 m
                        x &= 0xFFFFull;
                        x = (x | (x << 24)) & 0x000000FF000000FFull;
                        x = (x | (x << 12)) & 0x000F000F000F000Full;
                        x = (x | (x << 6)) & 0x0303030303030303ull;
                        x = (x | (x << 3)) & 0x1111111111111111ull;
                        return x;
                    };
                    const int sh = 16 * lane;
                    const uint64_t word = spread(b0 >> sh) | (spread(b1 >> sh) << 1) | (spread(b2 >> sh) << 2) | (spread(b3 >> sh) << 3);
                    const int wi = bx * 64 + it * 16 + (threadIdx.x / kWave) * 4 + lane;
                    if (wi < nwords) bits[(size_t)i * nwords + wi] = word;
                }
            }
            if (p0 >= HW) continue;
            if (vec) {
                // streaming stores: 3.7 MB per instance that this kernel never reads back
                if (inst_masks)
                    __builtin_nontemporal_store(f32x4{f0 ? 1.f : 0.f, f1 ? 1.f : 0.f, f2 ? 1.f : 0.f, f3 ? 1.f : 0.f},
                                                reinterpret_cast<f32x4*>(inst_masks + (size_t)i * HW + p0));
                if (oxy) {
                    __builtin_nontemporal_store(f32x4{f0 ? vx[it].x : 0.f, f1 ? vx[it].y : 0.f, f2 ? vx[it].z : 0.f, f3 ? vx[it].w : 0.f},
                                                reinterpret_cast<f32x4*>(oxy + ((size_t)i * 2 + 0) * HW + p0));
                    __builtin_nontemporal_store(f32x4{f0 ? vy[it].x : 0.f, f1 ? vy[it].y : 0.f, f2 ? vy[it].z : 0.f, f3 ? vy[it].w : 0.f},
                                                reinterpret_cast<f32x4*>(oxy + ((size_t)i * 2 + 1) * HW + p0));
                }
            } else {
                const bool fl[4] = {f0, f1, f2, f3};
                const float ax[4] = {vx[it].x, vx[it].y, vx[it].z, vx[it].w}, ay[4] = {vy[it].x, vy[it].y, vy[it].z, vy[it].w};
                for (int k = 0; k < 4 && p0 + k < HW; ++k) {
                    const int p = p0 + k;
                    if (inst_masks) inst_masks[(size_t)i * HW + p] = fl[k] ? 1.f : 0.f;
                    if (oxy) {
                        oxy[((size_t)i * 2 + 0) * HW + p] = fl[k] ? ax[k] : 0.f;
                        oxy[((size_t)i * 2 + 1) * HW + p] = fl[k] ? ay[k] : 0.f;
                    }
                }
            }
        }
    }
}

// ONE launch for both halves when the image of an instance is known without the accumulation (root_pix, from
// fpc_cc_label): blocks [0, nacc) accumulate, the rest write planes; the accumulating block that arrives last (ticket)
// finalises every instance.  The planes never wait for the sums.
struct AggFusedArgs {
    const int32_t* labels; const int64_t* cm; const float* quat; const float* scales; const float* xy; const float* z;
    const int32_t* n_dev; const int32_t* root_pix;
    int B, H, W, N, gax, gay, gpx, nwords;
    double* sums; int32_t* cnt; uint32_t* cls_min; int32_t* sample; int32_t* ticket;
    int64_t* class_ids; int64_t* sample_ids; float* inst_masks; float* oq; float* os; float* oz; float* oxy; float* stats;
    uint64_t* bits;
};

__global__ __launch_bounds__(256) void k_agg_fused(const AggFusedArgs a) {
    __shared__ AggLds s;
    __shared__ int s_last;
    const int N = a.n_dev ? min(a.N, *a.n_dev) : a.N;
    const int HW = a.H * a.W;
    const int nacc = a.gax * a.gay * a.B;
    int bid = blockIdx.x;
    if (bid < nacc) {
        const int bx = bid % a.gax; bid /= a.gax;
        const int by = bid % a.gay;
        const int bz = bid / a.gay;
        agg_accum_block(s, bx, by, bz, a.labels, a.cm, a.quat, a.scales, a.z, a.H, a.W, N, a.sums, a.cnt, a.cls_min, a.sample);
        // the flushing threads drain their device-scope atomics before the ticket is taken (no cache write-back / invalidate:
        // the sums live where atomics are performed, the last block reads them with cache-bypassing loads)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nacc - 1;
        __syncthreads();
        if (!s_last) return;
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
            // sums / counts were written by atomics of other workgroups: read them past this CU's caches
            double sm[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                sm[k] = __builtin_bit_cast(double, __hip_atomic_load((const unsigned long long*)(a.sums + (size_t)i * 8 + k), __ATOMIC_RELAXED,
                                                                     __HIP_MEMORY_SCOPE_AGENT));
            const int32_t c = __hip_atomic_load(a.cnt + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t cl = __hip_atomic_load(a.cls_min + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int32_t smp = a.root_pix[i] / HW;
            agg_finalize_one(0, sm, &c, &cl, &smp, a.class_ids + i, a.sample_ids + i, a.oq + (size_t)i * 4, a.os + (size_t)i * 3,
                             a.oz + i, a.stats ? a.stats + (size_t)i * 2 : nullptr);
        }
        return;
    }
    bid -= nacc;
    const int bx = bid % a.gpx, i = bid / a.gpx;
    if (i >= N) return;
    agg_planes_block(bx, i, a.root_pix[i] / HW, a.labels, a.xy, HW, a.inst_masks, a.oxy, a.bits, a.nwords);
}

}  // namespace fpc

using namespace fpc;

extern "C" size_t fpc_aggregate_workspace_bytes(int N) {
    if (N <= 0) return 256;
    return agg_carve(nullptr, N).total;
}

extern "C" int fpc_aggregate(const int32_t* labels, const int64_t* cat_mask, const float* quat, const float* scales,
                             const float* xy, const float* z, int B, int H, int W, int N, const int32_t* n_dev,
                             int64_t* class_ids,
                             int64_t* sample_ids, float* inst_masks, float* oq, float* os, float* oz, float* oxy,
                             float* out_stats, void* ws, size_t ws_bytes, fpc_stream_t stream) {
    return fpc_aggregate_bits(labels, cat_mask, quat, scales, xy, z, B, H, W, N, n_dev, class_ids, sample_ids, inst_masks, oq, os, oz, oxy,
                              out_stats, nullptr, nullptr, ws, ws_bytes, stream);
}

extern "C" int fpc_aggregate_bits(const int32_t* labels, const int64_t* cat_mask, const float* quat, const float* scales,
                                  const float* xy, const float* z, int B, int H, int W, int N, const int32_t* n_dev,
                                  int64_t* class_ids, int64_t* sample_ids, float* inst_masks, float* oq, float* os, float* oz,
                                  float* oxy, float* out_stats, uint64_t* inst_bits, const int32_t* root_pix, void* ws,
                                  size_t ws_bytes, fpc_stream_t stream) {
    if (B < 0 || H < 1 || W < 1 || N < 0) return FPC_EINVAL;
    if (N == 0 || B == 0) return FPC_OK;
    if (B > 65535 || N > 65535) return FPC_EINVAL;
    if (!labels || !cat_mask || !quat || !scales || !xy || !z || !class_ids || !sample_ids || !oq || !os || !oz ||
        !ws)
        return FPC_EINVAL;
    if (((uintptr_t)ws & 255) != 0) return FPC_EWORKSPACE;
    // float4 plane accesses need 16-byte aligned plane bases
    if (((uintptr_t)inst_masks & 15) || ((uintptr_t)oxy & 15) || ((uintptr_t)xy & 15) || ((uintptr_t)labels & 15))
        return FPC_EINVAL;
    AggWs w = agg_carve(ws, N);
    if (ws_bytes < w.total) return FPC_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    int HW = H * W;
    hipError_t e = hipMemsetAsync(ws, 0, w.zero_bytes, s);
    if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    const int nwords = cdiv(HW, 4096) * 64;                 // fpc_mask_bits_words(H, W): whole 4096-pixel chunks
    int gx = cdiv(HW, 4096);
    if (((uintptr_t)inst_bits & 7) != 0) return FPC_EINVAL;
    if (root_pix && (inst_masks || oxy || inst_bits) && (long long)B * HW <= 4LL * 640 * 480) {
        // the image of instance i is root_pix[i] / (H W): the planes do not wait for the accumulation, one launch does both.
        // For a few frames only (13.0 against 9.8 + 8.5 us on one): the planes part then runs at the accumulation's register
        // count, which costs the bandwidth-bound large batches more than the launch saves (236 against 51 + 171 us on 32).
        AggFusedArgs a{};
        a.labels = labels; a.cm = cat_mask; a.quat = quat; a.scales = scales; a.xy = xy; a.z = z; a.n_dev = n_dev; a.root_pix = root_pix;
        a.B = B; a.H = H; a.W = W; a.N = N; a.gax = cdiv(W, kWave); a.gay = cdiv(H, 4 * kAggRows); a.gpx = gx; a.nwords = nwords;
        a.sums = w.sums; a.cnt = w.cnt; a.cls_min = w.cls_min; a.sample = w.sample; a.ticket = w.ticket;
        a.class_ids = class_ids; a.sample_ids = sample_ids; a.inst_masks = inst_masks; a.oq = oq; a.os = os; a.oz = oz; a.oxy = oxy;
        a.stats = out_stats; a.bits = inst_bits;
        const long long grid = (long long)a.gax * a.gay * B + (long long)gx * N;
        if (grid > 0x7FFFFFFFLL) return FPC_EINVAL;
        hipLaunchKernelGGL(k_agg_fused, dim3((unsigned)grid), dim3(256), 0, s, a);
        return check_launch();
    }
    hipLaunchKernelGGL(k_agg_accum, dim3(cdiv(W, kWave), cdiv(H, 4 * kAggRows), B), dim3(256), 0, s, labels, cat_mask, quat, scales, z,
                       H, W, N, n_dev, w.sums,
                       w.cnt, w.cls_min, w.sample);
    if (inst_masks || oxy || inst_bits)
        hipLaunchKernelGGL(k_agg_planes_img, dim3(gx, B), dim3(256), 0, s, labels, xy, w.sample, HW, N, n_dev, inst_masks, oxy, w.sums, w.cnt,
                           w.cls_min, class_ids, sample_ids, oq, os, oz, out_stats, inst_bits, nwords);
    else
        hipLaunchKernelGGL(k_agg_finalize, dim3(cdiv(N, 64)), dim3(64), 0, s, N, n_dev, w.sums, w.cnt, w.cls_min, w.sample,
                           class_ids, sample_ids, oq, os, oz, out_stats);
    return check_launch();
}
