// aggregate.hip — AggregationLayer.forward (F/lib/aggregation_layer.py:61-158)
// from the label plane, without the per-sample one-hot scatter, torch.unique calls and
// [n,A,H,W] gathers of the reference:
//   k_agg_accum     one pass over labels + cat_mask + the 8 averaged planes: per-instance
//                   pixel count, smallest class id, fp64 sums (wave-level pre-reduction when a
//                   wave sees a single label, which is the common case)
//   k_agg_finalize  means, exp(z), quaternion re-normalisation, class / sample ids
//   k_agg_planes    the drop-in outputs instance_masks [N,H,W] and masked xy [N,2,H,W]
#include "common.hpp"

namespace fpc {

struct AggWs {
    double* sums;        // [N, 8]   zero-filled per call
    int32_t* cnt;        // [N]      zero-filled
    uint32_t* cls_min;   // [N]      0xFFFFFFFF-filled
    int32_t* sample;     // [N]
    size_t zero_bytes, ff_off, ff_bytes, total;
};

static AggWs agg_carve(void* base, int N) {
    AggWs w;
    char* p = (char*)base;
    size_t off = 0;
    w.sums = (double*)(p + off); off = align_up(off + sizeof(double) * 8 * (size_t)N, 256);
    w.cnt = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)N, 256);
    w.zero_bytes = off;
    w.ff_off = off;
    w.cls_min = (uint32_t*)(p + off); off = align_up(off + sizeof(uint32_t) * (size_t)N, 256);
    w.ff_bytes = off - w.ff_off;
    w.sample = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)N, 256);
    w.total = off;
    return w;
}

constexpr int kAggIters = 16;                 // 256 threads x 16 = 4096 pixels per workgroup
constexpr int kAggPx = 256 * kAggIters;
constexpr int kAggRecs = 4 * kAggIters;       // one record per (iteration, wave)

__device__ __forceinline__ void agg_flush(int i, const double* v, int n, uint32_t c, int b, double* sums,
                                          int32_t* cnt, uint32_t* cls_min, int32_t* sample) {
#pragma unroll
    for (int a = 0; a < 8; ++a) unsafeAtomicAdd(&sums[(size_t)i * 8 + a], v[a]);
    atomicAdd(&cnt[i], n);
    atomicMin(&cls_min[i], c);
    sample[i] = b;
}

// grid (ceil(HW/4096), B).  A wave whose 64 pixels carry one label (the common case) reduces
// them with shuffles into one LDS record; the workgroup then merges its <= 64 records per label
// in a fixed order and issues ONE set of global atomics per (workgroup, label).  Waves that
// straddle several labels fall back to per-lane atomics.
__global__ __launch_bounds__(256) void k_agg_accum(const int32_t* __restrict__ labels,
                                                   const int64_t* __restrict__ cm, const float* __restrict__ quat,
                                                   const float* __restrict__ scales, const float* __restrict__ z,
                                                   int HW, int N, double* __restrict__ sums,
                                                   int32_t* __restrict__ cnt, uint32_t* __restrict__ cls_min,
                                                   int32_t* __restrict__ sample) {
    __shared__ int s_label[kAggRecs];
    __shared__ int s_n[kAggRecs];
    __shared__ uint32_t s_cls[kAggRecs];
    __shared__ double s_v[kAggRecs][8];
    int b = blockIdx.y;
    int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    int p0 = blockIdx.x * kAggPx;
#pragma unroll 1
    for (int it = 0; it < kAggIters; ++it) {
        int p = p0 + it * 256 + threadIdx.x;
        int l = 0;
        if (p < HW) l = labels[(size_t)b * HW + p];
        if (l > N) l = 0;
        bool act = l > 0;
        unsigned long long m = __ballot(act);
        int rec = it * 4 + w;
        if (m == 0) {
            if (lane == 0) s_label[rec] = 0;
            continue;
        }
        double v[8];
        uint32_t c = 0xFFFFFFFFu;
        if (act) {
            size_t o = (size_t)b * HW + p;
            long long cc = cm[o];
            if (cc != 0) c = (uint32_t)cc;
#pragma unroll
            for (int a = 0; a < 4; ++a) v[a] = (double)quat[((size_t)b * 4 + a) * HW + p];
#pragma unroll
            for (int a = 0; a < 3; ++a) v[4 + a] = (double)scales[((size_t)b * 3 + a) * HW + p];
            v[7] = (double)z[o];
        } else {
#pragma unroll
            for (int a = 0; a < 8; ++a) v[a] = 0.0;
        }
        int first = __builtin_amdgcn_readlane(l, __ffsll((long long)m) - 1);
        bool uniform = __ballot(act && l != first) == 0;
        if (uniform) {
#pragma unroll
            for (int a = 0; a < 8; ++a) v[a] = wave_reduce_add(v[a]);
#pragma unroll
            for (int o = kWave / 2; o > 0; o >>= 1) c = min(c, (uint32_t)__shfl_down((int)c, o, kWave));
            if (lane == 0) {
                s_label[rec] = first;
                s_n[rec] = __popcll(m);
                s_cls[rec] = c;
#pragma unroll
                for (int a = 0; a < 8; ++a) s_v[rec][a] = v[a];
            }
        } else {
            if (lane == 0) s_label[rec] = 0;
            if (act) agg_flush(l - 1, v, 1, c, b, sums, cnt, cls_min, sample);
        }
    }
    __syncthreads();
    int r = threadIdx.x;
    if (r < kAggRecs) {
        int l = s_label[r];
        bool lead = l > 0;
        for (int k = 0; k < r && lead; ++k) lead = s_label[k] != l;
        if (lead) {
            double v[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) v[a] = s_v[r][a];
            int n = s_n[r];
            uint32_t c = s_cls[r];
            for (int k = r + 1; k < kAggRecs; ++k) {
                if (s_label[k] != l) continue;
#pragma unroll
                for (int a = 0; a < 8; ++a) v[a] += s_v[k][a];
                n += s_n[k];
                c = min(c, s_cls[k]);
            }
            agg_flush(l - 1, v, n, c, b, sums, cnt, cls_min, sample);
        }
    }
}

__global__ void k_agg_finalize(int N, const double* __restrict__ sums, const int32_t* __restrict__ cnt,
                               const uint32_t* __restrict__ cls_min, const int32_t* __restrict__ sample,
                               int64_t* __restrict__ class_ids, int64_t* __restrict__ sample_ids,
                               float* __restrict__ oq, float* __restrict__ os, float* __restrict__ oz) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double c = (double)cnt[i];
    float q[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) q[a] = (float)(sums[(size_t)i * 8 + a] / c);
    float nq = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (nq == 0.0f) nq = 1.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a) oq[(size_t)i * 4 + a] = q[a] / nq;
#pragma unroll
    for (int a = 0; a < 3; ++a) os[(size_t)i * 3 + a] = (float)(sums[(size_t)i * 8 + 4 + a] / c);
    oz[i] = expf((float)(sums[(size_t)i * 8 + 7] / c));
    uint32_t cm = cls_min[i];
    class_ids[i] = cm == 0xFFFFFFFFu ? 0 : (int64_t)cm;
    sample_ids[i] = cnt[i] > 0 ? (int64_t)sample[i] : -1;
}

// grid (ceil(HW/1024), N)
__global__ __launch_bounds__(256) void k_agg_planes(const int32_t* __restrict__ labels, const float* __restrict__ xy,
                                                    const int32_t* __restrict__ sample, int HW,
                                                    float* __restrict__ inst_masks, float* __restrict__ oxy) {
    int i = blockIdx.y;
    int b = sample[i];
    int p0 = blockIdx.x * 1024 + threadIdx.x * 4;
    if (p0 >= HW) return;
    const int32_t* L = labels + (size_t)b * HW;
    if ((HW & 3) == 0) {
        int4 l = *reinterpret_cast<const int4*>(L + p0);
        bool f0 = l.x == i + 1, f1 = l.y == i + 1, f2 = l.z == i + 1, f3 = l.w == i + 1;
        if (inst_masks)
            *reinterpret_cast<float4*>(inst_masks + (size_t)i * HW + p0) =
                make_float4(f0 ? 1.f : 0.f, f1 ? 1.f : 0.f, f2 ? 1.f : 0.f, f3 ? 1.f : 0.f);
        if (oxy) {
            bool any = f0 | f1 | f2 | f3;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (any) {
                    float4 s = *reinterpret_cast<const float4*>(xy + ((size_t)b * 2 + c) * HW + p0);
                    v = make_float4(f0 ? s.x : 0.f, f1 ? s.y : 0.f, f2 ? s.z : 0.f, f3 ? s.w : 0.f);
                }
                *reinterpret_cast<float4*>(oxy + ((size_t)i * 2 + c) * HW + p0) = v;
            }
        }
    } else {
        for (int k = 0; k < 4 && p0 + k < HW; ++k) {
            int p = p0 + k;
            bool f = L[p] == i + 1;
            if (inst_masks) inst_masks[(size_t)i * HW + p] = f ? 1.f : 0.f;
            if (oxy) {
                oxy[((size_t)i * 2 + 0) * HW + p] = f ? xy[((size_t)b * 2 + 0) * HW + p] : 0.f;
                oxy[((size_t)i * 2 + 1) * HW + p] = f ? xy[((size_t)b * 2 + 1) * HW + p] : 0.f;
            }
        }
    }
}

}  // namespace fpc

using namespace fpc;

extern "C" size_t fpc_aggregate_workspace_bytes(int N) {
    if (N <= 0) return 256;
    return agg_carve(nullptr, N).total;
}

extern "C" int fpc_aggregate(const int32_t* labels, const int64_t* cat_mask, const float* quat, const float* scales,
                             const float* xy, const float* z, int B, int H, int W, int N, int64_t* class_ids,
                             int64_t* sample_ids, float* inst_masks, float* oq, float* os, float* oz, float* oxy,
                             void* ws, size_t ws_bytes, fpc_stream_t stream) {
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
    if (e == hipSuccess) e = hipMemsetAsync((char*)ws + w.ff_off, 0xFF, w.ff_bytes, s);
    if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    int gx = cdiv(HW, 1024);
    hipLaunchKernelGGL(k_agg_accum, dim3(cdiv(HW, kAggPx), B), dim3(256), 0, s, labels, cat_mask, quat, scales, z, HW, N, w.sums,
                       w.cnt, w.cls_min, w.sample);
    hipLaunchKernelGGL(k_agg_finalize, dim3(cdiv(N, 64)), dim3(64), 0, s, N, w.sums, w.cnt, w.cls_min, w.sample,
                       class_ids, sample_ids, oq, os, oz);
    if (inst_masks || oxy)
        hipLaunchKernelGGL(k_agg_planes, dim3(gx, N), dim3(256), 0, s, labels, xy, w.sample, HW, inst_masks, oxy);
    return check_launch();
}
