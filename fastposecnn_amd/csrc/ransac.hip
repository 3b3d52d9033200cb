// ransac.hip — PVNet-derived RANSAC hough voting for gfx950.
//
//  * fpc_generate_hypothesis / fpc_voting_for_hypothesis: B1-compatible kernels
//    (reference: RV/src/ransac_voting_kernel.cu:11-167).
//  * fpc_ransac_voting_v3: the whole of ransac_voting_layer_v3
//    (RV/ransac_voting_gpu.py:518-607) for a batch of instances without a host
//    round trip and without the hn x tn inlier matrix:
//      k_chunk_count   foreground count per 1024-pixel chunk            (HBM: mask plane)
//      k_chunk_kept    only for instances above max_num: Bernoulli keep  (rare)
//      k_compact       order-preserving stream compaction -> float4 {x,y,dx,dy} list
//                                                                        (HBM: mask + vote planes)
//      k_hypothesis    pair sampling + 2-line intersection
//      k_count         inlier counts, one lane per hypothesis, pixel tile broadcast from LDS
//      k_select_refine arg-max (lowest index on ties), winner re-vote, fp64 normal equations
//
// One RANSAC round: the reference's rounds re-evaluate identical samples.
#include "common.hpp"

namespace fpc {

// ----------------------------------------------------------------------------
// B1 kernels

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
        float nx0 = direct[(size_t)t0 * vn * 2 + vi * 2 + 1];
        float ny0 = -direct[(size_t)t0 * vn * 2 + vi * 2];
        float cx0 = coords[(size_t)t0 * 2], cy0 = coords[(size_t)t0 * 2 + 1];
        float nx1 = direct[(size_t)t1 * vn * 2 + vi * 2 + 1];
        float ny1 = -direct[(size_t)t1 * vn * 2 + vi * 2];
        float cx1 = coords[(size_t)t1 * 2], cy1 = coords[(size_t)t1 * 2 + 1];
        float det_y = nx1 * ny0 - nx0 * ny1;
        float det_x = ny1 * nx0 - ny0 * nx1;
        if (!below_eps(fabsf(det_y)) && !below_eps(fabsf(det_x))) {
            y = (nx1 * (nx0 * cx0 + ny0 * cy0) - nx0 * (nx1 * cx1 + ny1 * cy1)) / det_y;
            x = (ny1 * (nx0 * cx0 + ny0 * cy0) - ny0 * (nx1 * cx1 + ny1 * cy1)) / det_x;
        }
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

constexpr int kChunk = 1024;      // pixels per compaction chunk = 256 threads x 4
constexpr int kTile = 256;        // pixels staged in LDS per counting step
constexpr int kMeta = 8;          // i32 per instance: fg, tn, win_idx, win_cnt, inl, pad...

struct Ws {
    int32_t* chunk_fg;    // [n, nch]
    int32_t* chunk_kept;  // [n, nch]
    int32_t* meta;        // [n, kMeta]
    int32_t* counts;      // [n, hn]
    float* hyp;           // [n, hn, 2]
    float4* px;           // [n, HW]  {x, y, dx, dy}
    size_t zero_bytes;    // leading bytes to clear per call (counts only; see layout)
    size_t total;
};

static Ws carve(void* base, int n, int H, int W, int hn) {
    Ws w;
    size_t HW = (size_t)H * W;
    int nch = cdiv((int)HW, kChunk);
    char* p = (char*)base;
    size_t off = 0;
    w.counts = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)n * hn, 256);
    w.zero_bytes = off;
    w.chunk_fg = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)n * nch, 256);
    w.chunk_kept = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)n * nch, 256);
    w.meta = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)n * kMeta, 256);
    w.hyp = (float*)(p + off); off = align_up(off + sizeof(float) * (size_t)n * hn * 2, 256);
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
                                                     int32_t* __restrict__ chunk_fg) {
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
                                                    int32_t* __restrict__ chunk_kept) {
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
                                                 int32_t* __restrict__ meta, float4* __restrict__ px) {
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

__global__ __launch_bounds__(256) void k_hypothesis(const float4* __restrict__ px, int HW, int hn,
                                                    const int32_t* __restrict__ idxs, uint64_t seed,
                                                    const int32_t* __restrict__ meta, float* __restrict__ hyp) {
    int inst = blockIdx.y;
    int hi = blockIdx.x * blockDim.x + threadIdx.x;
    if (hi >= hn) return;
    int tn = meta[inst * kMeta + 1];
    float x = 0.0f, y = 0.0f;
    if (tn > 0) {
        int t0, t1;
        if (idxs) {
            t0 = idxs[((size_t)inst * hn + hi) * 2];
            t1 = idxs[((size_t)inst * hn + hi) * 2 + 1];
        } else {
            t0 = fpc_rand_index(seed, (uint32_t)inst, (uint32_t)hi, 0u, (uint32_t)tn);
            t1 = fpc_rand_index(seed, (uint32_t)inst, (uint32_t)hi, 1u, (uint32_t)tn);
        }
        if (t0 >= 0 && t0 < tn && t1 >= 0 && t1 < tn) {
            const float4* P = px + (size_t)inst * HW;
            float4 a = P[t0], b = P[t1];
            // RV/src/ransac_voting_kernel.cu:28-45, normal = (dy, -dx)
            float nx0 = a.w, ny0 = -a.z, cx0 = a.x, cy0 = a.y;
            float nx1 = b.w, ny1 = -b.z, cx1 = b.x, cy1 = b.y;
            float det_y = nx1 * ny0 - nx0 * ny1;
            float det_x = ny1 * nx0 - ny0 * nx1;
            if (!below_eps(fabsf(det_y)) && !below_eps(fabsf(det_x))) {
                y = (nx1 * (nx0 * cx0 + ny0 * cy0) - nx0 * (nx1 * cx1 + ny1 * cy1)) / det_y;
                x = (ny1 * (nx0 * cx0 + ny0 * cy0) - ny0 * (nx1 * cx1 + ny1 * cy1)) / det_x;
            }
        }
    }
    hyp[((size_t)inst * hn + hi) * 2] = x;
    hyp[((size_t)inst * hn + hi) * 2 + 1] = y;
}

// grid (ceil(hn/256), S, n).  One lane per hypothesis; pixel tiles are staged in LDS
// (with |n| computed once per pixel) and read back as wave-uniform broadcasts.
__global__ __launch_bounds__(256) void k_count(const float4* __restrict__ px, int HW, int hn, float thresh,
                                               const int32_t* __restrict__ meta, const float* __restrict__ hyp,
                                               int32_t* __restrict__ counts) {
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

// grid (n), block 1024.
__global__ __launch_bounds__(1024) void k_select_refine(const float4* __restrict__ px, int HW, int hn, float thresh,
                                                        int32_t* __restrict__ meta, const float* __restrict__ hyp,
                                                        const int32_t* __restrict__ counts,
                                                        float* __restrict__ out_xy) {
    __shared__ int s_cnt[16], s_idx[16];
    __shared__ double s_sum[16][5];
    __shared__ int s_inl[16];
    __shared__ float s_w[2];
    int inst = blockIdx.x;
    int tn = meta[inst * kMeta + 1];
    int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    if (tn == 0) {
        if (threadIdx.x == 0) {
            out_xy[inst * 2] = 0.0f; out_xy[inst * 2 + 1] = 0.0f;
            meta[inst * kMeta + 2] = -1; meta[inst * kMeta + 3] = 0; meta[inst * kMeta + 4] = 0;
        }
        return;
    }
    // arg-max, lowest index on ties (torch.max, RV/ransac_voting_gpu.py:567)
    int bc = -1, bi = 0x7fffffff;
    for (int h = threadIdx.x; h < hn; h += blockDim.x) {
        int c = counts[(size_t)inst * hn + h];
        if (c > bc) { bc = c; bi = h; }
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        int oc = __shfl_down(bc, o, kWave), oi = __shfl_down(bi, o, kWave);
        if (oc > bc || (oc == bc && oi < bi)) { bc = oc; bi = oi; }
    }
    if (lane == 0) { s_cnt[w] = bc; s_idx[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i)
            if (s_cnt[i] > bc || (s_cnt[i] == bc && s_idx[i] < bi)) { bc = s_cnt[i]; bi = s_idx[i]; }
        // all_win_* start at zero and move only on a strictly larger ratio (:571-574)
        float wx = 0.0f, wy = 0.0f;
        if (bc > 0) { wx = hyp[((size_t)inst * hn + bi) * 2]; wy = hyp[((size_t)inst * hn + bi) * 2 + 1]; }
        s_w[0] = wx; s_w[1] = wy;
        meta[inst * kMeta + 2] = bc > 0 ? bi : -1;
        meta[inst * kMeta + 3] = bc;
    }
    __syncthreads();
    float wx = s_w[0], wy = s_w[1];
    // winner re-vote + normal equations (:583-599), fp64 accumulation
    const float4* P = px + (size_t)inst * HW;
    double a00 = 0, a01 = 0, a11 = 0, b0 = 0, b1 = 0;
    int inl = 0;
    for (int j = threadIdx.x; j < tn; j += blockDim.x) {
        float4 q = P[j];
        float n1 = sqrtf(q.z * q.z + q.w * q.w);
        if (!pair_is_inlier(q.x, q.y, q.z, q.w, n1, wx, wy, thresh)) continue;
        ++inl;
        double nx = (double)q.w, ny = -(double)q.z;
        double bb = nx * (double)q.x + ny * (double)q.y;
        a00 += nx * nx; a01 += nx * ny; a11 += ny * ny;
        b0 += nx * bb; b1 += ny * bb;
    }
    a00 = wave_reduce_add(a00); a01 = wave_reduce_add(a01); a11 = wave_reduce_add(a11);
    b0 = wave_reduce_add(b0); b1 = wave_reduce_add(b1);
    inl = wave_reduce_add(inl);
    if (lane == 0) {
        s_sum[w][0] = a00; s_sum[w][1] = a01; s_sum[w][2] = a11; s_sum[w][3] = b0; s_sum[w][4] = b1;
        s_inl[w] = inl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a00 = a01 = a11 = b0 = b1 = 0; inl = 0;
        for (int i = 0; i < 16; ++i) {
            a00 += s_sum[i][0]; a01 += s_sum[i][1]; a11 += s_sum[i][2]; b0 += s_sum[i][3]; b1 += s_sum[i][4];
            inl += s_inl[i];
        }
        // b_inv (:503-516): inverse when regular, pseudo-inverse when singular
        double x0 = 0.0, x1 = 0.0;
        double tr = a00 + a11, det = a00 * a11 - a01 * a01;
        if (tr > 0.0) {
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
        out_xy[inst * 2] = (float)x0;
        out_xy[inst * 2 + 1] = (float)x1;
        meta[inst * kMeta + 4] = inl;
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
                                    int64_t vs_c, int n, int H, int W, int hn, const int32_t* idxs,
                                    const uint8_t* keep, uint64_t seed, float inlier_thresh, int min_num,
                                    int max_num, float* out_xy, int32_t* out_tn, int32_t* out_win_idx,
                                    int32_t* out_win_count, int32_t* out_inl_count, float* out_hyp,
                                    int32_t* out_counts, void* ws, size_t ws_bytes, fpc_stream_t stream) {
    if (n < 0 || H < 1 || W < 1 || hn < 1 || max_num < 1) return FPC_EINVAL;
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
    hipLaunchKernelGGL(k_chunk_count, dim3(nch, n), dim3(256), 0, s, mask, HW, nch, w.chunk_fg);
    hipLaunchKernelGGL(k_chunk_kept, dim3(nch, n), dim3(256), 0, s, mask, keep, HW, nch, seed, max_num, w.chunk_fg,
                       w.chunk_kept);
    hipLaunchKernelGGL(k_compact, dim3(nch, n), dim3(256), 0, s, mask, vertex, vs_n, vs_h, vs_w, vs_c, keep, W, HW,
                       nch, seed, min_num, max_num, w.chunk_fg, w.chunk_kept, w.meta, w.px);
    int hb = cdiv(hn, 256);
    hipLaunchKernelGGL(k_hypothesis, dim3(hb, n), dim3(256), 0, s, w.px, HW, hn, idxs, seed, w.meta, w.hyp);
    int split = 2048 / (n * hb);
    split = split < 8 ? 8 : (split > 128 ? 128 : split);
    hipLaunchKernelGGL(k_count, dim3(hb, split, n), dim3(256), 0, s, w.px, HW, hn, inlier_thresh, w.meta, w.hyp,
                       w.counts);
    hipLaunchKernelGGL(k_select_refine, dim3(n), dim3(1024), 0, s, w.px, HW, hn, inlier_thresh, w.meta, w.hyp,
                       w.counts, out_xy);
    if (out_tn || out_win_idx || out_win_count || out_inl_count)
        hipLaunchKernelGGL(k_export_meta, dim3(cdiv(n, 256)), dim3(256), 0, s, w.meta, n, out_tn, out_win_idx,
                           out_win_count, out_inl_count);
    if (out_hyp) {
        e = hipMemcpyAsync(out_hyp, w.hyp, sizeof(float) * (size_t)n * hn * 2, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    }
    if (out_counts) {
        e = hipMemcpyAsync(out_counts, w.counts, sizeof(int32_t) * (size_t)n * hn, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    }
    return check_launch();
}
