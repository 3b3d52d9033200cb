// cc.hip — connected-component labelling of (cat_mask != 0), replacing the
// torch -> cupy -> cupyx.scipy.ndimage.label -> DLPack round trip of
// AggregationLayer.batchwise_break_segmentation_mask (F/lib/aggregation_layer.py:160-183).
//
// 4-connectivity inside an image, none across images (the 3x3x3 structuring element
// of :43-59).  Lock-free union-find on global linear pixel indices:
//   k_cc_init     row runs: every pixel points at the start of its run inside the
//                 64-pixel wave segment (ballot + clz), run starts chain to the
//                 previous segment
//   k_cc_merge    vertical unions, only where a run start / run break makes one necessary
//   k_cc_flatten  root of every pixel; root census per 1024-pixel block
//   k_cc_scan     exclusive scan of the block census (one block) -> N
//   k_cc_rank     roots in raster order get labels 1..N (scipy's numbering, continuing
//                 across the batch)
//   k_cc_relabel  labels[p] = rank of root(p)
// The root of a component is its minimum linear index, i.e. its first pixel in raster
// order, so ranking the roots by index reproduces scipy.ndimage.label's order exactly.
// (Round 3 tried flatten + scan + rank as ONE launch — blocks in ticket order, decoupled look-back over 64 predecessors per
// step: 43.4 us against 45.6 us for the six launches at B = 1, 225 us against 181 us at B = 32, where one ticket word
// serves ~88 draws / us and batching the draws serialises the tree walks.  The walks themselves (run starts climbing a
// union-find tree as deep as the blob is tall, one dependent load per level) are the 15-22 us that dominate; not kept.)
#include "common.hpp"

namespace fpc {

constexpr int kCcBlock = 1024;  // pixels per block = 256 threads x 4 (strided by 256)

__device__ __forceinline__ int cc_find(const int32_t* L, int a) {
    while (true) {
        int pa = L[a];
        if (pa == a) return a;
        a = pa;
    }
}

// find with path halving: every visited node is re-pointed at its grandparent.  atomicMin keeps
// parents monotonically decreasing under concurrent unions, so no cycle can form.
__device__ __forceinline__ int cc_find_halve(int32_t* L, int a) {
    while (true) {
        int pa = L[a];
        if (pa == a) return a;
        int gpa = L[pa];
        if (gpa == pa) return pa;
        atomicMin(&L[a], gpa);
        a = gpa;
    }
}

// Parents only ever decrease; a stale read yields an older ancestor and the loop
// repairs any link it displaces (old != a -> keep uniting old with b).
__device__ __forceinline__ void cc_unite(int32_t* L, int a, int b) {
    while (true) {
        a = cc_find_halve(L, a);
        b = cc_find_halve(L, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(&L[a], b);
        if (old == a) return;
        a = old;
    }
}

__global__ __launch_bounds__(256) void k_cc_init(const int64_t* __restrict__ cm, int W, int HW, long long total,
                                                 int32_t* __restrict__ L) {
    long long g0 = (long long)blockIdx.x * kCcBlock;
    int lane = threadIdx.x & (kWave - 1);
    const unsigned p0 = (unsigned)(g0 % HW);               // uniform: the 64-bit division runs once, on the scalar unit
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        long long g = g0 + it * 256 + threadIdx.x;
        bool in = g < total;
        unsigned pp = p0 + it * 256 + threadIdx.x;
        if (pp >= (unsigned)HW) pp %= (unsigned)HW;        // the block crosses into the next image (rare lanes)
        int p = in ? (int)pp : 0;
        int x = (int)((unsigned)p % (unsigned)W);
        bool fg = in && cm[g] != 0;
        unsigned long long m = __ballot(fg);
        bool left_in_wave = lane > 0 && ((m >> (lane - 1)) & 1ull);
        bool is_start = fg && (lane == 0 || x == 0 || !left_in_wave);
        unsigned long long s = __ballot(is_start);
        if (!in) continue;
        int parent = -1;
        if (fg) {
            unsigned long long below = s & ((2ull << lane) - 1ull);
            int start_lane = 63 - __clzll(below);
            parent = (int)(g - lane + start_lane);
            // a run that begins at lane 0 may continue the previous wave segment's run
            if (start_lane == lane && lane == 0 && x > 0 && cm[g - 1] != 0) parent = (int)(g - 1);
        }
        L[g] = parent;
    }
}

__global__ __launch_bounds__(256) void k_cc_merge(int W, int HW, long long total, int32_t* __restrict__ L) {
    long long g0 = (long long)blockIdx.x * kCcBlock;
    const unsigned p0 = (unsigned)(g0 % HW);               // uniform
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        long long g = g0 + it * 256 + threadIdx.x;
        if (g >= total) continue;
        unsigned pp = p0 + it * 256 + threadIdx.x;
        if (pp >= (unsigned)HW) pp %= (unsigned)HW;
        int p = (int)pp;
        if (p < W) continue;  // first row of its image
        if (L[g] < 0 || L[g - W] < 0) continue;
        int x = (int)((unsigned)p % (unsigned)W);
        // (p, p-W) is implied by (p-1, p-1-W) when both left neighbours are foreground
        if (x > 0 && L[g - 1] >= 0 && L[g - 1 - W] >= 0) continue;
        cc_unite(L, (int)g, (int)(g - W));
    }
}

// Root of every pixel + root census.  Only pixels whose parent lies OUTSIDE their 64-pixel wave segment
// (run starts that were united with another row / segment) walk the union-find tree; a pixel whose
// parent is a lower lane of its own wave — every non-start pixel of a run, k_cc_init — takes that
// lane's root by shuffle.  This removes ~60x of the finds and all of their atomic contention.
__global__ __launch_bounds__(256) void k_cc_flatten(long long total, int32_t* __restrict__ L,
                                                    int32_t* __restrict__ R, int32_t* __restrict__ blk_cnt) {
    __shared__ int scratch[4];
    long long g0 = (long long)blockIdx.x * kCcBlock;
    int lane = threadIdx.x & (kWave - 1);
    int roots = 0;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        long long g = g0 + it * 256 + threadIdx.x;
        bool in = g < total;
        int p = in ? L[g] : -1;
        bool fg = p >= 0;
        long long seg0 = g - lane;
        bool inseg = fg && p >= seg0 && p < g;
        int r = -1;
        if (fg && !inseg) r = cc_find_halve(L, (int)g);
        for (int k = 0; k < 8; ++k) {      // in-segment chains are one link deep unless runs of one row were united
            int pr = __shfl(r, inseg ? (int)(p - seg0) : lane, kWave);
            if (inseg && r < 0 && pr >= 0) r = pr;
            if (!__any(inseg && r < 0)) break;
        }
        if (fg && r < 0) r = cc_find(L, (int)g);
        if (in) {
            roots += (r == (int)g);
            R[g] = r;
        }
    }
    int tot = block_sum_bcast(roots, scratch);
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void k_cc_scan(int nb, const int32_t* __restrict__ blk_cnt,
                                                  int32_t* __restrict__ blk_off, int32_t* __restrict__ n_out) {
    __shared__ int s_part[1024];
    int per = (nb + 1023) / 1024;
    int lo = threadIdx.x * per, hi = min(nb, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += blk_cnt[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (int o = 1; o < 1024; o <<= 1) {
        int v = threadIdx.x >= o ? s_part[threadIdx.x - o] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = s_part[threadIdx.x] - sum;
    for (int i = lo; i < hi; ++i) {
        blk_off[i] = run;
        run += blk_cnt[i];
    }
    if (threadIdx.x == 1023) *n_out = s_part[1023];
}

// Thread t of a block owns pixels g0 + it*256 + t; raster order inside the block is
// (it, t), so the block-local exclusive scan runs over it-major order.
__global__ __launch_bounds__(256) void k_cc_rank(long long total, const int32_t* __restrict__ R,
                                                 const int32_t* __restrict__ blk_off, int32_t* __restrict__ rank,
                                                 int32_t* __restrict__ root_pix, int cap) {
    __shared__ int s_wave[4][4];
    long long g0 = (long long)blockIdx.x * kCcBlock;
    int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    bool isr[4];
    int pre[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        long long g = g0 + it * 256 + threadIdx.x;
        isr[it] = g < total && R[g] == (int)g;
        unsigned long long m = __ballot(isr[it]);
        pre[it] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[it][w] = __popcll(m);
    }
    __syncthreads();
    int base = blk_off[blockIdx.x];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        int off = base;
        for (int j = 0; j < it * 4 + w; ++j) off += s_wave[j / 4][j % 4];
        if (isr[it]) {
            long long g = g0 + it * 256 + threadIdx.x;
            int label = off + pre[it] + 1;
            rank[g] = label;
            if (root_pix && label <= cap) root_pix[label - 1] = (int)g;
        }
    }
}

__global__ __launch_bounds__(256) void k_cc_relabel(long long total, const int32_t* __restrict__ R,
                                                    const int32_t* __restrict__ rank, int32_t* __restrict__ labels) {
    long long g0 = (long long)blockIdx.x * kCcBlock;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        long long g = g0 + it * 256 + threadIdx.x;
        if (g >= total) continue;
        int r = R[g];
        labels[g] = r >= 0 ? rank[r] : 0;
    }
}

struct CcWs {
    int32_t *L, *R, *blk_cnt, *blk_off;
    size_t total;
};

static CcWs cc_carve(void* base, int B, int H, int W) {
    CcWs w;
    size_t tot = (size_t)B * H * W;
    int nb = (int)((tot + kCcBlock - 1) / kCcBlock);
    char* p = (char*)base;
    size_t off = 0;
    w.L = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * tot, 256);
    w.R = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * tot, 256);
    w.blk_cnt = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)nb, 256);
    w.blk_off = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)nb, 256);
    w.total = off;
    return w;
}

}  // namespace fpc

using namespace fpc;

extern "C" size_t fpc_cc_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H < 1 || W < 1) return 256;
    return cc_carve(nullptr, B, H, W).total;
}

extern "C" int fpc_cc_label(const int64_t* cat_mask, int B, int H, int W, int32_t* labels, int32_t* n_out,
                            int32_t* root_pix, int cap, void* ws, size_t ws_bytes, fpc_stream_t stream) {
    if (B < 0 || H < 1 || W < 1 || !n_out) return FPC_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) {
        hipError_t e = hipMemsetAsync(n_out, 0, sizeof(int32_t), s);
        if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
        return FPC_OK;
    }
    long long total = (long long)B * H * W;
    if (total >= (1ll << 31)) return FPC_EINVAL;  // linear indices are i32
    if (!cat_mask || !labels || !ws) return FPC_EINVAL;
    if (((uintptr_t)ws & 255) != 0) return FPC_EWORKSPACE;
    CcWs w = cc_carve(ws, B, H, W);
    if (ws_bytes < w.total) return FPC_EWORKSPACE;
    int HW = H * W;
    int nb = (int)((total + kCcBlock - 1) / kCcBlock);
    hipLaunchKernelGGL(k_cc_init, dim3(nb), dim3(256), 0, s, cat_mask, W, HW, total, w.L);
    hipLaunchKernelGGL(k_cc_merge, dim3(nb), dim3(256), 0, s, W, HW, total, w.L);
    hipLaunchKernelGGL(k_cc_flatten, dim3(nb), dim3(256), 0, s, total, w.L, w.R, w.blk_cnt);
    hipLaunchKernelGGL(k_cc_scan, dim3(1), dim3(1024), 0, s, nb, w.blk_cnt, w.blk_off, n_out);
    // rank is written only at root positions; L is dead after k_cc_flatten and is reused for it
    hipLaunchKernelGGL(k_cc_rank, dim3(nb), dim3(256), 0, s, total, w.R, w.blk_off, w.L, root_pix, cap);
    hipLaunchKernelGGL(k_cc_relabel, dim3(nb), dim3(256), 0, s, total, w.R, w.L, labels);
    return check_launch();
}
