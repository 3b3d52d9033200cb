// cc.hip — connected-component labelling of (cat_mask != 0), replacing the
// torch -> cupy -> cupyx.scipy.ndimage.label -> DLPack round trip of
// AggregationLayer.batchwise_break_segmentation_mask (F/lib/aggregation_layer.py:160-183).
//
// 4-connectivity inside an image, none across images (the 3x3x3 structuring element
// of :43-59).  Lock-free union-find on global linear pixel indices:
//   k_cc_tile     64 x 16 pixel tiles: row runs (ballot + clz) and the tile's vertical unions in LDS; every pixel
//                 points at the global index of its tile-local root
//   k_cc_border   unions across tile borders, only where a run start / run break makes one necessary
//   k_cc_flatten  root of every pixel; per 64-pixel segment which pixels are roots (bit word) and how many roots its
//                 1024-pixel block holds before it; root census per block
//   k_cc_scan     exclusive scan of the block census (one block) -> N  (only beyond 1024 blocks: k_cc_label scans a short
//                 census itself)
//   k_cc_label    labels[p] = 1 + roots before root(p) in raster order (scipy's numbering, continuing across the batch)
// The root of a component is its minimum linear index, i.e. its first pixel in raster
// order, so ranking the roots by index reproduces scipy.ndimage.label's order exactly.
// (Round 3 tried flatten + scan + rank as ONE launch — blocks in ticket order, decoupled look-back over 64 predecessors per
// step: 43.4 us against 45.6 us for the six launches at B = 1, 225 us against 181 us at B = 32, where one ticket word
// serves ~88 draws / us and batching the draws serialises the tree walks; not kept.  What was kept is the tile-local first
// stage below: 60 -> 35 us at 640 x 480, 178 -> 119 us for 32 frames.)
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

// ---- tile-local labelling ----------------------------------------------------------------
// One workgroup = a 64 x 16 pixel tile (a wave per row, four rows each).  Row runs by ballot + clz, the vertical unions of
// the tile in LDS (the same lock-free union-find, trees at most 16 deep, ~100-cycle links), then every pixel's parent is
// written as the GLOBAL linear index of its tile-local root: the first pixel of its tile component in raster order, which is
// also the smallest global index of that component.  What is left for global memory are the unions ACROSS tile borders
// (k_cc_border): trees as deep as a blob spans tiles, not as it is tall in rows.  (Round 3: the row-run init + all-rows
// merge walked chains of up to 150 dependent global loads per run start: 25 + 14 us at 640 x 480.)
constexpr int kTileW = 64, kTileH = 16;

__device__ __forceinline__ int lds_find_halve(int* l, int a) {
    while (true) {
        int pa = l[a];
        if (pa == a) return a;
        int gpa = l[pa];
        if (gpa == pa) return pa;
        atomicMin(&l[a], gpa);
        a = gpa;
    }
}
__device__ __forceinline__ void lds_unite(int* l, int a, int b) {
    while (true) {
        a = lds_find_halve(l, a);
        b = lds_find_halve(l, b);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; }
        int old = atomicMin(&l[a], b);
        if (old == a) return;
        a = old;
    }
}

// grid (ceil(W / 64), ceil(H / 16), B)
__global__ __launch_bounds__(256) void k_cc_tile(const int64_t* __restrict__ cm, int H, int W, int32_t* __restrict__ L) {
    __shared__ int l[kTileH * kTileW];          // parent as a tile-local index (row * 64 + column), -1 = background
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    const int x = x0 + lane;
    const size_t img = (size_t)blockIdx.z * H * W;
    bool fg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ly = w + 4 * r, y = y0 + ly;
        const bool in = x < W && y < H;
        fg[r] = in && cm[img + (size_t)y * W + x] != 0;
        const unsigned long long m = __ballot(fg[r]);
        const bool left = lane > 0 && ((m >> (lane - 1)) & 1ull);
        const unsigned long long s = __ballot(fg[r] && !left);
        int parent = -1;
        if (fg[r]) parent = ly * kTileW + 63 - __clzll(s & ((2ull << lane) - 1ull));
        l[ly * kTileW + lane] = parent;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ly = w + 4 * r;
        if (ly == 0 || !fg[r] || l[(ly - 1) * kTileW + lane] < 0) continue;
        // (p, p - row) is implied by the pair to the left when both left neighbours are foreground
        if (lane > 0 && l[ly * kTileW + lane - 1] >= 0 && l[(ly - 1) * kTileW + lane - 1] >= 0) continue;
        lds_unite(l, ly * kTileW + lane, (ly - 1) * kTileW + lane);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ly = w + 4 * r, y = y0 + ly;
        if (x >= W || y >= H) continue;
        int parent = -1;
        if (fg[r]) {
            const int root = lds_find_halve(l, ly * kTileW + lane);
            parent = (int)(img + (size_t)(y0 + (root >> 6)) * W + x0 + (root & 63));
        }
        L[img + (size_t)y * W + x] = parent;
    }
}

// Unions across tile borders: one thread per pixel of a tile's first row (against the pixel above) or first column (against
// the pixel to its left).  A pair is skipped when the neighbouring pair one step before it along the border is foreground
// too (all four pixels are then connected through unions that somebody performs).
__global__ __launch_bounds__(256) void k_cc_border(int B, int H, int W, int32_t* __restrict__ L) {
    const int nyb = (H - 1) / kTileH, nxb = (W - 1) / kTileW;      // border rows / columns per image
    const int per_img = nyb * W + nxb * H;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)B * per_img) return;
    const int b = (int)(t / per_img), k = (int)(t - (long long)b * per_img);
    const size_t img = (size_t)b * H * W;
    if (k < nyb * W) {                       // horizontal border: pixel (x, y) with y a multiple of 16, against (x, y - 1)
        const int y = (k / W + 1) * kTileH, x = k - (k / W) * W;
        const size_t g = img + (size_t)y * W + x;
        if (L[g] < 0 || L[g - W] < 0) return;
        if (x > 0 && L[g - 1] >= 0 && L[g - 1 - W] >= 0) return;
        cc_unite(L, (int)g, (int)(g - W));
    } else {                                 // vertical border: pixel (x, y) with x a multiple of 64, against (x - 1, y)
        const int k2 = k - nyb * W;
        const int x = (k2 / H + 1) * kTileW, y = k2 - (k2 / H) * H;
        const size_t g = img + (size_t)y * W + x;
        if (L[g] < 0 || L[g - 1] < 0) return;
        // implied by the pair one row up only when both vertical links are tile-LOCAL (not on a tile's first row: there the
        // link (x, y) ~ (x, y - 1) is a horizontal-border pair whose own skip rule may lean on this very union)
        if (y % kTileH != 0 && L[g - W] >= 0 && L[g - 1 - W] >= 0) return;
        cc_unite(L, (int)g, (int)(g - 1));
    }
}

// Root of every pixel + root census.  Only pixels whose parent lies OUTSIDE their 64-pixel wave segment
// (run starts that were united with another row / segment) walk the union-find tree; a pixel whose
// parent is a lower lane of its own wave — every non-start pixel of a run, k_cc_init — takes that
// lane's root by shuffle.  This removes ~60x of the finds and all of their atomic contention.
__global__ __launch_bounds__(256) void k_cc_flatten(long long total, int32_t* __restrict__ L,
                                                    int32_t* __restrict__ R, int32_t* __restrict__ blk_cnt,
                                                    unsigned long long* __restrict__ seg_bits, int32_t* __restrict__ seg_pre) {
    __shared__ int s_seg[16];
    long long g0 = (long long)blockIdx.x * kCcBlock;
    int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        long long g = g0 + it * 256 + threadIdx.x;
        bool in = g < total;
        int p = in ? L[g] : -1;
        bool fg = p >= 0;
        long long seg0 = g - lane;
        bool inseg = fg && p >= seg0 && p < g;
        int r = -1;
        // a tile root (its own parent after k_cc_tile, unless a border union re-pointed it) walks with path halving; every
        // other pixel starts at its tile root and only reads: the few writers shorten the chains the many readers follow,
        // and the readers' loads of a shared chain hit in cache (all pixels halving at once: 24 us instead of 9 at 640 x 480)
        if (fg && !inseg) r = (p == (int)g || L[p] == p) ? (p == (int)g ? cc_find_halve(L, (int)g) : p) : cc_find(L, p);
        for (int k = 0; k < 8; ++k) {      // in-segment chains are one link deep unless runs of one row were united
            int pr = __shfl(r, inseg ? (int)(p - seg0) : lane, kWave);
            if (inseg && r < 0 && pr >= 0) r = pr;
            if (!__any(inseg && r < 0)) break;
        }
        if (fg && r < 0) r = cc_find(L, (int)g);
        if (in) R[g] = r;
        // the block's 16 segments of 64 pixels (raster order: segment it * 4 + w): which of them are roots
        const unsigned long long m = __ballot(in && r == (int)g);
        if (lane == 0) {
            seg_bits[(size_t)blockIdx.x * 16 + it * 4 + w] = m;
            s_seg[it * 4 + w] = __popcll(m);
        }
    }
    __syncthreads();
    if (threadIdx.x < 16) {                 // roots of the block before each segment; the block's census
        int pre = 0;
        for (int j = 0; j < (int)threadIdx.x; ++j) pre += s_seg[j];
        seg_pre[(size_t)blockIdx.x * 16 + threadIdx.x] = pre;
        if (threadIdx.x == 15) blk_cnt[blockIdx.x] = pre + s_seg[15];
    }
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

// labels[p] = 1 + number of roots before root(p) in raster order (scipy's numbering, continuing across the batch)
//           = roots in the blocks before the root's + roots of its block before its segment + roots of its segment before it,
// three small reads per pixel (block prefix, k_cc_flatten's segment prefix and segment bit word; the pixels of a component
// share them).  blk_off == nullptr (few blocks: one or a few frames): every workgroup scans the block census itself in LDS
// and the last one writes the component count — no scan launch.  The root pixels also record themselves in root_pix.
// (Until round 3 this was two launches: ranks written at root positions, then gathered.)
__global__ __launch_bounds__(256) void k_cc_label(long long total, int nb, const int32_t* __restrict__ R,
                                                  const int32_t* __restrict__ blk_off, const int32_t* __restrict__ blk_cnt,
                                                  const unsigned long long* __restrict__ seg_bits,
                                                  const int32_t* __restrict__ seg_pre, int32_t* __restrict__ n_out,
                                                  int32_t* __restrict__ labels, int32_t* __restrict__ root_pix, int cap) {
    __shared__ int s_base[1025];
    __shared__ int s_w[4];
    if (!blk_off) {                          // exclusive scan of blk_cnt[0 .. nb), nb <= 1024: four entries per thread
        const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
        int v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = threadIdx.x * 4 + k; v[k] = i < nb ? blk_cnt[i] : 0; sum += v[k]; }
        int inc = sum;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) { const int t = __shfl_up(inc, o, kWave); if (lane >= o) inc += t; }
        if (lane == kWave - 1) s_w[w] = inc;
        __syncthreads();
        int run = inc - sum;
        for (int k = 0; k < w; ++k) run += s_w[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = threadIdx.x * 4 + k; if (i <= nb) s_base[i] = run; run += v[k]; }
        if (threadIdx.x == 255) s_base[1024] = run;      // nb == 1024: the total has no entry of its own above
        __syncthreads();
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_out = s_base[nb];
    }
    const long long g0 = (long long)blockIdx.x * kCcBlock;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const long long g = g0 + it * 256 + threadIdx.x;
        if (g >= total) continue;
        const int r = R[g];
        int label = 0;
        if (r >= 0) {
            const int rb = r >> 10, rs = (r >> 6) & 15, rl = r & 63;
            const int base = blk_off ? blk_off[rb] : s_base[rb];
            const unsigned long long m = seg_bits[(size_t)rb * 16 + rs];
            label = base + seg_pre[(size_t)rb * 16 + rs] + __popcll(m & ((1ull << rl) - 1ull)) + 1;
            if (r == (int)g && root_pix && label <= cap) root_pix[label - 1] = (int)g;
        }
        labels[g] = label;
    }
}

struct CcWs {
    int32_t *L, *R, *blk_cnt, *blk_off, *seg_pre;
    unsigned long long* seg_bits;
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
    w.seg_bits = (unsigned long long*)(p + off); off = align_up(off + sizeof(unsigned long long) * (size_t)nb * 16, 256);
    w.seg_pre = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)nb * 16, 256);
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
    int nb = (int)((total + kCcBlock - 1) / kCcBlock);
    if (B > 65535 || cdiv(H, kTileH) > 65535) return FPC_EINVAL;
    hipLaunchKernelGGL(k_cc_tile, dim3(cdiv(W, kTileW), cdiv(H, kTileH), B), dim3(256), 0, s, cat_mask, H, W, w.L);
    const long long nborder = (long long)B * (((H - 1) / kTileH) * (long long)W + ((W - 1) / kTileW) * (long long)H);
    if (nborder > 0)
        hipLaunchKernelGGL(k_cc_border, dim3((unsigned)((nborder + 255) / 256)), dim3(256), 0, s, B, H, W, w.L);
    hipLaunchKernelGGL(k_cc_flatten, dim3(nb), dim3(256), 0, s, total, w.L, w.R, w.blk_cnt, w.seg_bits, w.seg_pre);
    const bool fold_scan = nb <= 1024;          // each workgroup then scans <= 4 KB of census itself
    if (!fold_scan) hipLaunchKernelGGL(k_cc_scan, dim3(1), dim3(1024), 0, s, nb, w.blk_cnt, w.blk_off, n_out);
    hipLaunchKernelGGL(k_cc_label, dim3(nb), dim3(256), 0, s, total, nb, w.R, fold_scan ? nullptr : w.blk_off, w.blk_cnt, w.seg_bits,
                       w.seg_pre, n_out, labels, root_pix, cap);
    return check_launch();
}
