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
#include <algorithm>

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
        // a pixel that is still its own parent is its component's root; a tile root that a border union re-pointed, like every
        // other pixel, walks read-only from its tile root (p) — the chains cross tile borders only, and the readers' loads of
        // a shared chain hit in cache (all pixels halving at once: 24 us instead of 9 at 640 x 480)
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

// ---- whole-image labelling from foreground BIT WORDS (round 4) ------------------------------------------------------------
// The foreground of a 640 x 480 frame is 38 KB as bit words (bit j of word w = pixel 64 w + j) — it fits the LDS of ONE
// workgroup, and so does a union-find over its row RUNS (maximal sequences of foreground pixels of a row).  Two launches
// instead of four or five, no per-pixel parent plane in global memory:
//   k_cca_image   one 1024-thread workgroup per image: bit words -> LDS; run starts per word and their raster-order prefix
//                 (a run's id = the number of run starts before it); unions of the runs of adjacent rows, one per maximal
//                 sequence of columns in which both rows are foreground, in log2(H) levels of row pairs, block pairs, ...
//                 (trees a few links deep instead of as deep as a blob is tall); roots ranked in id order =
//                 raster order of each component's first pixel = scipy.ndimage.label's numbering; per run its image-local
//                 label (+ a root flag), per word its id base, per image its component count -> global memory.
//   k_cca_label   per pixel: word -> run id (id base + run starts up to the pixel) -> label + the components of the images
//                 before it (summed here: no scan launch); root pixels record themselves in root_pix; the last block writes N.
// (The band / border variant that preceded the levels is in the git history of this file.)
// The parent array sits in LDS for up to `cap` runs (23 000 at 640 x 480: a segmentation mask has ~1 000) and in global
// memory beyond (speckle noise: slower, same result; its accesses bypass the CU's vector cache, which would serve the
// values it held before another wave's atomic).  Requires W % 64 == 0 (rows start on word boundaries) and bit words +
// id bases within ~100 KB of LDS; every other shape keeps the tile pipeline above.
#ifndef FPC_CCA_THREADS
#define FPC_CCA_THREADS 1024
#endif
constexpr int kCcaThreads = FPC_CCA_THREADS;
#ifdef FPC_STAMP_CC      // diagnostic build: phase times of image 0 into the (then unused) global parent area
#define CCA_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(a.gparent)[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CCA_STAMP(k) do { } while (0)
#endif

struct CcaArgs {
    const unsigned long long* bits; size_t bstride;      // [B][bstride] words, bstride >= nwords
    int B, H, W, wpr, nwords, cap;                        // wpr = W / 64 words per row, nwords = H * wpr; cap = LDS parent entries
    unsigned wpr_inv;                                     // floor(2^32 / wpr) + 1: w / wpr = umulhi(w, wpr_inv) for w < 2^16 words
    int32_t* wbase;        // [B][nwords]  run starts of the image before word w
    int32_t* runlabel;     // [B][rstride] image-local label (1 ..) of run id, bit 31 = the run is its component's first
    int32_t* gparent;      // [B][rstride] parent array of an image with more than cap runs
    int32_t* ncomp;        // [B]
    size_t rstride;
};

// w / wpr without a division (one workgroup labels an image: every instruction costs all of its waves)
__device__ __forceinline__ int cca_div(int w, int wpr, unsigned inv) { return wpr == 1 ? w : (int)__umulhi((unsigned)w, inv); }

struct LdsParent {
    int* a;
    __device__ __forceinline__ int ld(int i) const { return a[i]; }
    __device__ __forceinline__ void st(int i, int v) const { a[i] = v; }
    __device__ __forceinline__ int amin(int i, int v) const { return atomicMin(&a[i], v); }
};
struct GlobalParent {      // past the CU's vector cache: values other waves' atomics have changed must be seen
    int32_t* a;
    __device__ __forceinline__ int ld(int i) const { return __hip_atomic_load(a + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ void st(int i, int v) const { __hip_atomic_store(a + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ __forceinline__ int amin(int i, int v) const { return atomicMin(&a[i], v); }
};

template <typename P>
__device__ __forceinline__ int run_find(const P& p, int a) {       // with path halving; parents only ever decrease
    while (true) {
        const int pa = p.ld(a);
        if (pa == a) return a;
        const int gpa = p.ld(pa);
        if (gpa == pa) return pa;
        p.amin(a, gpa);
        a = gpa;
    }
}
template <typename P>
__device__ __forceinline__ void run_unite(const P& p, int a, int b) {
    while (true) {
        a = run_find(p, a);
        b = run_find(p, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = p.amin(a, b);
        if (old == a) return;
        a = old;
    }
}

// exclusive scan of one value per thread over the 1024-thread workgroup; returns the prefix, `total` = the sum
__device__ __forceinline__ int cca_block_scan(int v, int* s_part /* 17 */, int& total) {
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    int inc = v;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) { const int t = __shfl_up(inc, o, kWave); if (lane >= o) inc += t; }
    __syncthreads();                                    // s_part of a previous scan has been read
    if (lane == kWave - 1) s_part[w] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < kCcaThreads / kWave; ++k) { const int x = s_part[k]; if (k < w) off += x; tot += x; }
    total = tot;
    return off + inc - v;
}

template <typename P>
__device__ __forceinline__ void cca_body(const CcaArgs& a, const P par, const unsigned long long* sb, const int* wb, int R,
                                         int* s_part) {
    const int b = blockIdx.x, wpr = a.wpr;
    for (int r = threadIdx.x; r < R; r += kCcaThreads) par.st(r, r);
    __syncthreads();
    CCA_STAMP(3);
    // unions: word (y, k) against the word above it, one per maximal column sequence in which both rows are foreground.
    // Level g = 1, 4, 16, ... takes the rows y = g q with q % 4 != 0: each joins finished blocks of g rows, three borders of a
    // group of four at once, so a tree grows by at most three links per level and a find stays a handful of steps — with every row united at
    // once a blob's runs formed a chain as long as the blob is tall, and the finds walking it were the kernel (5.6 + 4.4 us
    // of 16.5 at 640 x 480 with 16-row bands and a border pass).  One workgroup: every instruction costs 16 waves, hence
    // no division in the loops (umulhi by the reciprocal of the row's word count).
    for (int g = 1; g < a.H; g *= 4) {                          // level: rows y = g q, q % 4 != 0 (radix 4: five levels at H = 480)
        const int nq = (a.H - 1) / g;                             // q = 1 .. nq
        const int nrows = nq - nq / 4;
        for (int it = threadIdx.x; it < nrows * wpr; it += kCcaThreads) {
            const int j = cca_div(it, wpr, a.wpr_inv), k = it - j * wpr;
            const int j3 = (int)__umulhi((unsigned)j, 0xAAAAAAABu) >> 1;       // j / 3
            const int w = g * (4 * j3 + (j - 3 * j3) + 1) * wpr + k;
            const unsigned long long m = sb[w], up = sb[w - wpr];
            const unsigned long long ov = m & up;
            if (!ov) continue;
            const unsigned long long ml = k ? sb[w - 1] : 0ull, ul = k ? sb[w - wpr - 1] : 0ull;
            unsigned long long os = ov & ~((ov << 1) | ((ml & ul) >> 63));
            if (!os) continue;
            const unsigned long long st = m & ~((m << 1) | (ml >> 63)), su = up & ~((up << 1) | (ul >> 63));
            const int bm = wb[w], bu = wb[w - wpr];
            while (os) {
                const int x = __ffsll((long long)os) - 1;
                os &= os - 1ull;
                const unsigned long long upto = (2ull << x) - 1ull;
                run_unite(par, bm + __popcll(st & upto) - 1, bu + __popcll(su & upto) - 1);
            }
        }
        __syncthreads();
    }
    CCA_STAMP(5);
    // every run -> its root (no union is in flight any more: the value written is the root)
    for (int r = threadIdx.x; r < R; r += kCcaThreads) par.st(r, run_find(par, r));
    __syncthreads();
    CCA_STAMP(6);
    // roots ranked in id order; a root's entry becomes -(label), a run's label = -(entry of its root)
    const int per = (R + kCcaThreads - 1) / kCcaThreads;
    const int r_lo = min(R, (int)threadIdx.x * per), r_hi = min(R, r_lo + per);
    int cnt = 0;
    for (int r = r_lo; r < r_hi; ++r) cnt += par.ld(r) == r ? 1 : 0;
    int total;
    int rank = cca_block_scan(cnt, s_part, total);
    __syncthreads();                                    // every root test above has been made
    for (int r = r_lo; r < r_hi; ++r)
        if (par.ld(r) == r) par.st(r, -(++rank));
    __syncthreads();
    CCA_STAMP(7);
    int32_t* rl = a.runlabel + (size_t)b * a.rstride;
    for (int r = threadIdx.x; r < R; r += kCcaThreads) {
        const int v = par.ld(r);
        rl[r] = v < 0 ? (-v) | (int)0x80000000 : -par.ld(v);
    }
    if (threadIdx.x == 0) a.ncomp[b] = total;
    CCA_STAMP(8);
}

__global__ __launch_bounds__(kCcaThreads) void k_cca_image(const CcaArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_raw[];      // bits [nwords] | id bases [nwords] | parents [cap]
    __shared__ int s_part[kCcaThreads / kWave + 1];
    const int b = blockIdx.x, nwords = a.nwords, wpr = a.wpr;
    unsigned long long* sb = s_raw;
    int* wb = reinterpret_cast<int*>(sb + nwords);
    int* pl = wb + nwords;
    const unsigned long long* gb = a.bits + (size_t)b * a.bstride;
    CCA_STAMP(0);
    for (int w = threadIdx.x; w < nwords; w += kCcaThreads) sb[w] = gb[w];
    __syncthreads();
    CCA_STAMP(1);
    // run starts of this thread's words (a contiguous range), their prefix over the image
    const int per = (nwords + kCcaThreads - 1) / kCcaThreads;
    const int w_lo = min(nwords, (int)threadIdx.x * per), w_hi = min(nwords, w_lo + per);
    const int k_lo = w_lo - cca_div(w_lo, wpr, a.wpr_inv) * wpr;          // position of the first word in its row
    int cnt = 0;
    {
        int k = k_lo;
        for (int w = w_lo; w < w_hi; ++w) {
            const unsigned long long m = sb[w], left = k ? sb[w - 1] >> 63 : 0ull;
            cnt += __popcll(m & ~((m << 1) | left));
            if (++k == wpr) k = 0;
        }
    }
    int R;
    int base = cca_block_scan(cnt, s_part, R);
    int32_t* gw = a.wbase + (size_t)b * nwords;
    {
        int k = k_lo;
        for (int w = w_lo; w < w_hi; ++w) {
            const unsigned long long m = sb[w], left = k ? sb[w - 1] >> 63 : 0ull;
            wb[w] = base; gw[w] = base;
            base += __popcll(m & ~((m << 1) | left));
            if (++k == wpr) k = 0;
        }
    }
    __syncthreads();
    CCA_STAMP(2);
    if (R <= a.cap) cca_body(a, LdsParent{pl}, sb, wb, R, s_part);                // uniform
    else cca_body(a, GlobalParent{a.gparent + (size_t)b * a.rstride}, sb, wb, R, s_part);
}

// grid (ceil(nwords / 64), B): 4096 pixels per workgroup, four consecutive pixels per thread and sweep
__global__ __launch_bounds__(256) void k_cca_label(const CcaArgs a, int32_t* __restrict__ labels, int32_t* __restrict__ n_out,
                                                   int32_t* __restrict__ root_pix, int cap_roots) {
    __shared__ int s_sum[4];
    const int b = blockIdx.y, nwords = a.nwords, wpr = a.wpr;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    // components of the images before this one
    int part = 0;
    for (int i = threadIdx.x; i < b; i += 256) part += a.ncomp[i];
    part = wave_reduce_add(part);
    if (lane == 0) s_sum[wv] = part;
    __syncthreads();
    const int off = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    if (b == a.B - 1 && blockIdx.x == 0 && threadIdx.x == 0) *n_out = off + a.ncomp[b];
    const unsigned long long* gb = a.bits + (size_t)b * a.bstride;
    const int32_t* gw = a.wbase + (size_t)b * nwords;
    const int32_t* rl = a.runlabel + (size_t)b * a.rstride;
    const size_t HW = (size_t)nwords * 64;
    // Four sweeps, phase by phase: every phase's loads (bit word; left neighbour + id base; run labels) are independent of
    // each other across the sweeps, so a thread has up to 4 / 8 / 16 of them in flight instead of walking twelve dependent
    // round trips sweep by sweep.
    int p0[4], wq[4], base[4];
    unsigned long long m[4], stt[4];
    bool fg[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        p0[it] = blockIdx.x * 4096 + it * 1024 + threadIdx.x * 4;
        wq[it] = p0[it] >> 6;
        m[it] = wq[it] < nwords ? gb[wq[it]] : 0ull;
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int w = wq[it], x0 = p0[it] & 63;
        fg[it] = w < nwords && ((m[it] >> x0) & 15ull);
        unsigned long long left = 0ull;
        base[it] = 0;
        if (fg[it]) {
            left = (w - cca_div(w, wpr, a.wpr_inv) * wpr) ? gb[w - 1] >> 63 : 0ull;
            base[it] = gw[w];
        }
        stt[it] = m[it] & ~((m[it] << 1) | left);
    }
    int v[4][4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int x0 = p0[it] & 63;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + j;
            v[it][j] = 0;
            if (fg[it] && ((m[it] >> x) & 1ull)) v[it][j] = rl[base[it] + __popcll(stt[it] & ((2ull << x) - 1ull)) - 1];
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (wq[it] >= nwords) continue;
        const int x0 = p0[it] & 63;
        int lab[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + j;
            lab[j] = 0;
            if (fg[it] && ((m[it] >> x) & 1ull)) {
                lab[j] = off + (v[it][j] & 0x7fffffff);
                if (v[it][j] < 0 && ((stt[it] >> x) & 1ull) && root_pix && lab[j] <= cap_roots) root_pix[lab[j] - 1] = (int)((size_t)b * HW + p0[it] + j);
            }
        }
        *reinterpret_cast<int4*>(labels + (size_t)b * HW + p0[it]) = make_int4(lab[0], lab[1], lab[2], lab[3]);
    }
}

// foreground bit words of an i64 class mask (the drop-in entry's first step): one word per wave and sweep
__global__ __launch_bounds__(256) void k_cc_fg_bits(const int64_t* __restrict__ cm, int B, long long HW, size_t bstride,
                                                    unsigned long long* __restrict__ bits) {
    const int lane = threadIdx.x & (kWave - 1);
    const long long wpi = (long long)bstride;                    // words per image incl. padding
    const long long total = (long long)B * wpi;
    for (long long w = (long long)blockIdx.x * 4 + threadIdx.x / kWave; w < total; w += (long long)gridDim.x * 4) {
        const long long b = w / wpi, wi = w - b * wpi;
        const long long p = wi * 64 + lane;
        const bool fg = p < HW && cm[b * HW + p] != 0;
        const unsigned long long m = __ballot(fg);
        if (lane == 0) bits[w] = m;
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

// ---- host side of the bit-word path
constexpr size_t kCcaLdsBytes = 152 * 1024;      // of the CU's 160 KB: one 1024-thread workgroup per CU

static bool cca_eligible(int B, int H, int W) {
    if (W % 64 != 0 || B > 65535) return false;
    const long long nwords = (long long)H * (W / 64);
    return nwords * 12 + 4096 <= (long long)kCcaLdsBytes - 32 * 1024;      // bit words + id bases, and at least 8 000 parents
}

struct CcaWs {
    unsigned long long* bits;
    int32_t *wbase, *runlabel, *gparent, *ncomp;
    size_t bstride, rstride, total;
};

static CcaWs cca_carve(void* base, int B, int H, int W) {
    CcaWs w;
    const size_t HW = (size_t)H * W, nwords = HW / 64;
    w.bstride = (size_t)cdiv((int)HW, 4096) * 64;              // fpc_mask_bits_words(H, W)
    w.rstride = align_up(HW / 2 + 1, 64);
    char* p = (char*)base;
    size_t off = 0;
    w.bits = (unsigned long long*)(p + off); off = align_up(off + sizeof(unsigned long long) * (size_t)B * w.bstride, 256);
    w.wbase = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)B * nwords, 256);
    w.runlabel = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)B * w.rstride, 256);
    w.gparent = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)B * w.rstride, 256);
    w.ncomp = (int32_t*)(p + off); off = align_up(off + sizeof(int32_t) * (size_t)B, 256);
    w.total = off;
    return w;
}

static int cca_run(const unsigned long long* bits, size_t bstride, int B, int H, int W, int32_t* labels, int32_t* n_out,
                   int32_t* root_pix, int cap, const CcaWs& w, hipStream_t s) {
    static bool attr_set = false;      // (idempotent; a race sets it twice)
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_cca_image), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCcaLdsBytes);
        if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
        attr_set = true;
    }
    CcaArgs a{};
    a.bits = bits; a.bstride = bstride; a.B = B; a.H = H; a.W = W; a.wpr = W / 64; a.nwords = H * a.wpr;
    a.wpr_inv = a.wpr > 1 ? (unsigned)((1ull << 32) / (unsigned)a.wpr + 1) : 0u;       // exact quotients below 2^16 (cca_eligible: nwords < 10 000); wpr = 1: see cca_div
    a.wbase = w.wbase; a.runlabel = w.runlabel; a.gparent = w.gparent; a.ncomp = w.ncomp; a.rstride = w.rstride;
    const size_t fixed = (size_t)a.nwords * 12;
#ifndef FPC_CCA_PARENTS
#define FPC_CCA_PARENTS 0
#endif
    a.cap = (int)std::min<size_t>((kCcaLdsBytes - 1024 - fixed) / 4, w.rstride);
    if (FPC_CCA_PARENTS > 0) a.cap = std::min(a.cap, FPC_CCA_PARENTS);
    const size_t lds = fixed + (size_t)a.cap * 4;
    hipLaunchKernelGGL(k_cca_image, dim3(B), dim3(kCcaThreads), lds, s, a);
    hipLaunchKernelGGL(k_cca_label, dim3(cdiv(a.nwords, 64), B), dim3(256), 0, s, a, labels, n_out, root_pix, cap);
    return check_launch();
}

extern "C" size_t fpc_cc_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H < 1 || W < 1) return 256;
    const size_t tile = cc_carve(nullptr, B, H, W).total;
    return cca_eligible(B, H, W) ? std::max(tile, cca_carve(nullptr, B, H, W).total) : tile;
}

extern "C" int fpc_fg_bits(const int64_t* cat_mask, int B, int H, int W, uint64_t* fg_bits, fpc_stream_t stream) {
    if (B < 0 || H < 1 || W < 1) return FPC_EINVAL;
    if (B == 0) return FPC_OK;
    if (!cat_mask || !fg_bits || ((uintptr_t)fg_bits & 7)) return FPC_EINVAL;
    const size_t bstride = (size_t)cdiv(H * W, 4096) * 64;
    const long long words = (long long)B * bstride;
    hipLaunchKernelGGL(k_cc_fg_bits, dim3((unsigned)std::min<long long>((words + 3) / 4, 4096)), dim3(256), 0, (hipStream_t)stream, cat_mask, B,
                       (long long)H * W, bstride, reinterpret_cast<unsigned long long*>(fg_bits));
    return check_launch();
}

extern "C" int fpc_cc_label_bits(const uint64_t* fg_bits, int B, int H, int W, int32_t* labels, int32_t* n_out,
                                 int32_t* root_pix, int cap, void* ws, size_t ws_bytes, fpc_stream_t stream) {
    if (B < 0 || H < 1 || W < 1 || !n_out) return FPC_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) {
        hipError_t e = hipMemsetAsync(n_out, 0, sizeof(int32_t), s);
        if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
        return FPC_OK;
    }
    if ((long long)B * H * W >= (1ll << 31)) return FPC_EINVAL;
    if (!fg_bits || !labels || !ws || ((uintptr_t)fg_bits & 7) || ((uintptr_t)labels & 15)) return FPC_EINVAL;
    if (!cca_eligible(B, H, W)) return FPC_EINVAL;             // callers ask fpc_cc_bits_supported first
    if (((uintptr_t)ws & 255) != 0) return FPC_EWORKSPACE;
    CcaWs w = cca_carve(ws, B, H, W);
    if (ws_bytes < w.total) return FPC_EWORKSPACE;
    return cca_run(reinterpret_cast<const unsigned long long*>(fg_bits), w.bstride, B, H, W, labels, n_out, root_pix, cap, w, s);
}

extern "C" int fpc_cc_bits_supported(int B, int H, int W) { return (B >= 0 && H >= 1 && W >= 1 && cca_eligible(B, H, W)) ? 1 : 0; }

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
    if (cca_eligible(B, H, W) && ((uintptr_t)labels & 15) == 0) {
        // rows on word boundaries and an image whose bit words fit one workgroup's LDS: i64 mask -> bit words, then the
        // whole-image labelling (three launches; a caller that already holds the words calls fpc_cc_label_bits: two)
        CcaWs cw = cca_carve(ws, B, H, W);
        if (ws_bytes < cw.total) return FPC_EWORKSPACE;
        const long long words = (long long)B * cw.bstride;
        hipLaunchKernelGGL(k_cc_fg_bits, dim3((unsigned)std::min<long long>((words + 3) / 4, 4096)), dim3(256), 0, s, cat_mask, B, (long long)H * W,
                           cw.bstride, cw.bits);
        return cca_run(cw.bits, cw.bstride, B, H, W, labels, n_out, root_pix, cap, cw, s);
    }
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
