// merge_split.hip — the FPN merge + 1x1 head in two passes (round 5); k_merge_head (one pass) stays selectable: FPC_MERGE_SPLIT=0.
//
// smp: merged = up2(r5) + up2(r4) + up2(r3) + r2 with r_k = relu(gn(t_k)) (the three upsampled branches share one resolution),
// logits = bias + W merged (Dropout2d is the identity in eval mode; F/lib/pose_regressor.py:709-743, lib/backbone.py).  The head and
// the x2 bilinear upsample are both linear, so
//     logits = bias + up2( W (r5 + r4 + r3) ) + W r2 :
// pass LOW sums the three branches at THEIR resolution and applies the head there (32 of 128 channels leave the kernel: 2.4 MB per
// frame), pass HI applies the head to r2 and adds the x2 upsample of pass LOW's 32-channel result — 4 taps of 128 bytes per pixel
// instead of 12 taps of 512 bytes, one GroupNorm + ReLU per tap value instead of four to sixteen (985 -> 572 us on 32 frames, 32.5 ->
// 27 us on one).  The result differs from the one-pass form by
// rounding only (a different association of the same sum: ~1e-7 of the logits; the engine's 1e-4 bar is unchanged).
// Both passes: one workgroup = 32 pixels x 128 channels; GroupNorm + ReLU'd sum -> LDS; head on the f32 matrix cores (32 px x 32
// ch tile, K split over the four waves, partials summed in wave order) as in k_merge_head.
#include "net_kernels.hpp"

namespace fpc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kHpPx = 32, kHpC = 128, kHpCh = 32;

// NSUB: 32-pixel tiles per workgroup (consecutive pixels).  The head weights reach LDS once per workgroup and the next tile's
// values are requested before the current tile's head runs: with one tile per workgroup pass HI — 16 KB of input per 16 KB of
// weights staged, three barriers, 1024 matrix cycles — ran at 2.7 TB/s where pass LOW (three maps per tile) reaches 5.
template <bool HI, int NSUB>
__global__ __launch_bounds__(256) void k_head_part(const HeadPartArgs a) {
    __shared__ __attribute__((aligned(16))) float s_m[kHpPx][kHpC + 4];
    __shared__ __attribute__((aligned(16))) float s_w[kHpCh][kHpC + 4];
    const int z = blockIdx.z, b = blockIdx.y, tid = threadIdx.x;
    const int HW = a.H * a.W, C = kHpC;
    const int ch = a.ch[z], chp = a.chp[z];
    const int c4 = tid & 31, prow = tid >> 5;               // channel quad, first pixel slot (of 8 per sweep)
    for (int r = prow; r < kHpCh; r += 8)                   // head rows past ch: zero (the MFMA tile is 32 wide)
        *reinterpret_cast<f32x4*>(&s_w[r][4 * c4]) =
            r < ch ? *reinterpret_cast<const f32x4*>(a.hw[z] + (size_t)r * C + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int NM = HI ? 1 : 3;
    f32x4 sa[NM], sb[NM];
#pragma unroll
    for (int k = 0; k < NM; ++k) {                          // affine [B][C][2] interleaved (a, b)
        const float* aff = a.aff[z][k] + ((size_t)b * C + 4 * c4) * 2;
        const f32x4 u = *reinterpret_cast<const f32x4*>(aff), v = *reinterpret_cast<const f32x4*>(aff + 4);
        sa[k] = f32x4{u[0], u[2], v[0], v[2]};
        sb[k] = f32x4{u[1], u[3], v[1], v[3]};
    }
    const int ok = tid & 31;
    const float bias = (HI && ok < ch) ? a.hb[z][ok] : 0.f;
    const float sy = a.H > 1 ? (float)(a.hl - 1) / (float)(a.H - 1) : 0.f, sx = a.W > 1 ? (float)(a.wl - 1) / (float)(a.W - 1) : 0.f;
    const int lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    float* ob = a.out[z] + (size_t)b * HW * chp;
    // this thread's 4 pixels (slot prow + 8 i) x its channel quad of tile p0: every map's values
    f32x4 raw[4][NM];
    auto fetch = [&](int p0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = min(p0 + prow + 8 * i, HW - 1);
#pragma unroll
            for (int k = 0; k < NM; ++k) raw[i][k] = *reinterpret_cast<const f32x4*>(a.t[z][k] + ((size_t)b * HW + p) * C + 4 * c4);
        }
    };
    const int pbase = blockIdx.x * (kHpPx * NSUB);
    fetch(pbase);
#pragma unroll 1
    for (int sub = 0; sub < NSUB; ++sub) {
        const int p0 = pbase + sub * kHpPx;
        if (p0 >= HW) break;                                // uniform
        // HI: the x2 bilinear upsample (align_corners) of pass LOW's result for the FINAL stage's pixels (thread = pixel slot
        // prow + 8 i, channel tid % 32), requested here and not behind the three barriers in front of its use
        float up[4] = {0.f, 0.f, 0.f, 0.f};
        if (HI && ok < ch) {
            const float* L = a.lsum[z] + (size_t)b * a.hl * a.wl * chp + ok;
            float tap[4][4];
            Lerp lys[4], lxs[4];
            int y = (p0 + prow) / a.W, x = (p0 + prow) - y * a.W;      // one division; the other three pixels are 8 further each
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i) { x += 8; while (x >= a.W) { x -= a.W; ++y; } }
                if (y >= a.H) { y = a.H - 1; x = a.W - 1; }             // past the map (ragged last tile): any valid tap, never stored
                lys[i] = lerp_scaled(y, a.hl, sy); lxs[i] = lerp_scaled(x, a.wl, sx);
                tap[i][0] = L[((size_t)lys[i].i0 * a.wl + lxs[i].i0) * chp]; tap[i][1] = L[((size_t)lys[i].i0 * a.wl + lxs[i].i1) * chp];
                tap[i][2] = L[((size_t)lys[i].i1 * a.wl + lxs[i].i0) * chp]; tap[i][3] = L[((size_t)lys[i].i1 * a.wl + lxs[i].i1) * chp];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                up[i] = lys[i].l0 * (lxs[i].l0 * tap[i][0] + lxs[i].l1 * tap[i][1]) + lys[i].l1 * (lxs[i].l0 * tap[i][2] + lxs[i].l1 * tap[i][3]);
        }
        // ---- GroupNorm + ReLU + sum of the maps -> LDS
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NM; ++k) {
                f32x4 v = raw[i][k] * sa[k] + sb[k];
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                acc += v;
            }
            *reinterpret_cast<f32x4*>(&s_m[prow + 8 * i][4 * c4]) = acc;
        }
        if (sub + 1 < NSUB && p0 + kHpPx < HW) fetch(p0 + kHpPx);      // the next tile's values, under this tile's head
        __syncthreads();
        // ---- head: 32 px x 32 ch, K = 128 split over the four waves (A = summed activations, B = head weights, 16-byte LDS fragments)
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int k0 = wv * 32; k0 < (wv + 1) * 32; k0 += 8) {
            const f32x4 fa = *reinterpret_cast<const f32x4*>(&s_m[li][k0 + 4 * lh]);
            const f32x4 fb = *reinterpret_cast<const f32x4*>(&s_w[li][k0 + 4 * lh]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc, 0, 0, 0);
        }
        __syncthreads();                                        // all fragments read: s_m becomes the partial buffer
        float* part = &s_m[0][0];                               // [4 waves][32 px][33]
        static_assert(kHpPx * (kHpC + 4) >= 4 * 32 * 33, "partial buffer");
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = acc[r];
        __syncthreads();
        // ---- 32 lanes per pixel (channel ok < chp active); HI: + bias + the upsampled LOW result
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pl = prow + 8 * i, p = p0 + pl;
            if (p >= HW || ok >= chp) continue;
            float v = 0.f;
            if (ok < ch) {
#pragma unroll
                for (int w = 0; w < 4; ++w) v += part[(w * 32 + pl) * 33 + ok];
                if (HI) v = (bias + up[i]) + v;
            }
            ob[(size_t)p * chp + ok] = v;
        }
        __syncthreads();                                        // the partial buffer is the next tile's s_m
    }
}

// a.H x a.W: the resolution of THIS pass (LOW: the upsampled branches' own, HI: the merge resolution = 2 x that)
int launch_head_part(const HeadPartArgs& a, bool hi, int groups, hipStream_t s) {
    if (a.C != kHpC || groups < 1 || groups > kMaxGroup || a.B < 1 || a.B > 65535) return FPC_EINVAL;
    for (int z = 0; z < groups; ++z)
        if (a.ch[z] > kHpCh || a.chp[z] > kHpCh || a.chp[z] < a.ch[z] || !a.out[z] || !a.hw[z] || (hi && (!a.lsum[z] || !a.hb[z]))) return FPC_EINVAL;
    if (hi && (a.H != 2 * a.hl || a.W != 2 * a.wl)) return FPC_EINVAL;
    constexpr int kSubHi = 4, kSubLo = 2;
    // few pixels (one frame): one tile per workgroup keeps the grid wide
    const bool wide = (long long)a.B * a.H * a.W * groups >= 256LL * 4 * kHpPx * kSubHi;
    if (hi && wide) hipLaunchKernelGGL((k_head_part<true, kSubHi>), dim3(cdiv(a.H * a.W, kHpPx * kSubHi), a.B, groups), dim3(256), 0, s, a);
    else if (hi) hipLaunchKernelGGL((k_head_part<true, 1>), dim3(cdiv(a.H * a.W, kHpPx), a.B, groups), dim3(256), 0, s, a);
    else if (wide) hipLaunchKernelGGL((k_head_part<false, kSubLo>), dim3(cdiv(a.H * a.W, kHpPx * kSubLo), a.B, groups), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_head_part<false, 1>), dim3(cdiv(a.H * a.W, kHpPx), a.B, groups), dim3(256), 0, s, a);
    return check_launch();
}

}  // namespace fpc
