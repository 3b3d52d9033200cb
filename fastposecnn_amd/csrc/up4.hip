// up4.hip — k_up4_compress7x4: the x4 bilinear upsample of the four heads' low-resolution logits + xyz -> xy / z split + class
// compression (F/lib/pose_regressor.py:729-732, 445-457; F/lib/gpu_tensor_funcs.py:37-99) with four pixels per thread.
// The one-pixel-per-thread forms (any class count, any width) are k_up4_compress / k_up4_compress7 in net_kernels.hip.
#include "net_kernels.hpp"

namespace fpc {

// k_up4_compress7 with FOUR consecutive pixels of a row per thread (W % 4 == 0): the four pixels' x taps lie in at most three
// low-resolution columns (the x scale is < 1/2), so a thread fetches 3 x 2 taps per channel quad instead of 4 x 4 (27 instead of
// 72 16-byte loads per pixel), and every full-resolution plane store is 16 bytes per lane (1 KB per wave instruction: 21 store
// instructions per pixel become 5).  Each pixel's value is computed by the same expression on the same four tap values as in
// k_up4_compress7 (the taps are SELECTED from the three columns, never re-weighted): results are bit-identical.
__global__ __launch_bounds__(128, 3) void k_up4_compress7x4(const Up4Args a) {
    constexpr int C = 7, G = 6;
    const int HW = a.H * a.W;
    // XCD-aware block order.  Workgroups go to the 8 XCDs round-robin by linear id, and every output pixel reads ~100 bytes of
    // low-resolution taps: in plain order each XCD's L2 (4 MB) saw ALL rows of every frame's low-resolution logits (5.5 MB per
    // frame) and re-fetched them through the fabric — about 1.4 GB of reads beside 3.1 GB of writes per 32-frame batch.  When the
    // blocks of a frame divide by 8, XCD x takes the x-th eighth of the frame's rows (0.7 MB of taps): id -> (xcd = id % 8, rest).
    int b = blockIdx.y, blk = blockIdx.x;
    if ((gridDim.x & 7) == 0) {
        const unsigned id = blockIdx.x + gridDim.x * blockIdx.y, per = gridDim.x >> 3, j = id >> 3;
        b = (int)(j / per);
        blk = (int)((id & 7) * per + (j - (unsigned)b * per));
    }
    const int p0 = 4 * (blk * blockDim.x + threadIdx.x);
    if (p0 >= HW) return;                                   // whole waves only: HW % 256 == 0 (launcher)
    const int y = p0 / a.W, x0 = p0 - y * a.W;
    const Lerp ly = lerp_coord(y, a.hl, a.H);
    Lerp lx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) lx[j] = lerp_coord(x0 + j, a.wl, a.W);
    const int cb = lx[0].i0;
    bool s1[4];                                             // pixel j's left tap is column cb + 1 (else cb); its right tap the next one
#pragma unroll
    for (int j = 0; j < 4; ++j) s1[j] = lx[j].i0 != cb;
    // columns cb, cb + 1, cb + 2 clamped to the map: i1 = i0 + (i0 < wl - 1) is then the column after i0 in this list
    const int c1 = min(cb + 1, a.wl - 1), c2 = min(cb + 2, a.wl - 1);
    // tap pixels inside image b as 32-bit indices: every address below is a uniform base (scalar registers) + a 32-bit lane
    // offset — 64-bit per-lane addresses of 108 loads and 80 stores were most of the 318 registers of the first version
    const unsigned r0 = (unsigned)(ly.i0 * a.wl), r1 = (unsigned)(ly.i1 * a.wl);
    const unsigned t_u0 = r0 + cb, t_u1 = r0 + c1, t_u2 = r0 + c2, t_d0 = r1 + cb, t_d1 = r1 + c1, t_d2 = r1 + c2;
    const size_t img_lo = (size_t)b * a.hl * a.wl;
    // one channel quad of the four pixels: taps6 fetches the 3 x 2 taps, lerp4 -> out[j] = the quad of pixel j.  The loops below
    // request quad q + 1 before they work on quad q and end every iteration with a scheduling barrier: without it the compiler
    // hoists all 108 loads to the top (318 registers, one wave per SIMD); with it the kernel fits four waves per SIMD
    struct Taps { f32x4 u0, u1, u2, d0, d1, d2; };
    auto taps6 = [&](const float* L, unsigned stride, int q) {
        const float* Lb = L + img_lo * stride + 4 * q;      // uniform
        Taps t;
        t.u0 = *reinterpret_cast<const f32x4*>(Lb + t_u0 * stride);
        t.u1 = *reinterpret_cast<const f32x4*>(Lb + t_u1 * stride);
        t.u2 = *reinterpret_cast<const f32x4*>(Lb + t_u2 * stride);
        t.d0 = *reinterpret_cast<const f32x4*>(Lb + t_d0 * stride);
        t.d1 = *reinterpret_cast<const f32x4*>(Lb + t_d1 * stride);
        t.d2 = *reinterpret_cast<const f32x4*>(Lb + t_d2 * stride);
        return t;
    };
    auto lerp4 = [&](const Taps& t, f32x4 (&out)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 v00 = s1[j] ? t.u1 : t.u0, v01 = s1[j] ? t.u2 : t.u1, v10 = s1[j] ? t.d1 : t.d0, v11 = s1[j] ? t.d2 : t.d1;
            out[j] = ly.l0 * (lx[j].l0 * v00 + lx[j].l1 * v01) + ly.l1 * (lx[j].l0 * v10 + lx[j].l1 * v11);
        }
    };
    // plane `c` of a [B][planes][H][W] tensor: the four pixels' values of channel e of the quad
    auto put4 = [&](float* base, int planes, int c, const f32x4 (&v)[4], int e, bool stream) {
        f32x4 o = {v[0][e], v[1][e], v[2][e], v[3][e]};
        float* plane = base + ((size_t)b * planes + c) * HW;      // uniform
        f32x4* dst = reinterpret_cast<f32x4*>(plane + (unsigned)p0);
        if (stream) __builtin_nontemporal_store(o, dst);
        else *dst = o;
    };
    // ---- mask logits: class ids (arg-max of the log-softmax, first maximal index on ties: class_compress.hip)
    f32x4 m0[4], m1[4];
    Taps nxt;
    {
        const Taps ta = taps6(a.lm, 8, 0);
        nxt = taps6(a.lm, 8, 1);
        lerp4(ta, m0);
        __builtin_amdgcn_sched_barrier(0);
        const Taps tb = nxt;
        nxt = taps6(a.lq, 24, 0);
        lerp4(tb, m1);
        __builtin_amdgcn_sched_barrier(0);
    }
    int cls[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float vm[C] = {m0[j][0], m0[j][1], m0[j][2], m0[j][3], m1[j][0], m1[j][1], m1[j][2]};
        float mx = vm[0];
#pragma unroll
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, vm[c]);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) sum += expf(vm[c] - mx);
        const float lse = logf(sum);
        float best = (vm[0] - mx) - lse;
        int k = 0;
#pragma unroll
        for (int c = 1; c < C; ++c) {
            const float val = (vm[c] - mx) - lse;
            if (val > best) { best = val; k = c; }
        }
        cls[j] = k;
    }
    {
        long long* cm = a.cat_mask + (size_t)b * HW + p0;
        typedef long long i64x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<i64x2*>(cm) = i64x2{cls[0], cls[1]};
        *reinterpret_cast<i64x2*>(cm + 2) = i64x2{cls[2], cls[3]};
    }
    if (a.fg_bits) {            // 16 lanes = 64 consecutive pixels = one word (the host checked W % 64 == 0)
        const int lane = threadIdx.x & 63;
        const unsigned nib = (cls[0] != 0 ? 1u : 0u) | (cls[1] != 0 ? 2u : 0u) | (cls[2] != 0 ? 4u : 0u) | (cls[3] != 0 ? 8u : 0u);
        unsigned lo = (lane & 8) ? 0u : nib << (4 * (lane & 7)), hi = (lane & 8) ? nib << (4 * (lane & 7)) : 0u;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { lo |= __shfl_xor(lo, o, 16); hi |= __shfl_xor(hi, o, 16); }
        if ((lane & 15) == 0) a.fg_bits[(size_t)b * a.fg_stride + (p0 >> 6)] = (unsigned long long)lo | ((unsigned long long)hi << 32);
    }
    if (a.o_mask) {
#pragma unroll
        for (int e = 0; e < 4; ++e) put4(a.o_mask, C, e, m0, e, true);
#pragma unroll
        for (int e = 0; e < 3; ++e) put4(a.o_mask, C, 4 + e, m1, e, true);
    }
    // (the compiler otherwise SINKS the per-class selects below to the end of the kernel and keeps all 4 x 60 candidate values
    // alive until then — 82 of them in accumulator registers: `keep` pins a selected value where it is made)
    auto keep = [](float& x) { asm volatile("" : "+v"(x)); };
    // ---- the other heads, a channel quad at a time: full-resolution planes out, the selected class's values kept
    float q4[4][4], sc4[4][3], t4[4][3];                    // per pixel: quaternion, scales, xy + z
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        q4[j][0] = q4[j][1] = q4[j][2] = q4[j][3] = 0.f;
        sc4[j][0] = sc4[j][1] = sc4[j][2] = 0.f;
        t4[j][0] = t4[j][1] = t4[j][2] = 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 6; ++q) {                           // quaternion: class q's four channels are quad q
        f32x4 v[4];
        const Taps cur = nxt;
        nxt = q < 5 ? taps6(a.lq, 24, q + 1) : taps6(a.ls, 20, 0);
        lerp4(cur, v);
        if (a.o_quat) {
#pragma unroll
            for (int e = 0; e < 4; ++e) put4(a.o_quat, 4 * G, 4 * q + e, v, e, true);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { if (cls[j] - 1 == q) q4[j][e] = v[j][e]; keep(q4[j][e]); }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {                           // scales: channel 4 q + e = class (4 q + e) / 3, component (4 q + e) % 3
        f32x4 v[4];
        const Taps cur = nxt;
        nxt = q < 4 ? taps6(a.ls, 20, q + 1) : taps6(a.lt, 20, 0);
        lerp4(cur, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ch = 4 * q + e;
            if (ch >= 3 * G) continue;
            if (a.o_scales) put4(a.o_scales, 3 * G, ch, v, e, true);
#pragma unroll
            for (int j = 0; j < 4; ++j) { if (cls[j] - 1 == ch / 3) sc4[j][ch % 3] = v[j][e]; keep(sc4[j][ch % 3]); }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {                           // xyz: class k = channels 3 k (x), 3 k + 1 (y), 3 k + 2 (z)
        f32x4 v[4];
        const Taps cur = nxt;
        if (q < 4) nxt = taps6(a.lt, 20, q + 1);
        lerp4(cur, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ch = 4 * q + e;
            if (ch >= 3 * G) continue;
            const int k = ch / 3, comp = ch % 3;
            if (a.o_xy) {
                if (comp < 2) put4(a.o_xy, 2 * G, 2 * k + comp, v, e, true);
                else put4(a.o_z, G, k, v, e, true);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { if (cls[j] - 1 == k) t4[j][comp] = v[j][e]; keep(t4[j][comp]); }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- categorical planes (read by the aggregation next: ordinary stores)
    f32x4 cq[4], cxy[4], csc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        csc[j] = f32x4{sc4[j][0], sc4[j][1], sc4[j][2], 0.f};
        float nq = sqrtf(q4[j][0] * q4[j][0] + q4[j][1] * q4[j][1] + q4[j][2] * q4[j][2] + q4[j][3] * q4[j][3]);
        if (nq == 0.0f) nq = 1.0f;
        float nv = sqrtf(t4[j][0] * t4[j][0] + t4[j][1] * t4[j][1]);
        if (nv == 0.0f) nv = 1.0f;
        cq[j] = f32x4{q4[j][0] / nq, q4[j][1] / nq, q4[j][2] / nq, q4[j][3] / nq};
        cxy[j] = f32x4{t4[j][0] / nv, t4[j][1] / nv, t4[j][2], 0.f};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) put4(a.cq, 4, e, cq, e, false);
#pragma unroll
    for (int e = 0; e < 3; ++e) put4(a.cs, 3, e, csc, e, false);
#pragma unroll
    for (int e = 0; e < 2; ++e) put4(a.cxy, 2, e, cxy, e, false);
    put4(a.cz, 1, 0, cxy, 2, false);
}

void launch_up4_compress7x4(const Up4Args& a, hipStream_t s) {
    hipLaunchKernelGGL(k_up4_compress7x4, dim3((unsigned)(((long long)a.H * a.W + 511) / 512), a.B), dim3(128), 0, s, a);
}

}  // namespace fpc
