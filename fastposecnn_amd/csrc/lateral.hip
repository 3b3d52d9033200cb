// lateral.hip — k_lateral1x1: the FPN lateral 1x1 convolutions of the four decoders as ONE pixel-resident product
// (smp FPNBlock: p_k = skip_conv(c_k) + nearest_x2(p_{k+1}); fastposecnn_amd/lib/backbone.py, F/lib/pose_regressor.py:709-743).
//
// The lateral of the stride-4 / stride-8 maps is a K = 64 / 128 product that WRITES 256 channels x 4 decoders per pixel:
// 78.6 MB per 640x480 frame for p2 alone, against 4.9 MB read.  The implicit-GEMM kernel (k_conv_igemm, 64 x 64 tiles)
// re-stages — and, in its split-precision form, re-splits — the same 64 pixels once per 64 output channels and spends a
// whole prologue / epilogue on a two-step K loop: 1.45 TB/s written.  Here the roles are swapped: a wave keeps the A
// fragments of ITS 32 pixels (all K channels, split once into the three bf16 planes: 12 KG registers) for the whole
// workgroup lifetime and walks the 32-column weight tiles of all decoders (groups x Cout / 32 of them; `parts` > 1 gives a
// workgroup a contiguous share of that walk when there are too few pixel tiles to fill the chip).  The weight planes are the
// ones k_pack_weight_bf3 already wrote ([plane][Npad][Kpad] bf16): a tile's 3 x 32 rows are staged global -> registers ->
// LDS one tile ahead (16-byte chunks XOR-swizzled by the row: conflict-free ds_read_b128), six v_mfma_f32_32x32x16_bf16 per
// 16-deep k-group (the products p_i q_j with i + j <= 4, common.hpp: split_bf3), f32 accumulation started from the bias.
// A 32 x 32 accumulator register holds one output channel per lane for two pixels: stored as it stands, every store
// instruction writes two full 128-byte lines (MI355X_MICROARCH.md: the full-rate store shape), the nearest-x2 addend is
// fetched the same way before the tile's MFMAs.  Bound: HBM writes.
#include "net_kernels.hpp"

namespace fpc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KG /* K / 16 */, bool UP /* top-down addend */>
__global__ __launch_bounds__(256, 2) void k_lateral1x1(const LatArgs a) {
    constexpr bool UP_AHEAD = KG == 4;      // K = 128 keeps 96 registers of A planes: its addends are fetched in their own iteration
    constexpr int K = KG * 16, ROWB = K * 2, CPR = ROWB / 16, TILEB = 32 * ROWB, NLD = 3 * TILEB / 16 / 256;
    __shared__ __attribute__((aligned(16))) unsigned char s_w[2][3 * TILEB];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, h = lane >> 5;
    const int HW = a.Ho * a.Wo, mt = (HW + 127) >> 7;
    // XCD-aware tile order: workgroups go to the 8 XCDs round-robin by id; with the grid a multiple of 8, XCD x = id % 8 takes the
    // x-th contiguous eighth of the (image, pixel tile, part) list, so the input pixels and the top-down rows two pixel rows share
    // are fetched by one L2 instead of two to eight
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));
    const int part = bid % a.parts;
    bid /= a.parts;
    const int m = bid % mt, b = bid / mt;
    const int tpg = a.Cout >> 5, per = (a.groups * tpg) / a.parts;
    const int t0 = part * per, t1 = t0 + per;

    // ---- this wave's 32 pixels: A fragments of the three planes, all K
    u32x4 A1[KG], A2[KG], A3[KG];
    {
        const int pix = m * 128 + wv * 32 + col;
        const bool rv = pix < HW;
        const float* ap = a.in + ((size_t)b * HW + (rv ? pix : 0)) * K + 8 * h;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (rv) { v0 = *reinterpret_cast<const f32x4*>(ap + 16 * g); v1 = *reinterpret_cast<const f32x4*>(ap + 16 * g + 4); }
            u32x2 p1, p2, p3, q1, q2, q3;
            split_bf3(v0, p1, p2, p3);
            split_bf3(v1, q1, q2, q3);
            A1[g] = u32x4{p1[0], p1[1], q1[0], q1[1]};
            A2[g] = u32x4{p2[0], p2[1], q2[0], q2[1]};
            A3[g] = u32x4{p3[0], p3[1], q3[0], q3[1]};
        }
    }
    // ---- the 16 output rows of this lane (accumulator register i <-> row 8 (i / 4) + (i % 4) + 4 h of the wave's 32)
    const int prow = m * 128 + wv * 32 + 4 * h;
    constexpr bool has_up = UP;
    int uoff[16];
    const float rW = 1.0f / (float)a.Wo;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int P = min(prow + 8 * (i >> 2) + (i & 3), HW - 1);
        // P / Wo by the f32 reciprocal and one correction each way (exact for P < 2^22: launch_lateral1x1 checks the map size;
        // sixteen integer divisions were a third of the prologue)
        int y = (int)((float)P * rW);
        int x = P - y * a.Wo;
        if (x < 0) { --y; x += a.Wo; }
        if (x >= a.Wo) { ++y; x -= a.Wo; }
        uoff[i] = has_up ? ((y >> 1) * (a.Wo >> 1) + (x >> 1)) * a.Cout + col : 0;
    }
    const size_t img_out = (size_t)b * HW * a.Cout, img_up = (size_t)b * (HW >> 2) * a.Cout;

    // chunk swizzle of weight row n: a ds_read_b128 is served in groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the
    // same + 32: MI355X_MICROARCH.md, LDS) whose 16-byte slots must differ modulo 256 bytes.  K = 64: two 128-byte rows per
    // 256 bytes, slot = chunk ^ ((n >> 1) & 7) (the eight even and the eight odd rows of a group differ in (n >> 1) & 7);
    // K = 128: one row per 256 bytes, slot = chunk ^ (n & 15).  (The first version XORed n & 7: two-way conflicts, SQ_LDS_BANK_CONFLICT
    // 3 cycles per LDS instruction in profiles/r05_conv_pmc_c3.json.)
    auto swz = [](int n) { return KG == 4 ? ((n >> 1) & 7) : (n & 15); };
    // ---- weight tile t: 3 planes x 32 rows x ROWB bytes, chunk c = tid + 256 q
    u32x4 st[NLD];
    auto fetch = [&](int t) {
        const int d = t / tpg, c0 = (t - d * tpg) * 32;
        const unsigned short* wp = a.wpl[d];
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int c = tid + 256 * q, plane = c / (32 * CPR), cc = c - plane * (32 * CPR), n = cc / CPR, j = cc - n * CPR;
            st[q] = *reinterpret_cast<const u32x4*>(wp + ((size_t)plane * a.Npad + c0 + n) * K + j * 8);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const int c = tid + 256 * q, plane = c / (32 * CPR), cc = c - plane * (32 * CPR), n = cc / CPR, j = cc - n * CPR;
            *reinterpret_cast<u32x4*>(&s_w[buf][plane * TILEB + n * ROWB + ((j ^ swz(n)) << 4)]) = st[q];
        }
    };
    // the epilogue's addends of a tile: bias of this lane's channel and the 16 top-down values (same rows every tile)
    auto addends = [&](int t, float& bias, float (&upv)[16]) {
        const int d = t / tpg, c0 = (t - d * tpg) * 32;
        // (no bias: the host passes the weight planes as a readable address and a zero mask)
        bias = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a.shift[d][c0 + col]) & a.bias_mask);
        if (has_up) {
            const float* up = a.up[d] + img_up + c0;
#pragma unroll
            for (int i = 0; i < 16; ++i) upv[i] = up[uoff[i]];
        }
    };
    const bool full = m * 128 + wv * 32 + 32 <= HW;      // wave-uniform: all 32 rows of this wave exist (every wave but a ragged map's last)
    float bias_c = 0.f, upc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) upc[i] = 0.f;
    fetch(t0);
    addends(t0, bias_c, upc);
    stage(0);
    __syncthreads();
    // one weight tile: the tile's MFMAs on LDS buffer `buf`, epilogue with the addends (bias_c, upc); the NEXT tile's weight rows
    // (-> LDS after the MFMAs) and, where the registers allow (K = 64), its addends (bias_n, upn) are requested first.  The loop
    // below calls it with the two addend sets swapped every other tile, so no register copy waits for the loads.
    auto tile = [&](int t, int buf, float& bias_c, float (&upc)[16], float& bias_n, float (&upn)[16]) {
        const int d = t / tpg, c0 = (t - d * tpg) * 32;
        const int tn = min(t + 1, t1 - 1);                   // (the last tile re-requests itself: no branch in the loop body)
        fetch(tn);
        if (UP_AHEAD) addends(tn, bias_n, upn);
        else if (t > t0) addends(t, bias_c, upc);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const unsigned char* sb = &s_w[buf][col * ROWB];
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const int o = ((2 * g + h) ^ swz(col)) << 4;
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(sb + o));
            const bf16x8 b2 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(sb + TILEB + o));
            const bf16x8 b3 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(sb + 2 * TILEB + o));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A3[g]), b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A1[g]), b3, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A2[g]), b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A2[g]), b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A1[g]), b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A1[g]), b1, acc, 0, 0, 0);
        }
        float* out = a.out[d] + img_out + (size_t)prow * a.Cout + c0 + col;
        if (full) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = acc[i] + bias_c;
                if (has_up) v += upc[i];
                if (a.relu) v = fmaxf(v, 0.f);
                out[(size_t)(8 * (i >> 2) + (i & 3)) * a.Cout] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = acc[i] + bias_c;
                if (has_up) v += upc[i];
                if (a.relu) v = fmaxf(v, 0.f);
                if (prow + 8 * (i >> 2) + (i & 3) < HW) out[(size_t)(8 * (i >> 2) + (i & 3)) * a.Cout] = v;
            }
        }
        stage(buf ^ 1);
        __syncthreads();
    };
    float bias_n = 0.f, upn[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) upn[i] = 0.f;
    // every load of the prologue has landed before the loop: the compiler's wait-count state at the loop header then comes from
    // the back edge alone (with prologue loads pending it put a vmcnt(0) — stores included — at the top of every tile pair)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    int t = t0;
    for (; t + 1 < t1; t += 2) {                             // pairs: no conditional inside the loop (wait counts stay counted)
        tile(t, 0, bias_c, upc, bias_n, upn);
        if (UP_AHEAD) tile(t + 1, 1, bias_n, upn, bias_c, upc);
        else tile(t + 1, 1, bias_c, upc, bias_n, upn);
    }
    if (t < t1) tile(t, 0, bias_c, upc, bias_n, upn);       // odd walk length (uniform)
}

int launch_lateral1x1(const LatArgs& a, hipStream_t s) {
    if (a.groups < 1 || a.groups > kMaxGroup || a.parts < 1 || a.Cout % 32 != 0 || (a.Kpad != 64 && a.Kpad != 128) ||
        (a.groups * (a.Cout / 32)) % a.parts != 0 || a.B < 1 || a.Ho < 1 || a.Wo < 1 || !a.in)
        return FPC_EINVAL;
    for (int g = 0; g < a.groups; ++g) {
        if (!a.wpl[g] || !a.out[g] || (a.up[g] != nullptr) != (a.up[0] != nullptr)) return FPC_EINVAL;
    }
    if (a.up[0] && ((a.Ho | a.Wo) & 1)) return FPC_EINVAL;
    if ((long long)a.Ho * a.Wo * a.Cout >= (1LL << 31) || (long long)a.Ho * a.Wo >= (1LL << 22))
        return FPC_EINVAL;      // 32-bit offsets inside one image; pixel index / Wo through the f32 reciprocal
    const long long grid = (long long)((a.Ho * a.Wo + 127) / 128) * a.B * a.parts;
    if (grid >= (1LL << 31)) return FPC_EINVAL;
    LatArgs b = a;
    b.bias_mask = 0xFFFFFFFFu;
    for (int g = 0; g < a.groups; ++g)
        if (!b.shift[g]) { b.shift[g] = reinterpret_cast<const float*>(a.wpl[g]); b.bias_mask = 0u; }
    for (int g = 0; g < a.groups; ++g)
        if (b.bias_mask == 0u && a.shift[g]) return FPC_EINVAL;      // bias for all groups or for none
    const bool up = a.up[0] != nullptr;
    if (a.Kpad == 64 && up) hipLaunchKernelGGL((k_lateral1x1<4, true>), dim3((unsigned)grid), dim3(256), 0, s, b);
    else if (a.Kpad == 64) hipLaunchKernelGGL((k_lateral1x1<4, false>), dim3((unsigned)grid), dim3(256), 0, s, b);
    else if (up) hipLaunchKernelGGL((k_lateral1x1<8, true>), dim3((unsigned)grid), dim3(256), 0, s, b);
    else hipLaunchKernelGGL((k_lateral1x1<8, false>), dim3((unsigned)grid), dim3(256), 0, s, b);
    return check_launch();
}

}  // namespace fpc
