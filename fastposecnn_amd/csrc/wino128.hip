// wino128.hip — k_conv_wino_c128: split-precision Winograd F(2x2, 3x3) with 128 output channels per workgroup (round 6).
//
// Why a second shape.  k_conv_wino<8, ..., BF3> (net_kernels.hip) works on an 8 x 8 tile patch x 64 output channels per
// workgroup.  Its K-step is bound by vector issue, not by the matrix pipe: per wave 153 vector instructions beside 24 matrix
// instructions, 104 of them the input transform B^T d B and the exact three-way bf16 split of the transformed tile — and that
// input-side work is repeated by every Cout / 64 column-tile workgroup.  Here the patch is 8 x 4 tiles and the
// workgroup carries ALL 16 xi of 128 output channels, so the transform + split of a tile is paid once per 128 channels:
//
//   workgroup = 4 waves, ONE per SIMD (launch bound 256 threads, up to 512 registers per lane: the 256 KB of accumulators
//   — 32 tiles x 128 channels x 16 xi x f32 — are 256 accumulation registers per lane, the other half of the file holds the
//   operands); wave w owns transform row w: 4 xi x 4 column tiles of 32 x 32, 48 v_mfma_f32_32x32x16_bf16 per K-step of 8
//   input channels beside ~170 vector instructions (3.5 per matrix instruction instead of 6.4).  With one wave per SIMD a
//   wave's vector instructions only ever run between its OWN matrix instructions (2-4 cycles each there; beside a SIMD
//   partner that issues matrix instructions back to back they advanced one per matrix instruction, net_kernels.hip).
//
//   weights: a weight fragment is used by exactly one wave exactly once per patch, so staging it through LDS buys nothing:
//   k_wino_pack_c128 writes the image in FRAGMENT ORDER ([Cout/128][Cin/8][wave 4][xi 4][ {b3}: 4 tiles x 64 lanes x 8 B |
//   {b1, b2}: 4 tiles x 64 lanes x 16 B ] = 96 KB per K-step) and every lane loads its 16 + 8 bytes per (xi, tile) straight
//   into the operand registers with buffer loads (1 KB / 512 B per wave instruction, fully coalesced), one K-step ahead: a
//   fragment's registers are reloaded right after its last matrix instruction (96 registers of weights in flight or waiting).
//   The per-CU weight traffic per multiply-add doubles against the 64-channel shape (96 KB per K-step of 32 x 128 x 16 x 8
//   products instead of 48 KB per 64 x 64): at the ~60 B/clk a CU gets from its L2 that is ~1600 cycles per K-step beside
//   1536 cycles of matrix work — the shape trades vector issue for L2 bandwidth.
//
//   input: the raw 18 x 10 region of a K-step (8 channels) goes global -> LDS by LDS-DMA in the permuted, conflict-free unit
//   order of the 64-channel kernel (here 2 x 3 cells of 4 x 4 (row pair, column pair) x 8 parity / channel-half blocks = 12
//   pieces of 1 KB, 45 of every 96 units used), double-buffered, ONE barrier per K-step; the next step's fragments are read,
//   transformed and split between the current step's matrix instructions.  Out-of-image positions are never written (inactive
//   lanes on a zeroed buffer).  The DMA instructions are inline asm (the compiler knows nothing of them): they are issued
//   BEFORE the step's 32 weight loads and waited for with a COUNTED vmcnt(32), so the weight prefetch stays in flight across
//   the barrier; the compiler's own counts for the weight loads are then conservative by the DMA instructions issued since.
//
//   products: x = a1 + a2 + a3, w = b1 + b2 + b3 (bf16 by truncation, exact); kept a1 b1, a1 b2 | a2 b1, a2 b2 | a1 b3, a3 b1 as
//   three matrix instructions per (xi, tile) and K-step: A = {a1, a1} x B = {b1, b2}; {a2, a2} x {b1, b2}; {a1, a3} x {b3, b1}
//   (slot = 4 channels of the lane half); dropped products < 2^-23 of the term, f32 accumulation — the arithmetic of the
//   64-channel BF3 form.
//
//   output transform + epilogue as the 64-channel kernel's (column part in the wave, row part across the four waves through
//   LDS — all 128 channels in ONE pass of 128 KB: two barriers per workgroup instead of four), folded BatchNorm / bias,
//   residual, ReLU, GroupNorm partial sums per patch.
// Reference: the 3x3 / stride-1 convolutions of F/lib/pose_regressor.py:709-743 (smp encoder + FPN decoder, not vendored).
#include <algorithm>
#include <cstdlib>
#include "net_kernels.hpp"

namespace fpc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int kTX = 8, kTY = 4;                  // tile patch: 8 wide, 4 tall (16 x 8 output pixels)
constexpr int kRW = 2 * kTX + 2, kRH = 2 * kTY + 2;      // staged input region 18 x 10
constexpr int kBN = 128;                         // output channels per workgroup
constexpr int kNT = kTX * kTY;                   // 32 tiles = the M of every matrix instruction
constexpr int kInPieces = 12;                    // 1 KB LDS-DMA pieces of one K-step's input image
constexpr int kInFloats = kInPieces * 256;       // 3072 floats per input buffer
constexpr int kStepBytes = 96 * 1024;            // weight image of one K-step
constexpr int kWaveBytes = 24 * 1024;            // ... of which one wave's
constexpr int kXiBytes = 6 * 1024;               // ... of which one xi's: 2 KB {b3} + 4 KB {b1, b2}
constexpr int kLdsFloats = 4 * 2 * kNT * kBN;    // output transform image Z[row 4][cc 2][tile 32][co 128] = 128 KB
static_assert(kLdsFloats >= 2 * kInFloats, "the K loop's two input buffers live in the output image's space");

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ f32x4 fma_s4(float s, f32x4 b, f32x4 a) {      // s * b + a, one v_fma_f32 per element (net_kernels.hip)
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) { float x = b[k], y = a[k], z; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(z) : "v"(s), "v"(x), "v"(y)); r[k] = z; }
    return r;
}
__device__ __forceinline__ f32x4 sub_s4(f32x4 a, f32x4 b) {
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) { float x = a[k], y = b[k], z; asm("v_sub_f32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y)); r[k] = z; }
    return r;
}
__device__ __forceinline__ f32x4 add_s4(f32x4 a, f32x4 b) {
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) { float x = a[k], y = b[k], z; asm("v_add_f32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y)); r[k] = z; }
    return r;
}

}  // namespace

// MODE (diagnostic instantiations, FPC_W2_MODE at launch): bit 0 = the K loop reloads no weights, bit 1 = it stages no input and has
// no barrier — wrong results, the same instruction stream otherwise: what the loop costs without either memory path
template <int MODE>
__global__ __launch_bounds__(256, 1) void k_conv_wino_c128(const WinoArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
    const long long t_entry = a.dbg ? clock64() : 0;
    const int t = threadIdx.x, lane = t & 63;
    const int wi = __builtin_amdgcn_readfirstlane(t >> 6);      // transform row of this wave (wave-uniform)
    const int li = lane & 31, lh = lane >> 5;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
    const int nkb = Cin >> 3;
    // weight slice (group, 128-channel block) fastest: fixed per XCD under round-robin dispatch (k_conv_wino)
    int bid = blockIdx.x;
    const int nnb = Cout / kBN;
    const int nb = bid % nnb; bid /= nnb;
    const int grp = bid % a.groups; bid /= a.groups;
    const int bx = bid % a.tbx; bid /= a.tbx;
    const int by = bid % a.tby;
    const int b = bid / a.tby;
    ConvPtrs P = a.p[0];
    if (grp == 1) P = a.p[1];
    if (grp == 2) P = a.p[2];
    if (grp == 3) P = a.p[3];
    const int ty0 = by * kTY, tx0 = bx * kTX;
    const int y_in0 = 2 * ty0 - 1, x_in0 = 2 * tx0 - 1;

    f32x16 acc[4][4];      // [xi column j][32-channel tile nt]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][nt][r] = 0.f;

    // ---- weights: buffer loads, descriptor over this 128-channel block's images, lane offsets constant, K-step offset scalar
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(reinterpret_cast<const char*>(P.w) + (size_t)nb * nkb * kStepBytes);
    const int vo_u = lane * 16 + 2048, vo_t = lane * 8;       // {b1, b2} fragments behind the xi's 2 KB of {b3} fragments
    int so_w = wi * kWaveBytes;                                // + kStepBytes per K-step
    u32x4 U[4][4];
    u32x2 T[4][4];
#define FPC_W2_LOAD_U(J, NT) U[J][NT] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, vo_u, so_w + (J) * kXiBytes + (NT) * 1024, 0))
#define FPC_W2_LOAD_T(J, NT) T[J][NT] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_w, vo_t, so_w + (J) * kXiBytes + (NT) * 512, 0))
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { FPC_W2_LOAD_U(j, nt); FPC_W2_LOAD_T(j, nt); }
    if (nkb > 1) so_w += kStepBytes;

    // ---- input staging: LDS-DMA pieces (wave + 4 i), i < 3.  The 16-byte unit a lane's data lands in decides the global address
    // it fetches: unit = ((cell * 8 + block) * 16 + 4 * (qh & 3) + (ah & 3)), cell = (ah >> 2) * 3 + (qh >> 2), block = (ry & 1) * 4 +
    // (rx & 1) * 2 + channel half, ah = ry >> 1 (0..4), qh = rx >> 1 (0..8) — k_conv_wino's permuted image for an 18 x 10 region
    const float* isb = P.in + (size_t)b * HW * Cin;            // image base, + 8 floats per step
    unsigned ivo[3];
    bool iok[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int slot = (wi + 4 * i) * 64 + lane;
        const int blk = slot >> 4, res = slot & 15, cell = blk >> 3;
        const int ah = (cell / 3) * 4 + (res & 3), qh = (cell % 3) * 4 + (res >> 2);
        const int hf = blk & 1;
        const int ry = 2 * ah + ((blk >> 2) & 1), rx = 2 * qh + ((blk >> 1) & 1);
        const int y = y_in0 + ry, x = x_in0 + rx;
        iok[i] = ah <= kTY && qh <= kTX && y >= 0 && y < H && x >= 0 && x < W;
        ivo[i] = iok[i] ? (unsigned)((((size_t)y * W + x) * Cin + 4 * hf) * sizeof(float)) : 0u;
    }
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#define FPC_LDS_ADDR(PTR) ((unsigned)(size_t)(__attribute__((address_space(3))) void*)(PTR))
#define FPC_W2_ISSUE_IN(BUF)                                                                                  \
    do {                                                                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_)                                                      \
            if (iok[i_]) asm volatile("s_mov_b32 m0, %0\n s_nop 0\n global_load_lds_dwordx4 %1, %2\n"         \
                                      :: "s"(FPC_LDS_ADDR(lds + (BUF) * kInFloats + (wi + 4 * i_) * 256)), "v"(ivo[i_]), "s"(isb) : "memory", "m0"); \
    } while (0)

    // ---- fragment addressing: this lane's tile, the two region rows of transform row wi, columns 2 txl + c
    const int tyl = li >> 3, txl = li & 7;
    // row pair (ra, rb) and sign of B^T row wi:  0: d0-d2   1: d1+d2   2: d2-d1   3: d1-d3
    const int ra = (wi == 0) ? 0 : (wi == 2 ? 2 : 1);
    const int rb = (wi == 0) ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    const float sgn = (wi == 1) ? 1.f : -1.f;
    auto unit = [&](int r, int ch) {      // float offset of row 2 tyl + r, column 2 (txl + ch), this lane's channel half
        const int ah = tyl + (r >> 1), qh = txl + ch;
        return ((((ah >> 2) * 3 + (qh >> 2)) * 8 + (r & 1) * 4 + lh) * 16 + 4 * (qh & 3) + (ah & 3)) * 4;
    };
    constexpr int in_cs = 2 * 16 * 4;     // + 1 column: the (rx & 1) block bit
    const int in_a[2] = {unit(ra, 0), unit(ra, 1)}, in_b[2] = {unit(rb, 0), unit(rb, 1)};

    // a patch that reaches over the image border zeroes both input buffers once (inactive DMA lanes leave them alone); an
    // interior patch rewrites every unit the fragment reads touch with every step's DMA
    if (y_in0 < 0 || x_in0 < 0 || y_in0 + kRH > H || x_in0 + kRW > W) {
        for (int i = t; i < 2 * kInFloats / 4; i += 256) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    FPC_W2_ISSUE_IN(0);
    if (nkb > 1) isb += 8;
    FPC_W2_ISSUE_IN(1);
    if (nkb > 2) isb += 8;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // pieces of the current step's transformed fragments: pa[j][piece], four channels each
    u32x2 pa[4][3];
    {
        f32x4 e[4], v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            e[c] = fma_s4(sgn, *reinterpret_cast<const f32x4*>(lds + in_b[c >> 1] + (c & 1) * in_cs),
                          *reinterpret_cast<const f32x4*>(lds + in_a[c >> 1] + (c & 1) * in_cs));
        v[0] = sub_s4(e[0], e[2]); v[1] = add_s4(e[1], e[2]); v[2] = sub_s4(e[2], e[1]); v[3] = sub_s4(e[1], e[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) split_bf3(v[j], pa[j][0], pa[j][1], pa[j][2]);
    }
    __syncthreads();       // buffer 0 is refilled by step 0's DMA

#define FPC_W2_MFMA(J, NT, A, B) acc[J][NT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc[J][NT], 0, 0, 0)
    int cur = 0;
    const long long c_begin = a.dbg ? clock64() : 0, r_begin = a.dbg ? wall_clock64() : 0;
#pragma unroll 1
    for (int kb = 0; kb < nkb; ++kb) {
        // input of step kb + 2 -> the buffer step kb's fragments were read from during step kb - 1 (oldest in the queue: see the wait below)
        if (!(MODE & 2)) FPC_W2_ISSUE_IN(cur);
        const float* In = lds + (cur ^ 1) * kInFloats;
        f32x4 da[4], db[4], e[4], vn[4];
        u32x2 pn[4][3];
        float sr[4][4], sq[4][4];      // split residuals of vn[.]: x - p1, x - p1 - p2
        __builtin_amdgcn_s_setprio(1);
        // One wave per SIMD: a vector instruction costs nothing only while the matrix pipe is busy with this wave's previous matrix
        // instruction (32 cycles = ~7 vector issue slots), so every matrix instruction is FOLLOWED by its own small share of the
        // step's other work and a scheduling barrier (four matrix instructions back to back and then 20 vector instructions left the
        // pipe idle behind the fourth: 1980 cycles per K-step without any memory traffic against 1536 of matrix work).
        // Slot s = 12 j + 4 g + nt: g = 0: a1 b1 + a1 b2, g = 1: a2 b1 + a2 b2, g = 2: a1 b3 + a3 b1.  Shares: the next step's fragment
        // reads (slots 0-1), row transform (4-7), column transform (8-11), the four three-way splits in 24 pieces of 3-4 instructions
        // (g = 0 and g = 2 slots of xi 1-3); g = 1 slots build the {b3, b1} operand of their tile, g = 2 slots reload their tile's weights.
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 A0 = {pa[j][0][0], pa[j][0][1], pa[j][0][0], pa[j][0][1]};
            const u32x4 A1 = {pa[j][1][0], pa[j][1][1], pa[j][1][0], pa[j][1][1]};
            const u32x4 A2 = {pa[j][0][0], pa[j][0][1], pa[j][2][0], pa[j][2][1]};
            u32x4 C[4];
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int sl = 12 * j + 4 * g + nt;
                    if (g == 0) FPC_W2_MFMA(j, nt, A0, U[j][nt]);
                    if (g == 1) {
                        FPC_W2_MFMA(j, nt, A1, U[j][nt]);
                        C[nt] = u32x4{T[j][nt][0], T[j][nt][1], U[j][nt][0], U[j][nt][1]};
                    }
                    if (g == 2) {
                        FPC_W2_MFMA(j, nt, A2, C[nt]);
                        if (!(MODE & 1)) { FPC_W2_LOAD_U(j, nt); FPC_W2_LOAD_T(j, nt); }
                    }
                    if (sl < 2) {
#pragma unroll
                        for (int c = 2 * sl; c < 2 * sl + 2; ++c) {
                            da[c] = *reinterpret_cast<const f32x4*>(In + in_a[c >> 1] + (c & 1) * in_cs);
                            db[c] = *reinterpret_cast<const f32x4*>(In + in_b[c >> 1] + (c & 1) * in_cs);
                        }
                    }
                    if (sl >= 4 && sl < 8) e[sl - 4] = fma_s4(sgn, db[sl - 4], da[sl - 4]);
                    if (sl == 8) vn[0] = sub_s4(e[0], e[2]);
                    if (sl == 9) vn[1] = add_s4(e[1], e[2]);
                    if (sl == 10) vn[2] = sub_s4(e[2], e[1]);
                    if (sl == 11) vn[3] = sub_s4(e[1], e[3]);
                    if (j >= 1 && g != 1) {
                        const int q = (j - 1) * 8 + (g == 0 ? nt : 4 + nt), sp = q / 6, part = q % 6;      // split `sp`, piece `part`
                        if (part < 4) {
                            const float x = vn[sp][part];
                            const unsigned xb = __builtin_bit_cast(unsigned, x) & 0xFFFF0000u;
                            const float r = x - __builtin_bit_cast(float, xb);
                            const unsigned rb = __builtin_bit_cast(unsigned, r) & 0xFFFF0000u;
                            sr[sp][part] = r;
                            sq[sp][part] = r - __builtin_bit_cast(float, rb);
                        }
                        if (part == 4) {
                            pn[sp][0] = u32x2{pack_hi16(vn[sp][0], vn[sp][1]), pack_hi16(vn[sp][2], vn[sp][3])};
                            pn[sp][1] = u32x2{pack_hi16(sr[sp][0], sr[sp][1]), pack_hi16(sr[sp][2], sr[sp][3])};
                        }
                        if (part == 5) pn[sp][2] = u32x2{pack_hi16(sq[sp][0], sq[sp][1]), pack_hi16(sq[sp][2], sq[sp][3])};
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) { pa[j][0] = pn[j][0]; pa[j][1] = pn[j][1]; pa[j][2] = pn[j][2]; }
        so_w += kb + 2 < nkb ? kStepBytes : 0;
        isb += kb + 3 < nkb ? 8 : 0;
        // this wave's DMA pieces (issued before the step's 32 weight loads, which stay in flight) have landed
        if (!(MODE & 2)) {
            if (MODE & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            __syncthreads();                                  // everybody's have; this step's fragment reads are done
        }
        cur ^= 1;
    }
#undef FPC_W2_MFMA
#undef FPC_W2_ISSUE_IN
#undef FPC_W2_LOAD_U
#undef FPC_W2_LOAD_T
#undef FPC_LDS_ADDR
#pragma clang diagnostic pop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the last steps' redundant staging has landed before LDS is reused
    const long long t_kend = a.dbg ? clock64() : 0;
    if (a.dbg && lane == 0) {      // tools_dev/wino_stamps.py: shader-clock ticks and 100 MHz reference ticks of the K loop, entry -> loop
        long long* o = a.dbg + ((size_t)blockIdx.x * 4 + wi) * 8;
        o[0] = 0; o[1] = 0; o[2] = 0;
        o[3] = t_kend - c_begin; o[4] = wall_clock64() - r_begin; o[5] = nkb; o[6] = c_begin - t_entry;
    }

    // ---- output transform.  Column part inside the wave: z0 = m0 + m1 + m2, z1 = m1 - m2 - m3; row part across the four
    // transform-row waves through LDS: y0 = z[0] + z[1] + z[2], y1 = z[1] - z[2] - z[3].  Z[row][cc][tile][co 128], one pass.
    // Output stage: thread = (tile pair member, 16-byte channel quad of a 64-channel half): within a ds_read_b128 lane group the
    // 16 quads are 16 different bank slots; a wave stores 4 tiles x 256 contiguous bytes.
    const int oq = t & 15, otl = t >> 4;                      // quad 0..15, tile 0..15 (+ 16 per tile pass)
    f32x4 e_sc[2], e_sh[2];
#pragma unroll
    for (int hc = 0; hc < 2; ++hc) {
        const int n = nb * kBN + hc * 64 + oq * 4;
        e_sc[hc] = P.scale ? *reinterpret_cast<const f32x4*>(P.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
        e_sh[hc] = P.shift ? *reinterpret_cast<const f32x4*>(P.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float m0 = acc[0][nt][r], m1 = acc[1][nt][r], m2 = acc[2][nt][r], m3 = acc[3][nt][r];
            lds[((wi * 2 + 0) * kNT + m) * kBN + nt * 32 + li] = m0 + m1 + m2;
            lds[((wi * 2 + 1) * kNT + m) * kBN + nt * 32 + li] = m1 - m2 - m3;
        }
    __syncthreads();
    f32x4 s1[2], s2[2];
#pragma unroll
    for (int hc = 0; hc < 2; ++hc) { s1[hc] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[hc] = s1[hc]; }
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {
        const int ot = otl + 16 * tp;
        const int oty = ty0 + (ot >> 3), otx = tx0 + (ot & 7);
#pragma unroll
        for (int hc = 0; hc < 2; ++hc) {
            const int n = nb * kBN + hc * 64 + oq * 4;
            f32x4 z[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) z[i][cc] = *reinterpret_cast<const f32x4*>(lds + ((i * 2 + cc) * kNT + ot) * kBN + hc * 64 + oq * 4);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const int y = 2 * oty + rr, x = 2 * otx + cc;
                    if (y >= H || x >= W) continue;
                    f32x4 val = rr == 0 ? z[0][cc] + z[1][cc] + z[2][cc] : z[1][cc] - z[2][cc] - z[3][cc];
                    if (P.scale) val = val * e_sc[hc];
                    val = val + e_sh[hc];
                    const size_t o = ((size_t)b * HW + (size_t)y * W + x) * Cout + n;
                    if (P.res) val += *reinterpret_cast<const f32x4*>(P.res + o);
                    if (a.relu) { val[0] = fmaxf(val[0], 0.f); val[1] = fmaxf(val[1], 0.f); val[2] = fmaxf(val[2], 0.f); val[3] = fmaxf(val[3], 0.f); }
                    *reinterpret_cast<f32x4*>(P.out + o) = val;
                    s1[hc] += val;
                    s2[hc] += val * val;
                }
        }
    }
    if (P.gn_part) {
        // per-channel sums of this workgroup's outputs: a wave holds 4 tiles (lane bits 4-5) x 16 quads (lane bits 0-3) per half:
        // butterfly over the tile bits, then the four waves' sums through LDS in wave order
#pragma unroll
        for (int hc = 0; hc < 2; ++hc)
#pragma unroll
            for (int o = 16; o < 64; o <<= 1)
#pragma unroll
                for (int k = 0; k < 4; ++k) { s1[hc][k] += __shfl_xor(s1[hc][k], o, 64); s2[hc][k] += __shfl_xor(s2[hc][k], o, 64); }
        __syncthreads();
        float* red = lds;                                     // [4 waves][128 ch][2]
        if (lane < 16) {
#pragma unroll
            for (int hc = 0; hc < 2; ++hc)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    red[(wi * kBN + hc * 64 + oq * 4 + k) * 2] = s1[hc][k];
                    red[(wi * kBN + hc * 64 + oq * 4 + k) * 2 + 1] = s2[hc][k];
                }
        }
        __syncthreads();
        if (t < kBN) {
            float u1 = 0.f, u2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { u1 += red[(w * kBN + t) * 2]; u2 += red[(w * kBN + t) * 2 + 1]; }
            const int Pn = a.tbx * a.tby;
            float* g = P.gn_part + (((size_t)b * Pn + by * a.tbx + bx) * Cout + nb * kBN + t) * 2;
            g[0] = u1; g[1] = u2;
        }
    }
    if (a.dbg && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        a.dbg[((size_t)blockIdx.x * 4 + wi) * 8 + 7] = clock64() - t_kend;      // K loop end -> last store acknowledged
    }
}

// OIHW 3x3 weights -> U = G g G^T, every value split exactly into three bf16 pieces (truncation, as split_bf3), packed in the
// fragment order k_conv_wino_c128's lanes load: [Cout/128][Cin/8][wave = xi >> 2][xi & 3][ {b3}: tile 4 x lane 64 x 4 ch (2 KB)
// | {b1 x 4 ch, b2 x 4 ch}: tile 4 x lane 64 (4 KB) ], lane = (channel half) * 32 + (co & 31), tile = (co & 127) >> 5.
__global__ __launch_bounds__(256) void k_wino_pack_c128(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout, int Cin) {
    const long long total = (long long)Cout * Cin;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(g % Cin), co = (int)(g / Cin);
        const float* k = w + ((size_t)co * Cin + ci) * 9;
        float gg[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g0 = k[c], g1 = k[3 + c], g2 = k[6 + c];
            gg[0][c] = g0;
            gg[1][c] = 0.5f * (g0 + g1 + g2);
            gg[2][c] = 0.5f * (g0 - g1 + g2);
            gg[3][c] = g2;
        }
        const int nb = co >> 7, col = co & 127, nt = col >> 5, kb = ci >> 3, cil = ci & 7, e = cil & 3;
        const int ln = (cil >> 2) * 32 + (col & 31);
        unsigned short* img = out + ((size_t)nb * (Cin >> 3) + kb) * (kStepBytes / 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float r0 = gg[i][0], r1 = gg[i][1], r2 = gg[i][2];
            const float u[4] = {r0, 0.5f * (r0 + r1 + r2), 0.5f * (r0 - r1 + r2), r2};
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                const float x = u[jx];
                const unsigned xb = __builtin_bit_cast(unsigned, x) & 0xFFFF0000u;
                const float r = x - __builtin_bit_cast(float, xb);
                const unsigned rb = __builtin_bit_cast(unsigned, r) & 0xFFFF0000u;
                const float q = r - __builtin_bit_cast(float, rb);
                unsigned short* xi = img + (i * kWaveBytes + jx * kXiBytes) / 2;
                xi[(nt * 512 + ln * 8) / 2 + e] = (unsigned short)(__builtin_bit_cast(unsigned, q) >> 16);      // {b3}
                unsigned short* um = xi + (2048 + nt * 1024 + ln * 16) / 2;
                um[e] = (unsigned short)(xb >> 16);                                                              // b1
                um[4 + e] = (unsigned short)(rb >> 16);                                                          // b2
            }
        }
    }
}

int launch_conv_wino_c128(const WinoArgs& a, int groups, hipStream_t s) {
    if (groups < 1 || groups > kMaxGroup || a.Cin % 8 != 0 || a.Cout % kBN != 0 || a.waves != 4) return FPC_EINVAL;
    if ((long long)a.H * a.W * a.Cin * (long long)sizeof(float) >= (1LL << 32)) return FPC_EINVAL;      // 32-bit lane offsets inside one image
    if ((long long)(a.Cin >> 3) * kStepBytes >= (1LL << 31)) return FPC_EINVAL;                          // 31-bit buffer offsets inside one block's images
    if (a.tbx != cdiv(cdiv(a.W, 2), kTX) || a.tby != cdiv(cdiv(a.H, 2), kTY)) return FPC_EINVAL;
    const long long nblk = (long long)a.tbx * a.tby * a.B * (a.Cout / kBN) * groups;
    if (nblk < 1 || nblk >= (1LL << 31)) return FPC_EINVAL;
    static const int mode = getenv("FPC_W2_MODE") ? atoi(getenv("FPC_W2_MODE")) : 0;      // diagnostic
    if (mode == 1) hipLaunchKernelGGL(k_conv_wino_c128<1>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    else if (mode == 2) hipLaunchKernelGGL(k_conv_wino_c128<2>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    else if (mode == 3) hipLaunchKernelGGL(k_conv_wino_c128<3>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_conv_wino_c128<0>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    return check_launch();
}

// split-precision fragment-order image: 24 * Cout * Cin floats (every byte is written)
int launch_wino_pack_c128(const float* w_oihw, float* packed, int Cout, int Cin, hipStream_t s) {
    if (Cin % 8 != 0 || Cout % kBN != 0) return FPC_EINVAL;
    const long long work = (long long)Cout * Cin;
    hipLaunchKernelGGL(k_wino_pack_c128, dim3((unsigned)std::min<long long>((work + 255) / 256, 4096)), dim3(256), 0, s, w_oihw,
                       reinterpret_cast<unsigned short*>(packed), Cout, Cin);
    return check_launch();
}

}  // namespace fpc
