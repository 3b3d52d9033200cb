// wino_h3.hip — k_conv_wino_h3: wino_h2.hip's four-wave Winograd F(2x2, 3x3) convolution on two fp16 pieces per operand with THREE
// piece products instead of four, over PAIRS of 8-channel K-steps (round 6, last form).
//
// Why.  k_conv_wino_h2 runs at the chip's power limit: taking its barrier out saves 10 % of its cycles and none of its time (the clock
// drops from 1.93 to 1.77 GHz), taking a quarter of its matrix instructions out saves no cycles and 11 % of its time (the clock rises to
// 2.15 GHz) — profiles/r06_wino_forms.md.  What shortens it is less work per product, not fewer stalls.
//   x w = (h1 + h2 + rx)(g1 + g2 + rw),  |h2| <= 2^-11 |x|, |g2| <= 2^-11 |w|, |rx| <= 2^-22 |x|, |rw| <= 2^-22 |w|
// The form keeps h1 g1 + h2 g1 + h1 g2 and drops h2 g2 (<= 2^-22 |x w|: the size of the two terms every two-piece form already
// drops, x rw and rx w).  Three products do not fit two matrix instructions per 8 channels, but they fit THREE per 16: the K dimension
// of v_mfma_f32_32x32x16_f16 carries 4 channels of the even K-step and 4 of the odd one,
//   A1 = {h1 even, h1 odd}   A2 = {h2 even, h2 odd}      B1 = {g1 even, g1 odd}   B2 = {g2 even, g2 odd}
//   acc += A1 B1 + A2 B1 + A1 B2
// so that 16 channels cost 48 matrix instructions per wave instead of 64, and — the pieces of a value are written ONCE, into the half of
// the operand tuple that belongs to its K-step — none of the 32 operand copies per step of the {h1, h1} / {h2, h2} form.
//
// Schedule of a pair p (K-steps 2p, 2p + 1), one barrier:
//   E        no matrix instruction: the sixteen fragment reads of step 2p + 1 (64 KB per workgroup = 512 cycles of the LDS pipe) are
//            issued first; under them the LDS-DMA burst of steps 2p + 3 and 2p + 4 (ring of four 18 KB buffers inside the output image's
//            space; a step past the last is not staged) and the pending split of xi 3 (even half); then the transform and xi 0's split
//            into the odd halves
//   O        48 matrix instructions (xi j: A1 B1 x 4, A2 B1 x 4, A1 B2 x 4) with one item of work behind each of the first 40 — two
//            behind the first twelve: the odd step's xi 1-3 splits (before xi 1's first instruction) beside step 2p + 2's fragment
//            reads and row transform; then its column transform, and the split of xi 0-2 into the even halves once xi j's last
//            matrix instruction has issued (xi 3 waits for the next E); each weight fragment is reloaded in place for the next pair
//            after its last use
//   end      counted wait for this wave's staging pieces (the 16 weight loads behind them stay in flight), barrier
// Entry: the weight fragments of pair 0 and steps 0, 1, 2 are requested together; step 0's transform runs while steps 1 and 2 land.
// Range, scaling, the input image's layout, the entry and the output transform are wino_h2.hip's; Cin must be a multiple of 16.
// Reference: the 3x3 / stride-1 convolutions of F/lib/pose_regressor.py:709-743 (smp encoder + FPN decoder, not vendored).
#include <algorithm>
#include <cstdlib>
#include "net_kernels.hpp"

namespace fpc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int kTX = 8, kTY = 8;                  // tile patch 8 x 8 (16 x 16 output pixels)
constexpr int kRW = 2 * kTX + 2, kRH = 2 * kTY + 2;      // staged input region 18 x 18
constexpr int kBN = 64;                          // output channels per workgroup
constexpr int kNT = kTX * kTY;                   // 64 tiles = two M halves
constexpr int kInPieces = 18;                    // 1 KB LDS-DMA pieces of one K-step's input image (k_conv_wino's permuted image)
constexpr int kInFloats = kInPieces * 256;       // 4608 floats per input buffer
constexpr int kRing = 4;                         // input buffers: step s lives in buffer s & 3
constexpr int kPairBytes = 16 * 2 * 2 * 64 * 16; // k_wino_pack_h3's image of one pair of K-steps: [xi 16][tile 2][piece 2][lane 64] x 16 bytes {4 ch of the even step, 4 ch of the odd step}
constexpr int kLdsFloats = 4 * 2 * kNT * kBN;    // output transform image = 128 KB
static_assert(kLdsFloats >= kRing * kInFloats, "the K loop's input ring lives in the output image's space");

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000);
}
// one scalar instruction per element (the file is built with -fno-slp-vectorize: beside matrix instructions a packed f32 instruction
// costs more than the two scalar ones it replaces).  Plain C++, not inline asm: the compiler brackets an asm statement it cannot see
// into with hazard s_nops (4 issue cycles each).  sgn = +-1: the fused form is exact either way.
__device__ __forceinline__ f32x4 fma_s4(float s, f32x4 b, f32x4 a) {
    return f32x4{__builtin_fmaf(s, b[0], a[0]), __builtin_fmaf(s, b[1], a[1]), __builtin_fmaf(s, b[2], a[2]), __builtin_fmaf(s, b[3], a[3])};
}
__device__ __forceinline__ f32x4 sub_s4(f32x4 a, f32x4 b) { return f32x4{a[0] - b[0], a[1] - b[1], a[2] - b[2], a[3] - b[3]}; }
__device__ __forceinline__ f32x4 add_s4(f32x4 a, f32x4 b) { return f32x4{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]}; }

}  // namespace

// VAR (diagnostic, FPC_H3_VAR at launch): 1 = a piece's residual by conversion + subtraction instead of v_fma_mix_f32 (the same bits)
template <int VAR>
__global__ __launch_bounds__(256, 1) void k_conv_wino_h3(const WinoArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
    const long long t_entry = a.dbg ? clock64() : 0;
    const int t = threadIdx.x, lane = t & 63;
    const int wi = __builtin_amdgcn_readfirstlane(t >> 6);      // transform row of this wave (wave-uniform)
    const int li = lane & 31, lh = lane >> 5;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
    const int nkb = Cin >> 3;
    // weight slice (group, 64-channel block) fastest: fixed per XCD under round-robin dispatch (k_conv_wino)
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

    const int npair = nkb >> 1;      // (the launcher refuses an odd number of K-steps)

    f32x16 acc[4][2][2];      // [xi column j][tile half mt][32-channel tile nt]; zeroed while the first operands are on their way
    // ---- weights: buffer loads of this wave's fragments, per pair of K-steps one 16-byte B1 = {g1 even, g1 odd} and one B2 = {g2 even,
    // g2 odd} per (xi, 32-channel tile, lane)
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(reinterpret_cast<const char*>(P.w) + (size_t)nb * npair * kPairBytes);
    const float inv_s = P.w[(size_t)nnb * npair * (kPairBytes / 4)];      // 1 / (the power of two the weights were scaled by)
    const int vo_u = lane * 16;
    int so_u = wi * 4 * 4096;      // this wave's four xi; + kPairBytes per pair
    u32x4 U1[4][2], U2[4][2];
#define FPC_H3_LOAD_U1_AT(SO, J, NT) U1[J][NT] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, vo_u, (SO) + (J) * 4096 + (NT) * 2048, 0))
#define FPC_H3_LOAD_U2_AT(SO, J, NT) U2[J][NT] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, vo_u, (SO) + (J) * 4096 + (NT) * 2048 + 1024, 0))
#define FPC_H3_LOAD_U1(J, NT) FPC_H3_LOAD_U1_AT(so_u, J, NT)
#define FPC_H3_LOAD_U2(J, NT) FPC_H3_LOAD_U2_AT(so_u, J, NT)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) { FPC_H3_LOAD_U1(j, nt); FPC_H3_LOAD_U2(j, nt); }
    if (npair > 1) so_u += kPairBytes;

    // ---- input staging: LDS-DMA pieces (wave + 4 i), i < 5 (18 pieces).  The 16-byte unit a lane's data lands in decides the
    // global address it fetches (k_conv_wino, PERM): unit = (cell * 8 + block) * 16 + 4 * (qh & 3) + (ah & 3), cell = (ah >> 2) * 3 +
    // (qh >> 2), block = (ry & 1) * 4 + (rx & 1) * 2 + channel half, ah = ry >> 1, qh = rx >> 1 (0..8)
    const float* isb = P.in + (size_t)b * HW * Cin;            // image base, + 8 floats per step
    unsigned ivo[5];
    bool iok[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int slot = (wi + 4 * i) * 64 + lane;
        const int blk = slot >> 4, res = slot & 15, cell = blk >> 3;
        const int ah = (cell / 3) * 4 + (res & 3), qh = (cell % 3) * 4 + (res >> 2);
        const int hf = blk & 1;
        const int ry = 2 * ah + ((blk >> 2) & 1), rx = 2 * qh + ((blk >> 1) & 1);
        const int y = y_in0 + ry, x = x_in0 + rx;
        iok[i] = wi + 4 * i < kInPieces && ah <= kTY && qh <= kTX && y >= 0 && y < H && x >= 0 && x < W;
        ivo[i] = iok[i] ? (unsigned)((((size_t)y * W + x) * Cin + 4 * hf) * sizeof(float)) : 0u;
    }
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#define FPC_LDS_ADDR(PTR) ((unsigned)(size_t)(__attribute__((address_space(3))) void*)(PTR))
    // One asm block, no branch: EXEC is set to each piece's lane mask (a wave-uniform 64-bit value; 0 for a piece this wave does not
    // have or whose positions all lie outside the image: the instruction then moves nothing but still counts in vmcnt, so every wave
    // issues exactly five VMEM instructions per step whatever the patch).  The compiler's own if (mask) form cost ~10 scalar /
    // branch instructions per piece, in a loop that is bound by instruction issue.
    unsigned long long imask[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) imask[i] = __ballot(iok[i]);
#define FPC_H3_ISSUE_IN(BUF, PTR)                                                                             \
    do {                                                                                                      \
        unsigned long long sv_;                                                                               \
        const unsigned l0_ = FPC_LDS_ADDR(lds + (BUF) * kInFloats + wi * 256);                                \
        asm volatile("s_mov_b64 %0, exec\n"                                                                   \
                     "s_mov_b64 exec, %1\n s_mov_b32 m0, %6\n s_nop 0\n global_load_lds_dwordx4 %11, %16\n"   \
                     "s_mov_b64 exec, %2\n s_mov_b32 m0, %7\n s_nop 0\n global_load_lds_dwordx4 %12, %16\n"   \
                     "s_mov_b64 exec, %3\n s_mov_b32 m0, %8\n s_nop 0\n global_load_lds_dwordx4 %13, %16\n"   \
                     "s_mov_b64 exec, %4\n s_mov_b32 m0, %9\n s_nop 0\n global_load_lds_dwordx4 %14, %16\n"   \
                     "s_mov_b64 exec, %5\n s_mov_b32 m0, %10\n s_nop 0\n global_load_lds_dwordx4 %15, %16\n"  \
                     "s_mov_b64 exec, %0\n"                                                                   \
                     : "=&s"(sv_)                                                                             \
                     : "s"(imask[0]), "s"(imask[1]), "s"(imask[2]), "s"(imask[3]), "s"(imask[4]),             \
                       "s"(l0_), "s"(l0_ + 4096), "s"(l0_ + 8192), "s"(l0_ + 12288), "s"(l0_ + 16384),        \
                       "v"(ivo[0]), "v"(ivo[1]), "v"(ivo[2]), "v"(ivo[3]), "v"(ivo[4]), "s"(PTR)              \
                     : "memory", "m0");                                                                       \
    } while (0)

    // ---- fragment addressing: this lane's tile of half 0 (half 1 = four tile rows further down = + 3 cells), the two region rows of
    // transform row wi, columns 2 txl + c
    const int tyl = li >> 3, txl = li & 7;
    // row pair (ra, rb) and sign of B^T row wi:  0: d0-d2   1: d1+d2   2: d2-d1   3: d1-d3
    const int ra = (wi == 0) ? 0 : (wi == 2 ? 2 : 1);
    const int rb = (wi == 0) ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    const float sgn = (wi == 1) ? 1.f : -1.f;
    auto unit = [&](int r, int ch) {      // float offset of row 2 tyl + r, column 2 (txl + ch), this lane's channel half
        const int ah = tyl + (r >> 1), qh = txl + ch;
        return ((((ah >> 2) * 3 + (qh >> 2)) * 8 + (r & 1) * 4 + lh) * 16 + 4 * (qh & 3) + (ah & 3)) * 4;
    };
    constexpr int in_cs = 2 * 16 * 4;          // + 1 column: the (rx & 1) block bit
    constexpr int in_ms = 3 * 8 * 16 * 4;      // + 4 tile rows (tile half 1): the next row of cells
    const int in_a[2] = {unit(ra, 0), unit(ra, 1)}, in_b[2] = {unit(rb, 0), unit(rb, 1)};

    // a patch that reaches over the image border zeroes the input ring once (inactive DMA lanes leave it alone); an
    // interior patch rewrites every unit the fragment reads touch with every step's DMA
    if (y_in0 < 0 || x_in0 < 0 || y_in0 + kRH > H || x_in0 + kRW > W) {
        for (int i = t; i < kRing * kInFloats / 4; i += 256) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // steps 0, 1, 2 -> buffers 0, 1, 2
    FPC_H3_ISSUE_IN(0, isb);
    isb += 8;                      // (nkb >= 2)
    FPC_H3_ISSUE_IN(1, isb);
    isb += 8;
    FPC_H3_ISSUE_IN(2, nkb > 2 ? isb : isb - 8);      // (nkb = 2: step 1 again — the same count of pieces in front of the first wait)
    isb += 8;
    int sn = 3;                    // the next step to stage
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][mt][nt][r] = 0.f;

    const long long t_issued = a.dbg ? clock64() : 0;
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // the weight fragments and step 0 (steps 1, 2 land under step 0's transform)
    const long long t_landed = a.dbg ? clock64() : 0;
    __syncthreads();
    const long long t_synced = a.dbg ? clock64() : 0;

    // operands as the matrix instructions take them: TA1[j][mt] = {h1 of the pair's even step, h1 of its odd step}, TA2[j][mt] = {h2
    // even, h2 odd} (four channels per piece and half); vn[mt][j]: the transformed values of the step whose pieces are being built
    // (xi 3's wait there from the end of O to the next E)
    u32x4 TA1[4][2], TA2[4][2];
    f32x4 vn[2][4];
    float m1;
    asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(m1));      // -1.0f, opaque
    // one pair of values of vn[MT][J] -> its fp16 pieces, into half HALF of the operands (pinned to its slot by the volatile asm that
    // reads them, as wino_w4.hip's items)
#define FPC_H3_SPLIT_PAIR(J, MT, PAIR, HALF) FPC_H3_SPLIT_PAIR_V(vn, J, MT, PAIR, HALF)
#define FPC_H3_SPLIT_PAIR_V(V, J, MT, PAIR, HALF)                                                             \
    do {                                                                                                      \
        const float x0_ = V[MT][J][2 * (PAIR)], x1_ = V[MT][J][2 * (PAIR) + 1];                               \
        const fp16x2 h_ = __builtin_amdgcn_cvt_pkrtz(x0_, x1_);                                               \
        /* x - h1 in ONE instruction: v_fma_mix_f32 reads the fp16 piece in place (m1 = -1 in a scalar register the compiler cannot  \
           fold); exact like the conversion + subtraction it replaces (VAR 1) */                                                  \
        const float r0_ = VAR == 1 ? x0_ - (float)h_[0] : __builtin_fmaf((float)h_[0], m1, x0_);              \
        const float r1_ = VAR == 1 ? x1_ - (float)h_[1] : __builtin_fmaf((float)h_[1], m1, x1_);              \
        const unsigned k1_ = __builtin_bit_cast(unsigned, h_);                                                \
        const unsigned k2_ = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(r0_, r1_));              \
        asm volatile("" :: "v"(k1_), "v"(k2_));                                                               \
        TA1[J][MT][2 * (HALF) + (PAIR)] = k1_; TA2[J][MT][2 * (HALF) + (PAIR)] = k2_;                         \
    } while (0)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) { TA1[j][mt] = u32x4{0u, 0u, 0u, 0u}; TA2[j][mt] = u32x4{0u, 0u, 0u, 0u}; }
    {      // step 0 -> the even halves of xi 0-2; xi 3 stays in vn for E of pair 0
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 e[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                e[c] = fma_s4(sgn, *reinterpret_cast<const f32x4*>(lds + in_b[c >> 1] + (c & 1) * in_cs + mt * in_ms),
                              *reinterpret_cast<const f32x4*>(lds + in_a[c >> 1] + (c & 1) * in_cs + mt * in_ms));
            vn[mt][0] = sub_s4(e[0], e[2]); vn[mt][1] = add_s4(e[1], e[2]); vn[mt][2] = sub_s4(e[2], e[1]); vn[mt][3] = sub_s4(e[1], e[3]);
#pragma unroll
            for (int j = 0; j < 3; ++j) { FPC_H3_SPLIT_PAIR(j, mt, 0, 0); FPC_H3_SPLIT_PAIR(j, mt, 1, 0); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // steps 1 and 2
    __syncthreads();       // everybody's have landed; buffer 0 is refilled by the pieces of step 4, issued at the top of pair 0

#define FPC_H3_MFMA(J, MT, NT, A, B) acc[J][MT][NT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), acc[J][MT][NT], 0, 0, 0)
#define FPC_H3_PIN4(V) asm volatile("" :: "v"(V))
#define FPC_H3_SPLIT_Q(J, Q, HALF) do { if ((Q) == 0) FPC_H3_SPLIT_PAIR(J, 0, 0, HALF); if ((Q) == 1) FPC_H3_SPLIT_PAIR(J, 0, 1, HALF); if ((Q) == 2) FPC_H3_SPLIT_PAIR(J, 1, 0, HALF); if ((Q) == 3) FPC_H3_SPLIT_PAIR(J, 1, 1, HALF); } while (0)
    const long long c_begin = a.dbg ? clock64() : 0, r_begin = a.dbg ? wall_clock64() : 0;
#pragma unroll 1
    for (int p = 0; p < npair; ++p) {
        // ---- E: step 2p + 1's fragment reads and transform; the odd halves of xi 0 (xi 1-3 follow behind O's first matrix instructions).
        // No matrix instruction is in flight behind the barrier: plain code, the sixteen fragment reads issued together.
        {
            // the sixteen fragment reads first: 64 KB per workgroup = 512 cycles of the LDS pipe, under the staging burst and xi 3's split
            const float* InE = lds + ((2 * p + 1) & 3) * kInFloats;
            f32x4 ea[2][4], eb[2][4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ea[mt][c] = *reinterpret_cast<const f32x4*>(InE + in_a[c >> 1] + (c & 1) * in_cs + mt * in_ms);
                    eb[mt][c] = *reinterpret_cast<const f32x4*>(InE + in_b[c >> 1] + (c & 1) * in_cs + mt * in_ms);
                }
            __builtin_amdgcn_sched_barrier(0);
            // inputs of steps 2p + 3 and 2p + 4 -> the buffers steps 2p - 1 and 2p were read from before the last barrier
            // (a step past the last is not staged: the end-of-pair wait counts the sixteen weight loads BEHIND the pieces, so fewer
            // pieces in front of them keep it exact)
            if (sn < nkb) { FPC_H3_ISSUE_IN((sn & 3), isb); isb += 8; }
            if (sn + 1 < nkb) { FPC_H3_ISSUE_IN(((sn + 1) & 3), isb); isb += 8; }
            sn += 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) FPC_H3_SPLIT_Q(3, q, 0);      // xi 3 of step 2p: its operands were in use until the end of O
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x4 e[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) e[c] = fma_s4(sgn, eb[mt][c], ea[mt][c]);
                vn[mt][0] = sub_s4(e[0], e[2]); vn[mt][1] = add_s4(e[1], e[2]); vn[mt][2] = sub_s4(e[2], e[1]); vn[mt][3] = sub_s4(e[1], e[3]);
                FPC_H3_SPLIT_PAIR(0, mt, 0, 1); FPC_H3_SPLIT_PAIR(0, mt, 1, 1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- O: the pair's 48 matrix instructions; slot sl = 12 j + 4 g + 2 mt + nt, g = 0: A1 B1, 1: A2 B1, 2: A1 B2.  One item per slot:
        //   sl  0- 3  fragment reads of step 2p + 2       sl  4-11  row transform e       sl 12-19  column transform vn
        //   sl  0- 3 / 4-7 / 8-11  ALSO the split of xi 1 / 2 / 3 of step 2p + 1 into the odd halves (before xi 1's first matrix instruction, slot 12,
        //             and before slot 12 overwrites vn)
        //   sl 20-23 / 24-27 / 36-39  split of xi 0 / 1 / 2 of step 2p + 2 into the even halves (xi j's last matrix instruction: slot 12 j + 11)
        // A weight fragment's last matrix instruction is followed by its reload for the next pair.
        const float* In = lds + ((2 * p + 2) & 3) * kInFloats;
        f32x4 da[2][4], db[2][4], e[2][4];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int sl = 12 * j + 4 * g + 2 * mt + nt;
                        if (g == 0) FPC_H3_MFMA(j, mt, nt, TA1[j][mt], U1[j][nt]);
                        if (g == 1) {
                            FPC_H3_MFMA(j, mt, nt, TA2[j][mt], U1[j][nt]);
                            if (mt == 1) FPC_H3_LOAD_U1(j, nt);
                        }
                        if (g == 2) {
                            FPC_H3_MFMA(j, mt, nt, TA1[j][mt], U2[j][nt]);
                            if (mt == 1) FPC_H3_LOAD_U2(j, nt);
                        }
                        if (sl < 4) {
#pragma unroll
                            for (int c = 2 * (sl & 1); c < 2 * (sl & 1) + 2; ++c) {
                                da[sl >> 1][c] = *reinterpret_cast<const f32x4*>(In + in_a[c >> 1] + (c & 1) * in_cs + (sl >> 1) * in_ms);
                                db[sl >> 1][c] = *reinterpret_cast<const f32x4*>(In + in_b[c >> 1] + (c & 1) * in_cs + (sl >> 1) * in_ms);
                            }
                        }
                        if (sl >= 4 && sl < 12) {
                            const int m_ = (sl - 4) >> 2, c = (sl - 4) & 3;
                            e[m_][c] = fma_s4(sgn, db[m_][c], da[m_][c]);
                            FPC_H3_PIN4(e[m_][c]);
                        }
                        if (sl < 4) FPC_H3_SPLIT_Q(1, sl, 1);
                        if (sl >= 4 && sl < 8) FPC_H3_SPLIT_Q(2, sl - 4, 1);
                        if (sl >= 8 && sl < 12) FPC_H3_SPLIT_Q(3, sl - 8, 1);
                        if (sl >= 12 && sl < 20) {
                            const int m_ = (sl - 12) >> 2, jx = (sl - 12) & 3;
                            if (jx == 0) vn[m_][0] = sub_s4(e[m_][0], e[m_][2]);
                            if (jx == 1) vn[m_][1] = add_s4(e[m_][1], e[m_][2]);
                            if (jx == 2) vn[m_][2] = sub_s4(e[m_][2], e[m_][1]);
                            if (jx == 3) vn[m_][3] = sub_s4(e[m_][1], e[m_][3]);
                            FPC_H3_PIN4(vn[m_][jx]);
                        }
                        if (sl >= 20 && sl < 24) FPC_H3_SPLIT_Q(0, sl - 20, 0);
                        if (sl >= 24 && sl < 28) FPC_H3_SPLIT_Q(1, sl - 24, 0);
                        if (sl >= 36 && sl < 40) FPC_H3_SPLIT_Q(2, sl - 36, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
        __builtin_amdgcn_s_setprio(0);
        so_u += p + 2 < npair ? kPairBytes : 0;
        // this wave's ten pieces (issued before the pair's 16 weight loads, which stay in flight) have landed
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __syncthreads();                                       // everybody's have; this pair's fragment reads are done
    }
#undef FPC_H3_MFMA
#undef FPC_H3_SPLIT_PAIR
#undef FPC_H3_SPLIT_PAIR_V
#undef FPC_H3_SPLIT_Q
#undef FPC_H3_PIN4
#undef FPC_H3_ISSUE_IN
#undef FPC_H3_LOAD_U1
#undef FPC_H3_LOAD_U2
#undef FPC_H3_LOAD_U1_AT
#undef FPC_H3_LOAD_U2_AT
#undef FPC_LDS_ADDR
#pragma clang diagnostic pop
    // (no staging is in flight here: a step past the last is never issued, every real step was waited for at the end of its pair; the
    // last pair's redundant weight reloads target registers, whose reuse the compiler guards itself)
    const long long t_kend = a.dbg ? clock64() : 0;
    if (a.dbg && lane == 0) {      // tools_dev/wino_stamps.py: shader-clock ticks and 100 MHz reference ticks of the K loop, entry -> loop
        long long* o = a.dbg + ((size_t)blockIdx.x * 4 + wi) * 8;
        o[0] = t_issued - t_entry; o[1] = t_landed - t_issued; o[2] = t_synced - t_landed;      // entry: set-up + issue | first operands land | barrier
        o[3] = t_kend - c_begin; o[4] = wall_clock64() - r_begin; o[5] = nkb; o[6] = c_begin - t_entry;
    }

    // ---- output transform.  Column part inside the wave: z0 = m0 + m1 + m2, z1 = m1 - m2 - m3; row part across the four
    // transform-row waves through LDS: y0 = z[0] + z[1] + z[2], y1 = z[1] - z[2] - z[3].  Z[row][cc][tile 64][co 64], one pass.
    // Output stage: thread = (tile of a 16-tile pass, 16-byte channel quad): within a ds_read_b128 lane group the 16 quads are 16
    // different bank slots; a wave stores 4 tiles x 256 contiguous bytes.
    const int oq = t & 15, otl = t >> 4;                      // quad 0..15, tile 0..15 (+ 16 per tile pass)
    const int n = nb * kBN + oq * 4;
    const f32x4 e_sc = P.scale ? *reinterpret_cast<const f32x4*>(P.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
    const f32x4 e_sh = P.shift ? *reinterpret_cast<const f32x4*>(P.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    // the residual of this thread's 4 x 4 outputs is requested BEFORE the output transform's barriers (one workgroup per CU: nothing
    // else hides that latency; k_conv_wino does the same)
    f32x4 e_res[4][4];
#pragma unroll
    for (int tp = 0; tp < 4; ++tp)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ot = otl + 16 * tp;
            const int y = 2 * (ty0 + (ot >> 3)) + (q >> 1), x = 2 * (tx0 + (ot & 7)) + (q & 1);
            e_res[tp][q] = (P.res && y < H && x < W) ? *reinterpret_cast<const f32x4*>(P.res + ((size_t)b * HW + (size_t)y * W + x) * Cout + n)
                                                     : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    __syncthreads();
    float* const zb = lds + ((wi * 2) * kNT + 4 * lh) * kBN + li;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // one base register per lane + a compile-time offset (< 64 KB: the instruction's immediate) per store
                const int mc = mt * 32 + (r & 3) + 8 * (r >> 2);
                const float m0 = acc[0][mt][nt][r], m1 = acc[1][mt][nt][r], m2 = acc[2][mt][nt][r], m3 = acc[3][mt][nt][r];
                zb[(0 * kNT + mc) * kBN + nt * 32] = m0 + m1 + m2;
                zb[(1 * kNT + mc) * kBN + nt * 32] = m1 - m2 - m3;
            }
    __syncthreads();
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tp = 0; tp < 4; ++tp) {
        const int ot = otl + 16 * tp;
        const int oty = ty0 + (ot >> 3), otx = tx0 + (ot & 7);
        f32x4 z[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) z[i][cc] = *reinterpret_cast<const f32x4*>(lds + ((i * 2 + cc) * kNT + ot) * kBN + oq * 4);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int y = 2 * oty + rr, x = 2 * otx + cc;
                if (y >= H || x >= W) continue;
                f32x4 val = (rr == 0 ? z[0][cc] + z[1][cc] + z[2][cc] : z[1][cc] - z[2][cc] - z[3][cc]) * inv_s;      // (a power of two: exact)
                if (P.scale) val = val * e_sc;
                val = val + e_sh;
                const size_t o = ((size_t)b * HW + (size_t)y * W + x) * Cout + n;
                if (P.res) val += e_res[tp][2 * rr + cc];
                if (a.relu) { val[0] = fmaxf(val[0], 0.f); val[1] = fmaxf(val[1], 0.f); val[2] = fmaxf(val[2], 0.f); val[3] = fmaxf(val[3], 0.f); }
                *reinterpret_cast<f32x4*>(P.out + o) = val;
                s1 += val;
                s2 += val * val;
            }
    }
    if (P.gn_part) {
        // per-channel sums of this workgroup's outputs: a wave holds 4 tiles (lane bits 4-5) x 16 quads (lane bits 0-3) per pass:
        // butterfly over the tile bits, then the four waves' sums through LDS in wave order
#pragma unroll
        for (int o = 16; o < 64; o <<= 1)
#pragma unroll
            for (int k = 0; k < 4; ++k) { s1[k] += __shfl_xor(s1[k], o, 64); s2[k] += __shfl_xor(s2[k], o, 64); }
        __syncthreads();
        float* red = lds;                                     // [4 waves][64 ch][2]
        if (lane < 16) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { red[(wi * kBN + oq * 4 + k) * 2] = s1[k]; red[(wi * kBN + oq * 4 + k) * 2 + 1] = s2[k]; }
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

// k_wino_pack_h2's transform, scale and split; the fragment order of the PAIR form:
// [Cout/64][Cin/16][xi 16][tile 2][piece 2][lane 64] x {4 ch of the even K-step, 4 ch of the odd one}, lane = (channel half) * 32 + (co & 31),
// tile = (co & 63) >> 5; tail[0] = 1 / s (f32), tail[1] max |w|'s bits (k_absmax_bits, wino_h2.hip).
__global__ __launch_bounds__(256) void k_wino_pack_h3(const float* __restrict__ w, unsigned short* __restrict__ out, float* __restrict__ tail,
                                                      int Cout, int Cin) {
    const float wmax = __builtin_bit_cast(float, reinterpret_cast<const unsigned*>(tail)[1]);
    int ex = 0;
    if (wmax > 0.f && wmax < 3.0e38f) { (void)frexpf(2.25f * wmax, &ex); ex = 13 - ex; }
    ex = max(-100, min(100, ex));
    const float sc = ldexpf(1.0f, ex);
    if (blockIdx.x == 0 && threadIdx.x == 0) tail[0] = ldexpf(1.0f, -ex);
    const long long total = (long long)Cout * Cin;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(g % Cin), co = (int)(g / Cin);
        const float* k = w + ((size_t)co * Cin + ci) * 9;
        float gg[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g0 = k[c] * sc, g1 = k[3 + c] * sc, g2 = k[6 + c] * sc;
            gg[0][c] = g0;
            gg[1][c] = 0.5f * (g0 + g1 + g2);
            gg[2][c] = 0.5f * (g0 - g1 + g2);
            gg[3][c] = g2;
        }
        const int nb = co >> 6, col = co & 63, nt = col >> 5, kb = ci >> 3, cil = ci & 7, e = cil & 3;
        const int ln = (cil >> 2) * 32 + (col & 31);
        unsigned short* img = out + ((size_t)nb * (Cin >> 4) + (kb >> 1)) * (kPairBytes / 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float r0 = gg[i][0], r1 = gg[i][1], r2 = gg[i][2];
            const float u[4] = {r0, 0.5f * (r0 + r1 + r2), 0.5f * (r0 - r1 + r2), r2};
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                const float x = u[jx];
                const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(x, 0.f);
                const float r = x - (float)h[0];
                const fp16x2 h2 = __builtin_amdgcn_cvt_pkrtz(r, 0.f);
                unsigned short* frag = img + ((4 * i + jx) * 4096 + nt * 2048 + ln * 16) / 2;
                frag[(kb & 1) * 4 + e] = (unsigned short)(__builtin_bit_cast(unsigned, h) & 0xFFFFu);
                frag[512 + (kb & 1) * 4 + e] = (unsigned short)(__builtin_bit_cast(unsigned, h2) & 0xFFFFu);
            }
        }
    }
}

// .w = the k_wino_pack_h3 image (+ its tail), tby = ceil(ceil(H / 2) / 8): 8 x 8 tile patches
int launch_conv_wino_h3(const WinoArgs& a, int groups, hipStream_t s) {
    if (groups < 1 || groups > kMaxGroup || a.Cin % 16 != 0 || a.Cout % kBN != 0) return FPC_EINVAL;      // pairs of 8-channel K-steps
    if ((long long)a.H * a.W * a.Cin * (long long)sizeof(float) >= (1LL << 32)) return FPC_EINVAL;      // 32-bit lane offsets inside one image
    if ((long long)(a.Cin >> 4) * kPairBytes >= (1LL << 31)) return FPC_EINVAL;                          // 31-bit buffer offsets inside one block's images
    if (a.tbx != cdiv(cdiv(a.W, 2), kTX) || a.tby != cdiv(cdiv(a.H, 2), kTY)) return FPC_EINVAL;
    const long long nblk = (long long)a.tbx * a.tby * a.B * (a.Cout / kBN) * groups;
    if (nblk < 1 || nblk >= (1LL << 31)) return FPC_EINVAL;
    static const int var = getenv("FPC_H3_VAR") ? atoi(getenv("FPC_H3_VAR")) : 0;      // diagnostic
    if (var == 1) hipLaunchKernelGGL(k_conv_wino_h3<1>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_conv_wino_h3<0>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    return check_launch();
}

// pair-order fp16 x 2 image: 16 * Cout * Cin floats + a tail of 2 (1 / scale, max |w| bits); every byte is written
int launch_wino_pack_h3(const float* w_oihw, float* packed, int Cout, int Cin, hipStream_t s) {
    if (Cin % 16 != 0 || Cout % kBN != 0) return FPC_EINVAL;
    float* tail = packed + (size_t)16 * Cout * Cin;
    if (hipMemsetAsync(tail, 0, 2 * sizeof(float), s) != hipSuccess) return FPC_ELAUNCH;
    if ((uintptr_t)w_oihw & 15) return FPC_EINVAL;
    const int rc = launch_absmax_bits(w_oihw, (long long)Cout * Cin * 9, reinterpret_cast<unsigned*>(tail) + 1, s);
    if (rc) return rc;
    const long long work = (long long)Cout * Cin;
    hipLaunchKernelGGL(k_wino_pack_h3, dim3((unsigned)std::min<long long>((work + 255) / 256, 4096)), dim3(256), 0, s, w_oihw,
                       reinterpret_cast<unsigned short*>(packed), tail, Cout, Cin);
    return check_launch();
}

}  // namespace fpc
