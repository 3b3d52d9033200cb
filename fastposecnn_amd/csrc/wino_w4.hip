// wino_w4.hip — k_conv_wino_w4: the split-precision Winograd F(2x2, 3x3) convolution as FOUR waves of 512 registers (round 6).
//
// k_conv_wino<8, ..., BF3> (net_kernels.hip: 8 x 8 tile patch x 64 output channels, 8 waves = 2 per SIMD, 242 registers) spends 3 270
// cycles on a K-step that holds 1 536 cycles of matrix work: per wave 153 vector instructions beside 24 matrix instructions, and a
// wave's vector instructions crawl (one per matrix instruction) whenever its SIMD partner is the one issuing matrix instructions.
// The 128-channel shape (wino128.hip) halves the vector work per product but doubles the weight bytes per product and sits on the
// L2's ~18 TB/s (96 KB per K-step and CU: 2 400 cycles).  This kernel keeps the 64-channel shape's bytes and removes the PARTNER:
//
//   * ONE wave per SIMD (256 threads, launch bound 1 -> up to 512 registers per lane): wave w owns transform row w — its 4 xi x
//     all 64 tiles (two 32-tile halves) x 64 channels (two 32-channel tiles) = 16 accumulators of 32 x 32 = 256 accumulation
//     registers; 48 v_mfma_f32_32x32x16_bf16 per K-step of 8 input channels.  Every matrix instruction is followed by ITS share
//     (one item of 4-7 vector instructions) of the step's other work and a scheduling barrier: between a wave's own matrix
//     instructions a vector instruction costs its 4 issue cycles and nothing else, and 6-7 of them fit under the 32 cycles the
//     matrix pipe is busy.
//   * weights straight into the operand registers: a weight fragment is used by two matrix instructions of ONE wave, so LDS buys
//     nothing.  The image is k_wino_pack_bf3's, unchanged — its 16-byte {b1, b2} and 8-byte {b3} slots per (xi, channel, half) are
//     contiguous per wave: a lane offset picks the slot — fetched with buffer loads one K-step ahead into the registers the last
//     matrix instruction of the fragment has just read (48 registers of weights in flight or waiting; 48 KB per K-step and CU).
//   * input: the permuted, conflict-free 18 x 18 region image of the 8-wave kernel (18 LDS-DMA pieces of 1 KB per K-step, two
//     buffers, one barrier per K-step).  The DMA instructions are inline asm: issued BEFORE the step's 16 weight loads and waited
//     for with a COUNTED vmcnt(16), so the weight prefetch stays in flight across the barrier.
//   * the next step's fragments are transformed and split IN PLACE: the three bf16 pieces of xi j's fragments are overwritten
//     once xi j's matrix instructions are done (xi 0-2 during this step's xi 2-3, xi 3 — whose transformed values wait in eight
//     registers — during the next step's xi 0-1), so only one set of pieces (48 registers) exists.
//   Products, split and accumulation order per accumulator are the 8-wave BF3 form's: results are bit-identical to it.
//   Output transform through LDS in ONE pass (Z[row 4][cc 2][tile 64][channel 64] = 128 KB), epilogue as k_conv_wino's.
// Reference: the 3x3 / stride-1 convolutions of F/lib/pose_regressor.py:709-743 (smp encoder + FPN decoder, not vendored).
#include <algorithm>
#include <cstdlib>
#include "net_kernels.hpp"

namespace fpc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int kTX = 8, kTY = 8;                  // tile patch 8 x 8 (16 x 16 output pixels)
constexpr int kRW = 2 * kTX + 2, kRH = 2 * kTY + 2;      // staged input region 18 x 18
constexpr int kBN = 64;                          // output channels per workgroup
constexpr int kNT = kTX * kTY;                   // 64 tiles = two M halves
constexpr int kInPieces = 18;                    // 1 KB LDS-DMA pieces of one K-step's input image (k_conv_wino's permuted image)
constexpr int kInFloats = kInPieces * 256;       // 4608 floats per input buffer
constexpr int kStepBytes = 12288 * 4;            // k_wino_pack_bf3's image of one K-step: 32 KB {b1, b2} + 16 KB {b3}
constexpr int kLdsFloats = 4 * 2 * kNT * kBN;    // output transform image = 128 KB
static_assert(kLdsFloats >= 2 * kInFloats, "the K loop's two input buffers live in the output image's space");

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

// MODE (diagnostic instantiations, FPC_W4_MODE at launch): bit 0 = the K loop reloads no weights, bit 1 = it stages no input and has
// no barrier — wrong results, the same instruction stream otherwise
template <int MODE>
__global__ __launch_bounds__(256, 1) void k_conv_wino_w4(const WinoArgs a) {
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

    f32x16 acc[4][2][2];      // [xi column j][tile half mt][32-channel tile nt]; zeroed while the first operands are on their way
    // ---- weights: buffer loads from k_wino_pack_bf3's image.  {b1, b2} of (xi, channel co, channel half hw): 16 bytes at
    // xi * 2048 + co * 32 + 16 * (hw ^ ((co >> 3) & 1)); {b3}: 8 bytes at 32768 + xi * 1024 + co * 16 + 8 * (hw ^ ((co >> 4) & 1))
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(reinterpret_cast<const char*>(P.w) + (size_t)nb * nkb * kStepBytes);
    const int vo_u = li * 32 + 16 * (lh ^ ((li >> 3) & 1)), vo_t = li * 16 + 8 * (lh ^ ((li >> 4) & 1));
    int so_u = wi * 8192, so_t = 32768 + wi * 4096;      // this wave's four xi; + kStepBytes per K-step
    u32x4 U[4][2];
    u32x2 T[4][2];
#define FPC_W4_LOAD_U(J, NT) U[J][NT] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, vo_u, so_u + (J) * 2048 + (NT) * 1024, 0))
#define FPC_W4_LOAD_T(J, NT) T[J][NT] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_w, vo_t, so_t + (J) * 1024 + (NT) * 512, 0))
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) { FPC_W4_LOAD_U(j, nt); FPC_W4_LOAD_T(j, nt); }
    if (nkb > 1) { so_u += kStepBytes; so_t += kStepBytes; }

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
#define FPC_W4_ISSUE_IN(BUF)                                                                                  \
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
                       "v"(ivo[0]), "v"(ivo[1]), "v"(ivo[2]), "v"(ivo[3]), "v"(ivo[4]), "s"(isb)              \
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

    // a patch that reaches over the image border zeroes both input buffers once (inactive DMA lanes leave them alone); an
    // interior patch rewrites every unit the fragment reads touch with every step's DMA
    if (y_in0 < 0 || x_in0 < 0 || y_in0 + kRH > H || x_in0 + kRW > W) {
        for (int i = t; i < 2 * kInFloats / 4; i += 256) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    FPC_W4_ISSUE_IN(0);
    if (nkb > 1) isb += 8;
    FPC_W4_ISSUE_IN(1);
    if (nkb > 2) isb += 8;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][mt][nt][r] = 0.f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // pieces of the current step's transformed fragments pa[j][mt][piece] (four channels each); transformed values vn[mt][j] of the
    // step whose pieces are being built (xi 3's wait there across the loop's back edge)
    u32x2 pa[4][2][3];
    f32x4 vn[2][4];
    {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 e[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                e[c] = fma_s4(sgn, *reinterpret_cast<const f32x4*>(lds + in_b[c >> 1] + (c & 1) * in_cs + mt * in_ms),
                              *reinterpret_cast<const f32x4*>(lds + in_a[c >> 1] + (c & 1) * in_cs + mt * in_ms));
            vn[mt][0] = sub_s4(e[0], e[2]); vn[mt][1] = add_s4(e[1], e[2]); vn[mt][2] = sub_s4(e[2], e[1]); vn[mt][3] = sub_s4(e[1], e[3]);
#pragma unroll
            for (int j = 0; j < 3; ++j) split_bf3(vn[mt][j], pa[j][mt][0], pa[j][mt][1], pa[j][mt][2]);
        }
    }
    __syncthreads();       // buffer 0 is refilled by step 0's DMA

#define FPC_W4_MFMA(J, MT, NT, A, B) acc[J][MT][NT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc[J][MT][NT], 0, 0, 0)
    // one item of a three-way split: element `part` of vn[MT][J] -> its two residuals; items 1 and 3 also pack the finished pair
// (Items are PINNED to their slot: plain arithmetic has no ordering against __builtin_amdgcn_sched_barrier before instruction
// selection, and the compiler gathered 44 instructions of two splits into one slot.  An empty volatile asm that READS an item's
// results at its end is ordered against the barriers: the item cannot sink below its slot.  Input-only on purpose: an asm with
// outputs in front of the item's first instruction drew a hazard s_nop — 4 issue cycles — per item.)
#define FPC_W4_SPLIT_ITEM(J, MT, PART)                                                                        \
    do {                                                                                                      \
        const float x_ = vn[MT][J][PART];                                                                     \
        const unsigned xb_ = __builtin_bit_cast(unsigned, x_) & 0xFFFF0000u;                                  \
        const float r_ = x_ - __builtin_bit_cast(float, xb_);                                                 \
        const unsigned rb_ = __builtin_bit_cast(unsigned, r_) & 0xFFFF0000u;                                  \
        const float q_ = r_ - __builtin_bit_cast(float, rb_);                                                 \
        if ((PART) & 1) {                                                                                     \
            const unsigned k1_ = pack_hi16(sx[(PART) - 1], x_), k2_ = pack_hi16(sr[(PART) - 1], r_), k3_ = pack_hi16(sq[(PART) - 1], q_); \
            asm volatile("" :: "v"(k1_), "v"(k2_), "v"(k3_));                                                 \
            pa[J][MT][0][(PART) >> 1] = k1_; pa[J][MT][1][(PART) >> 1] = k2_; pa[J][MT][2][(PART) >> 1] = k3_; \
        } else {                                                                                              \
            asm volatile("" :: "v"(r_), "v"(q_));                                                             \
            sx[PART] = x_; sr[PART] = r_; sq[PART] = q_;                                                      \
        }                                                                                                     \
    } while (0)
#define FPC_W4_PIN4(V) asm volatile("" :: "v"(V))
    u32x4 Atup = {pa[0][0][0][0], pa[0][0][0][1], pa[0][0][0][0], pa[0][0][0][1]};      // operand of the first pair of slots (xi 0, a1 b1 + a1 b2, half 0)
    int cur = 0;
    const long long c_begin = a.dbg ? clock64() : 0, r_begin = a.dbg ? wall_clock64() : 0;
#pragma unroll 1
    for (int kb = 0; kb < nkb; ++kb) {
        // input of step kb + 2 -> the buffer step kb's fragments were read from during step kb - 1 (oldest in the queue: see the wait below)
        if (!(MODE & 2) && !(MODE & 8)) FPC_W4_ISSUE_IN(cur);
        const float* In = lds + (cur ^ 1) * kInFloats;
        f32x4 da[2][4], db[2][4], e[2][4];
        float sx[4], sr[4], sq[4];
        __builtin_amdgcn_s_setprio(1);
        // Slot sl = 12 j + 4 g + 2 mt + nt: g = 0: a1 b1 + a1 b2, g = 1: a2 b1 + a2 b2, g = 2: a1 b3 + a3 b1.  One item per slot:
        //   sl  0- 3  fragment reads of step kb + 1 (two columns of one tile half each) + split of THIS step's xi 3, half 0
        //   sl  4-11  row transform e = da + sgn db (one column of one half each)
        //   sl 12-15  split of this step's xi 3, half 1      sl 16-23  column transform vn (one xi of one half each)
        //   sl 24-35  split of step kb + 1's xi 0 (both halves), xi 1 half 0      sl 36-47  xi 1 half 1, xi 2 (both halves)
        // g = 1 slots of half 0 also build their tile's {b3, b1} operand; the last matrix instruction of a weight fragment (g = 1, half 1 /
        // g = 2, half 1) is followed by the fragment's reload for step kb + 1.
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 C[2];
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    // the A operand of this pair of slots was built (and pinned) during the previous pair's first slot: a tuple
                    // written right before the matrix instruction that reads it costs hazard s_nops (4 issue cycles each)
                    const u32x4 Ause = Atup;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int sl = 12 * j + 4 * g + 2 * mt + nt;
                        if (g < 2) FPC_W4_MFMA(j, mt, nt, Ause, U[j][nt]);
                        if (g == 1) {
                            if (mt == 0) C[nt] = u32x4{T[j][nt][0], T[j][nt][1], U[j][nt][0], U[j][nt][1]};
                            if (mt == 1 && !(MODE & 1)) FPC_W4_LOAD_U(j, nt);
                        }
                        if (g == 2) {
                            FPC_W4_MFMA(j, mt, nt, Ause, C[nt]);
                            if (mt == 1 && !(MODE & 1)) FPC_W4_LOAD_T(j, nt);
                        }
                        if (nt == 0) {      // the next pair's operand (the loop's last pair builds the next step's first)
                            const int pr = (6 * j + 2 * g + mt + 1) % 24, nj = pr / 6, ng = (pr % 6) >> 1, nm = pr & 1;
                            const u32x2 lo = pa[nj][nm][ng == 1 ? 1 : 0], hi = pa[nj][nm][ng == 0 ? 0 : (ng == 1 ? 1 : 2)];
                            Atup = u32x4{lo[0], lo[1], hi[0], hi[1]};
                            FPC_W4_PIN4(Atup);
                        }
                        if (sl < 4) {
#pragma unroll
                            for (int c = 2 * (sl & 1); c < 2 * (sl & 1) + 2; ++c) {
                                da[sl >> 1][c] = *reinterpret_cast<const f32x4*>(In + in_a[c >> 1] + (c & 1) * in_cs + (sl >> 1) * in_ms);
                                db[sl >> 1][c] = *reinterpret_cast<const f32x4*>(In + in_b[c >> 1] + (c & 1) * in_cs + (sl >> 1) * in_ms);
                            }
                            FPC_W4_SPLIT_ITEM(3, 0, sl);
                        }
                        if (sl >= 4 && sl < 12) {
                            const int m_ = (sl - 4) >> 2, c = (sl - 4) & 3;
                            e[m_][c] = fma_s4(sgn, db[m_][c], da[m_][c]);
                            FPC_W4_PIN4(e[m_][c]);
                        }
                        if (sl >= 12 && sl < 16) FPC_W4_SPLIT_ITEM(3, 1, sl - 12);
                        if (sl >= 16 && sl < 24) {
                            const int m_ = (sl - 16) >> 2, jx = (sl - 16) & 3;
                            if (jx == 0) vn[m_][0] = sub_s4(e[m_][0], e[m_][2]);
                            if (jx == 1) vn[m_][1] = add_s4(e[m_][1], e[m_][2]);
                            if (jx == 2) vn[m_][2] = sub_s4(e[m_][2], e[m_][1]);
                            if (jx == 3) vn[m_][3] = sub_s4(e[m_][1], e[m_][3]);
                            FPC_W4_PIN4(vn[m_][jx]);
                        }
                        if (sl >= 24) {      // 24 items: (xi 0, half 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1) x 4 parts
                            const int q = (sl - 24) >> 2;
                            FPC_W4_SPLIT_ITEM(q >> 1, q & 1, (sl - 24) & 3);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
        __builtin_amdgcn_s_setprio(0);
        so_u += kb + 2 < nkb ? kStepBytes : 0;
        so_t += kb + 2 < nkb ? kStepBytes : 0;
        isb += kb + 3 < nkb ? 8 : 0;
        // this wave's DMA pieces (issued before the step's 16 weight loads, which stay in flight) have landed
        if (!(MODE & 2)) {
            if (MODE & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            if (!(MODE & 4)) __syncthreads();                 // everybody's have; this step's fragment reads are done
        }
        cur ^= 1;
    }
#undef FPC_W4_MFMA
#undef FPC_W4_SPLIT_ITEM
#undef FPC_W4_PIN4
#undef FPC_W4_ISSUE_IN
#undef FPC_W4_LOAD_U
#undef FPC_W4_LOAD_T
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
                f32x4 val = rr == 0 ? z[0][cc] + z[1][cc] + z[2][cc] : z[1][cc] - z[2][cc] - z[3][cc];
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

// .w = the k_wino_pack_bf3 image (as variant 3 of launch_conv_wino), .waves = 8 (tby = ceil(ceil(H / 2) / 8): 8 x 8 tile patches)
int launch_conv_wino_w4(const WinoArgs& a, int groups, hipStream_t s) {
    if (groups < 1 || groups > kMaxGroup || a.Cin % 8 != 0 || a.Cout % kBN != 0) return FPC_EINVAL;
    if ((long long)a.H * a.W * a.Cin * (long long)sizeof(float) >= (1LL << 32)) return FPC_EINVAL;      // 32-bit lane offsets inside one image
    if ((long long)(a.Cin >> 3) * kStepBytes >= (1LL << 31)) return FPC_EINVAL;                          // 31-bit buffer offsets inside one block's images
    if (a.tbx != cdiv(cdiv(a.W, 2), kTX) || a.tby != cdiv(cdiv(a.H, 2), kTY)) return FPC_EINVAL;
    const long long nblk = (long long)a.tbx * a.tby * a.B * (a.Cout / kBN) * groups;
    if (nblk < 1 || nblk >= (1LL << 31)) return FPC_EINVAL;
    static const int mode = getenv("FPC_W4_MODE") ? atoi(getenv("FPC_W4_MODE")) : 0;      // diagnostic
    if (mode == 1) hipLaunchKernelGGL(k_conv_wino_w4<1>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    else if (mode == 2) hipLaunchKernelGGL(k_conv_wino_w4<2>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    else if (mode == 3) hipLaunchKernelGGL(k_conv_wino_w4<3>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    else if (mode == 5) hipLaunchKernelGGL(k_conv_wino_w4<5>, dim3((unsigned)nblk), dim3(256), 0, s, a);      // no weights, DMA, no barrier
    else if (mode == 9) hipLaunchKernelGGL(k_conv_wino_w4<9>, dim3((unsigned)nblk), dim3(256), 0, s, a);      // no weights, barrier, no DMA
    else if (mode == 4) hipLaunchKernelGGL(k_conv_wino_w4<4>, dim3((unsigned)nblk), dim3(256), 0, s, a);      // everything but the barrier
    else hipLaunchKernelGGL(k_conv_wino_w4<0>, dim3((unsigned)nblk), dim3(256), 0, s, a);
    return check_launch();
}

}  // namespace fpc
