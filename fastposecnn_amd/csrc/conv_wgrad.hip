// conv_wgrad.hip — weight gradient of a 2-D convolution on the f32 matrix cores, for the training step
// (BASELINE.json configs[4]; the reference leaves it to cuDNN through autograd: F/lib/pose_regressor.py:709-743 run
// under Lightning's backward).  The data gradient of a stride-1 convolution is the forward kernel on flipped weights
// (lib/train_conv.py); this file is the other half.
//
//   dW[co][ci][kh][kw] = sum over pixels p = (b, ho, wo) of  dY[p][co] * X[b][ho*s + kh - pad][wo*s + kw - pad][ci]
//
// As a GEMM: M = Cout, N = Cin (per kernel tap), K = B*Ho*Wo pixels.  Both operands are NHWC, i.e. K-major: a pixel's
// channels are one contiguous row, so a K-step of 32 pixels is staged as [32][64] rows in LDS exactly as it lies in
// memory, and the 32x32x2 MFMA operands (lane l: row l % 32 of the tile, k = l / 32) are read as 128-byte row segments.
// One workgroup = one (tap, 64 or 128 output channels, 64 or 128 input channels) tile over a slice of the pixels (split-K:
// the early layers have 9 tiles and 150 000 pixels); k_wgrad_reduce sums the slices in slice order and writes OIHW.
// 256 threads = 2 x 2 waves, one or two 32 x 32 blocks of the tile per wave and side.  Double-buffered LDS, the next K-step's global loads in
// flight under the current step's MFMAs, one barrier per step.
#include <algorithm>

#include "common.hpp"

namespace fpc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kWgBK = 32;         // pixels per K-step

struct WgradArgs {
    const float* x;       // input activation, channel-last (element strides sb, sh, sw; channel stride 1)
    const float* dy;      // output gradient, NHWC contiguous [B, Ho, Wo, Cout]
    float* part;          // [nsplit][taps][Cout][Cin] partial sums
    long long sb, sh, sw;
    int B, Hi, Wi, Cin, Ho, Wo, Cout, Kh, Kw, stride, pad;
    int bm, bn;                                   // tile: output channels x input channels (64 or 128 each)
    int mtiles, ntiles, nsplit, ksteps, per;      // per = K-steps per slice
};

// TM x TN blocks of 32 x 32 per wave: the workgroup tile is (64 TM) x (64 TN).  The 128-wide forms halve the LDS reads
// and the global bytes per MFMA; layers with 64 channels on a side keep the 64-wide form for that side.
// BF3: split-precision products (common.hpp: split_bf3).  The LDS image stays f32 [pixel][channel]; a lane gathers its
// eight pixels of a 16-pixel group with eight 4-byte reads of one column, splits them into three bf16 pieces and issues
// the six piece products on v_mfma_f32_32x32x16_bf16 (the same six, smallest first, as the forward kernels): 24 matrix
// instructions of 8 passes per 16 pixels and 2 x 2 blocks against 32 of 16 passes for the f32 form.
// (Round 3 also tried the pieces split ONCE by the staging thread into K-contiguous bf16 planes — one ds_read_b128 per
// fragment and plane, half the vector-ALU work: 6.4 ms per training step against 5.9 for this form.  Neither is bound by
// the matrix pipe: a 128 x 128 tile moves 1 byte per 32 flop and every tap re-reads both operands, ~13 TB/s from L2 at
// the split-precision matrix rate.  Sharing dy across the taps of a kernel row is what would help.  Also measured and not
// kept: all tiles of a pixel slice on one XCD (blockIdx remap: 5.91 ms, unchanged — the operands are not missing L2) and
// global loads leading by two K-steps through a second register set (spills at 256 VGPRs: 6.8 ms).  A split-precision
// step lasts 0.65 us, shorter than a load's latency, with 34 KB per workgroup in flight: the bound is bytes in flight.)
template <int TM, int TN, bool BF3>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad(const WgradArgs a) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int RA = BM + 4, RB = BN + 4;       // floats per LDS row: consecutive pixels start 4 banks apart
    constexpr int NA = BM / 32, NB = BN / 32;     // float4 loads per thread and K-step (32 pixels x BM / 4 float4 over 256 threads)
    __shared__ __attribute__((aligned(16))) float lds[2 * kWgBK * (RA + RB)];      // [buffer][A rows | B rows]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    int bid = blockIdx.x;
    const int nt = bid % a.ntiles; bid /= a.ntiles;
    const int mt = bid % a.mtiles; bid /= a.mtiles;
    const int tap = bid % (a.Kh * a.Kw);
    const int sp = bid / (a.Kh * a.Kw);
    const int kh = tap / a.Kw, kw = tap - kh * a.Kw;
    const int m0 = mt * BM, n0 = nt * BN;
    const int HoWo = a.Ho * a.Wo;
    const long long P = (long long)a.B * HoWo;
    const int ks0 = sp * a.per, ks1 = min(a.ksteps, ks0 + a.per);

    // staging: A rows are BM / 4 float4 wide: thread t owns float4 column qa = t % (BM / 4) of pixels ra0 + i * (1024 / BM)
    constexpr int QA = BM / 4, QB = BN / 4, SA = 256 / QA, SB = 256 / QB;
    const int qa = t % QA, ra0 = t / QA, qb = t % QB, rb0 = t / QB;
    const bool a_col = m0 + 4 * qa < a.Cout;        // Cout % 4 == 0: a float4 is inside or outside as a whole
    const bool b_col = n0 + 4 * qb < a.Cin;
    f32x4 ra[NA], rb[NB];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // pixel walk of the B rows: (image, row, column) of pixel ks0 * 32 + rb0 + SB * i, advanced by 32 pixels per K-step
    // with adds and compares (a division per row and step costs as many vector cycles as the step's MFMAs)
    int pb[NB], py[NB], px[NB];
    const int step_y = kWgBK / a.Wo, step_x = kWgBK - step_y * a.Wo;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const long long p = (long long)ks0 * kWgBK + rb0 + SB * i;
        pb[i] = (int)(p / HoWo);
        const int rem = (int)(p - (long long)pb[i] * HoWo);
        py[i] = rem / a.Wo; px[i] = rem - py[i] * a.Wo;
    }
    const float* a_ptr = a.dy + ((long long)ks0 * kWgBK + ra0) * a.Cout + m0 + 4 * qa;      // + SA * i rows; + 32 rows per step
    long long a_left = P - ((long long)ks0 * kWgBK + ra0);                                     // rows left from this thread's first
    // MUST be issued for consecutive K-steps ks0, ks0 + 1, ... (it advances the walk)
#define FPC_WG_LOAD(KS)                                                                                       \
    do {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < NA; ++i)                                                        \
            ra[i] = (SA * i < a_left && a_col) ? *reinterpret_cast<const f32x4*>(a_ptr + (long long)(SA * i) * a.Cout) : zero; \
        a_ptr += (long long)kWgBK * a.Cout; a_left -= kWgBK;                                                  \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                      \
            rb[i] = zero;                                                                                     \
            const int hi = py[i] * a.stride + kh - a.pad, wi = px[i] * a.stride + kw - a.pad;                 \
            if (pb[i] < a.B && b_col && hi >= 0 && hi < a.Hi && wi >= 0 && wi < a.Wi)                         \
                rb[i] = *reinterpret_cast<const f32x4*>(a.x + pb[i] * a.sb + hi * a.sh + wi * a.sw + n0 + 4 * qb); \
            px[i] += step_x; py[i] += step_y;                                                                 \
            if (px[i] >= a.Wo) { px[i] -= a.Wo; ++py[i]; }                                                    \
            while (py[i] >= a.Ho) { py[i] -= a.Ho; ++pb[i]; }                                                 \
        }                                                                                                     \
    } while (0)
#define FPC_WG_STORE(BUF)                                                                                     \
    do {                                                                                                      \
        float* As_ = lds + (BUF) * kWgBK * (RA + RB);                                                         \
        float* Bs_ = As_ + kWgBK * RA;                                                                        \
        _Pragma("unroll") for (int i = 0; i < NA; ++i)                                                        \
            *reinterpret_cast<f32x4*>(As_ + (ra0 + SA * i) * RA + 4 * qa) = ra[i];                            \
        _Pragma("unroll") for (int i = 0; i < NB; ++i)                                                        \
            *reinterpret_cast<f32x4*>(Bs_ + (rb0 + SB * i) * RB + 4 * qb) = rb[i];                            \
    } while (0)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (ks0 < ks1) {
        FPC_WG_LOAD(ks0);
        FPC_WG_STORE(0);
    }
    __syncthreads();
    for (int ks = ks0; ks < ks1; ++ks) {
        const int buf = (ks - ks0) & 1;
        if (ks + 1 < ks1) FPC_WG_LOAD(ks + 1);
        if constexpr (BF3) {
            const float* As = lds + buf * kWgBK * (RA + RB) + 8 * lh * RA + wm * (BM / 2) + li;
            const float* Bs = lds + buf * kWgBK * (RA + RB) + kWgBK * RA + 8 * lh * RB + wn * (BN / 2) + li;
#pragma unroll
            for (int k16 = 0; k16 < kWgBK; k16 += 16) {
                bf16x8 FA[3][TM], FB[3][TN];
#pragma unroll
                for (int i = 0; i < TM + TN; ++i) {
                    const float* src = i < TM ? As + k16 * RA + 32 * i : Bs + k16 * RB + 32 * (i - TM);
                    const int R = i < TM ? RA : RB;
                    f32x4 lo, hi;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) { lo[kk] = src[kk * R]; hi[kk] = src[(4 + kk) * R]; }
                    u32x2 l1, l2, l3, h1, h2, h3;
                    split_bf3(lo, l1, l2, l3);
                    split_bf3(hi, h1, h2, h3);
                    const bf16x8 q1 = __builtin_bit_cast(bf16x8, u32x4{l1.x, l1.y, h1.x, h1.y});
                    const bf16x8 q2 = __builtin_bit_cast(bf16x8, u32x4{l2.x, l2.y, h2.x, h2.y});
                    const bf16x8 q3 = __builtin_bit_cast(bf16x8, u32x4{l3.x, l3.y, h3.x, h3.y});
                    if (i < TM) { FA[0][i < TM ? i : 0] = q1; FA[1][i < TM ? i : 0] = q2; FA[2][i < TM ? i : 0] = q3; }
                    else { FB[0][i < TM ? 0 : i - TM] = q1; FB[1][i < TM ? 0 : i - TM] = q2; FB[2][i < TM ? 0 : i - TM] = q3; }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[2][i], FB[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][i], FB[2][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1][i], FB[1][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1][i], FB[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][i], FB[1][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][i], FB[0][j], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
        const float* As = lds + buf * kWgBK * (RA + RB) + lh * RA + wm * (BM / 2) + li;
        const float* Bs = lds + buf * kWgBK * (RA + RB) + kWgBK * RA + lh * RB + wn * (BN / 2) + li;
        // 4 pixel pairs at a time: their fragments are read before the MFMAs that use them are issued
#pragma unroll
        for (int k4 = 0; k4 < 16; k4 += 4) {
            float fa[4][TM], fb[4][TN];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[kk][i] = As[2 * (k4 + kk) * RA + 32 * i];
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[kk][j] = Bs[2 * (k4 + kk) * RB + 32 * j];
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
        }
        }
        if (ks + 1 < ks1) FPC_WG_STORE(buf ^ 1);
        __syncthreads();
    }
#undef FPC_WG_LOAD
#undef FPC_WG_STORE

    // C/D layout: column (input channel) = lane & 31, row (output channel) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* out = a.part + ((size_t)sp * a.Kh * a.Kw + tap) * ((size_t)a.Cout * a.Cin);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ci = n0 + wn * (BN / 2) + 32 * j + li;
            if (ci >= a.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm * (BM / 2) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < a.Cout) out[(size_t)co * a.Cin + ci] = acc[i][j][r];
            }
        }
}

// dw[co][ci][tap] = sum over slices (in slice order) of part[slice][tap][co][ci]; one thread per (tap, co, ci), four
// independent partial sums per thread would change the order: the loop is kept sequential (deterministic, and the slices
// are few once the tiles are large)
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int nsplit,
                                                      int taps, int Cout, int Cin) {
    const size_t cc = (size_t)Cout * Cin;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int tap = blockIdx.y;
    if (i >= cc) return;
    const float* src = part + (size_t)tap * cc + i;
    const size_t step = (size_t)taps * cc;
    float s = 0.f;
    int sp = 0;
    for (; sp + 4 <= nsplit; sp += 4) {          // four loads in flight, summed in slice order
        const float v0 = src[(size_t)sp * step], v1 = src[(size_t)(sp + 1) * step], v2 = src[(size_t)(sp + 2) * step],
                    v3 = src[(size_t)(sp + 3) * step];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; sp < nsplit; ++sp) s += src[(size_t)sp * step];
    dw[i * taps + tap] = s;
}

static void wgrad_plan(int B, int Ho, int Wo, int Cin, int Cout, int taps, WgradArgs& a) {
    a.bm = Cout > 64 ? 128 : 64;
    a.bn = Cin > 64 ? 128 : 64;
    a.mtiles = cdiv(Cout, a.bm);
    a.ntiles = cdiv(Cin, a.bn);
    const long long P = (long long)B * Ho * Wo;
    a.ksteps = (int)((P + kWgBK - 1) / kWgBK);
    const long long tiles = (long long)taps * a.mtiles * a.ntiles;
    long long ns = std::max<long long>(1, 1024 / tiles);
    ns = std::min<long long>(ns, std::max(1, a.ksteps / 8));
    ns = std::min<long long>(ns, 256);
    a.per = cdiv(a.ksteps, (int)ns);
    a.nsplit = cdiv(a.ksteps, a.per);
}

}  // namespace fpc

using namespace fpc;

// Scratch of fpc_conv2d_wgrad: the split-K partial sums.
extern "C" size_t fpc_conv2d_wgrad_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int Kh, int Kw) {
    if (B < 1 || Ho < 1 || Wo < 1 || Cin < 64 || Cout < 1 || Kh < 1 || Kw < 1) return 0;
    WgradArgs a{};
    wgrad_plan(B, Ho, Wo, Cin, Cout, Kh * Kw, a);
    return (size_t)a.nsplit * Kh * Kw * Cout * Cin * sizeof(float);
}

// dw (OIHW, contiguous, overwritten) of a convolution with input x [B, Hi, Wi, Cin] (channel-last: channel stride 1,
// element strides sb / sh / sw, 16-byte aligned rows) and output gradient dy [B, Ho, Wo, Cout] (NHWC contiguous).
// Requires Cin % 64 == 0 and Cout % 4 == 0 (FPC_EINVAL otherwise: the caller keeps another path for the stem and the
// odd-width heads).  Any stride / padding.  Deterministic: fixed slice order.
static int conv2d_wgrad(bool split, const float* x, int64_t sb, int64_t sh, int64_t sw, const float* dy, float* dw, int B, int Hi,
                        int Wi, int Cin, int Cout, int Kh, int Kw, int stride, int pad, void* ws, size_t ws_bytes,
                        fpc_stream_t stream) {
    if (!x || !dy || !dw || !ws || B < 1 || Kh < 1 || Kw < 1 || stride < 1 || pad < 0) return FPC_EINVAL;
    if (Cin % 64 != 0 || Cout % 4 != 0 || Cout < 4) return FPC_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || (sb & 3) || (sh & 3) || (sw & 3)) return FPC_EINVAL;
    const int Ho = (Hi + 2 * pad - Kh) / stride + 1, Wo = (Wi + 2 * pad - Kw) / stride + 1;
    if (Ho < 1 || Wo < 1) return FPC_EINVAL;
    WgradArgs a{};
    a.x = x; a.dy = dy; a.part = (float*)ws; a.sb = sb; a.sh = sh; a.sw = sw;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.Kh = Kh; a.Kw = Kw;
    a.stride = stride; a.pad = pad;
    wgrad_plan(B, Ho, Wo, Cin, Cout, Kh * Kw, a);
    if (ws_bytes < fpc_conv2d_wgrad_workspace_bytes(B, Ho, Wo, Cin, Cout, Kh, Kw) || ((uintptr_t)ws & 15)) return FPC_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const long long grid = (long long)a.nsplit * Kh * Kw * a.mtiles * a.ntiles;
    if (grid > 0x7FFFFFFFLL) return FPC_EINVAL;
#define FPC_WG_LAUNCH(TM_, TN_)                                                                               \
    do {                                                                                                      \
        if (split) hipLaunchKernelGGL((k_conv_wgrad<TM_, TN_, true>), dim3((unsigned)grid), dim3(256), 0, s, a);  \
        else hipLaunchKernelGGL((k_conv_wgrad<TM_, TN_, false>), dim3((unsigned)grid), dim3(256), 0, s, a);   \
    } while (0)
    if (a.bm == 128 && a.bn == 128) FPC_WG_LAUNCH(2, 2);
    else if (a.bm == 128) FPC_WG_LAUNCH(2, 1);
    else if (a.bn == 128) FPC_WG_LAUNCH(1, 2);
    else FPC_WG_LAUNCH(1, 1);
#undef FPC_WG_LAUNCH
    int rc = check_launch();
    if (rc) return rc;
    const size_t cc = (size_t)Cout * Cin;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((cc + 255) / 256), (unsigned)(Kh * Kw)), dim3(256), 0, s, (const float*)ws, dw,
                       a.nsplit, Kh * Kw, Cout, Cin);
    return check_launch();
}

extern "C" int fpc_conv2d_wgrad(const float* x, int64_t sb, int64_t sh, int64_t sw, const float* dy, float* dw, int B, int Hi,
                                int Wi, int Cin, int Cout, int Kh, int Kw, int stride, int pad, void* ws, size_t ws_bytes,
                                fpc_stream_t stream) {
    return conv2d_wgrad(false, x, sb, sh, sw, dy, dw, B, Hi, Wi, Cin, Cout, Kh, Kw, stride, pad, ws, ws_bytes, stream);
}

// The same with split-precision matrix products (three bf16 pieces per f32 operand, six piece products accumulated in
// f32: include/fpc.h "split precision").  Same workspace, same determinism; results differ from fpc_conv2d_wgrad in the
// last bits only (both are within f32 summation error of the exact sums).
extern "C" int fpc_conv2d_wgrad_split(const float* x, int64_t sb, int64_t sh, int64_t sw, const float* dy, float* dw, int B,
                                      int Hi, int Wi, int Cin, int Cout, int Kh, int Kw, int stride, int pad, void* ws,
                                      size_t ws_bytes, fpc_stream_t stream) {
    return conv2d_wgrad(true, x, sb, sh, sw, dy, dw, B, Hi, Wi, Cin, Cout, Kh, Kw, stride, pad, ws, ws_bytes, stream);
}
