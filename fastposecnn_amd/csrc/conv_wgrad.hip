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
// One workgroup = one (tap, 64 output channels, 64 input channels) tile over a slice of the pixels (split-K: the early
// layers have 9 tiles and 150 000 pixels); k_wgrad_reduce sums the slices in slice order and writes OIHW.
// 256 threads = 2 x 2 waves, a 32 x 32 block of the tile each.  Double-buffered LDS, the next K-step's global loads in
// flight under the current step's MFMAs, one barrier per step.
#include <algorithm>

#include "common.hpp"

namespace fpc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWgBK = 32;         // pixels per K-step
constexpr int kWgRow = 68;        // floats per LDS row (64 + 4: consecutive pixels start 4 banks apart)

struct WgradArgs {
    const float* x;       // input activation, channel-last (element strides sb, sh, sw; channel stride 1)
    const float* dy;      // output gradient, NHWC contiguous [B, Ho, Wo, Cout]
    float* part;          // [nsplit][taps][Cout][Cin] partial sums
    long long sb, sh, sw;
    int B, Hi, Wi, Cin, Ho, Wo, Cout, Kh, Kw, stride, pad;
    int mtiles, ntiles, nsplit, ksteps, per;      // per = K-steps per slice
};

__global__ __launch_bounds__(256, 2) void k_conv_wgrad(const WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][kWgBK * kWgRow];      // [buffer][A | B][pixel][channel]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    int bid = blockIdx.x;
    const int nt = bid % a.ntiles; bid /= a.ntiles;
    const int mt = bid % a.mtiles; bid /= a.mtiles;
    const int tap = bid % (a.Kh * a.Kw);
    const int sp = bid / (a.Kh * a.Kw);
    const int kh = tap / a.Kw, kw = tap - kh * a.Kw;
    const int m0 = mt * 64, n0 = nt * 64;
    const int HoWo = a.Ho * a.Wo;
    const long long P = (long long)a.B * HoWo;
    const int ks0 = sp * a.per, ks1 = min(a.ksteps, ks0 + a.per);

    // staging: thread (r, q) owns pixels r and r + 16 of the K-step and the float4 at channel 4q of both operands
    const int r = t >> 4, q = t & 15;
    const bool a_col = m0 + 4 * q < a.Cout;        // Cout % 4 == 0: a float4 is inside or outside as a whole
    f32x4 ra[2], rb[2];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#define FPC_WG_LOAD(KS)                                                                                       \
    do {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                       \
            const long long p = (long long)(KS) * kWgBK + r + 16 * i;                                         \
            ra[i] = zero; rb[i] = zero;                                                                       \
            if (p < P) {                                                                                      \
                if (a_col) ra[i] = *reinterpret_cast<const f32x4*>(a.dy + p * a.Cout + m0 + 4 * q);           \
                const int b = (int)(p / HoWo), rem = (int)(p - (long long)b * HoWo);                          \
                const int ho = rem / a.Wo, wo = rem - ho * a.Wo;                                              \
                const int hi = ho * a.stride + kh - a.pad, wi = wo * a.stride + kw - a.pad;                   \
                if (hi >= 0 && hi < a.Hi && wi >= 0 && wi < a.Wi)                                             \
                    rb[i] = *reinterpret_cast<const f32x4*>(a.x + b * a.sb + hi * a.sh + wi * a.sw + n0 + 4 * q); \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)
#define FPC_WG_STORE(BUF)                                                                                     \
    do {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                       \
            *reinterpret_cast<f32x4*>(&lds[BUF][0][(r + 16 * i) * kWgRow + 4 * q]) = ra[i];                   \
            *reinterpret_cast<f32x4*>(&lds[BUF][1][(r + 16 * i) * kWgRow + 4 * q]) = rb[i];                   \
        }                                                                                                     \
    } while (0)

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    if (ks0 < ks1) {
        FPC_WG_LOAD(ks0);
        FPC_WG_STORE(0);
    }
    __syncthreads();
    for (int ks = ks0; ks < ks1; ++ks) {
        const int buf = (ks - ks0) & 1;
        if (ks + 1 < ks1) FPC_WG_LOAD(ks + 1);
        const float* As = &lds[buf][0][lh * kWgRow + wm * 32 + li];
        const float* Bs = &lds[buf][1][lh * kWgRow + wn * 32 + li];
        float fa[16], fb[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) { fa[kk] = As[2 * kk * kWgRow]; fb[kk] = Bs[2 * kk * kWgRow]; }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk], fb[kk], acc, 0, 0, 0);
        if (ks + 1 < ks1) FPC_WG_STORE(buf ^ 1);
        __syncthreads();
    }
#undef FPC_WG_LOAD
#undef FPC_WG_STORE

    // C/D layout: column (input channel) = lane & 31, row (output channel) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* out = a.part + ((size_t)sp * a.Kh * a.Kw + tap) * ((size_t)a.Cout * a.Cin);
    const int ci = n0 + wn * 32 + li;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int co = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
        if (co < a.Cout) out[(size_t)co * a.Cin + ci] = acc[i];
    }
}

// dw[co][ci][tap] = sum over slices (in slice order) of part[slice][tap][co][ci]; one thread per (co, ci), all taps
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int nsplit,
                                                      int taps, int Cout, int Cin) {
    const size_t cc = (size_t)Cout * Cin;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= cc) return;
    for (int tap = 0; tap < taps; ++tap) {
        const float* src = part + (size_t)tap * cc + i;
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += src[(size_t)sp * taps * cc];
        dw[i * taps + tap] = s;
    }
}

static void wgrad_plan(int B, int Ho, int Wo, int Cin, int Cout, int taps, WgradArgs& a) {
    a.mtiles = cdiv(Cout, 64);
    a.ntiles = Cin / 64;
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
extern "C" int fpc_conv2d_wgrad(const float* x, int64_t sb, int64_t sh, int64_t sw, const float* dy, float* dw, int B, int Hi,
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
    hipLaunchKernelGGL(k_conv_wgrad, dim3((unsigned)grid), dim3(256), 0, s, a);
    int rc = check_launch();
    if (rc) return rc;
    const size_t cc = (size_t)Cout * Cin;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((cc + 255) / 256)), dim3(256), 0, s, (const float*)ws, dw, a.nsplit, Kh * Kw,
                       Cout, Cin);
    return check_launch();
}
