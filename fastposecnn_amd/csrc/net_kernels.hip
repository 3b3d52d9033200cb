// net_kernels.hip — device kernels of the backbone engine (encoder + FPN decoders + heads),
// replacing the torch/cuDNN op chain of PoseRegressor.pure_model_forward
// (F/lib/pose_regressor.py:709-743; encoder / FPNDecoder / SegmentationHead come from
// segmentation_models_pytorch, call sites :608-666) for inference.
//
//   k_conv_igemm        implicit-GEMM convolution on the f32 matrix cores
//                       (v_mfma_f32_32x32x2_f32: exact f32 FMA chains, no reduced precision).
//                       M = output pixels of one image, N = output channels, K = (kh, kw, ci).
//                       Activations are NHWC so a K-step of 32 channels is one 128-byte row per
//                       output pixel; weights are pre-packed OHWI [Npad][Kpad].  256 threads =
//                       2x2 wave64, each wave owns a (BM/2)x(BN/2) block of 32x32 MFMA tiles.
//                       Global -> registers -> LDS (double buffered, rows padded to 36 floats:
//                       conflict-free ds_read_b128 fragments), one barrier per K-step; the next
//                       K-step's global loads are in flight under the current step's MFMAs.
//                       Epilogue: folded BatchNorm / bias, residual, FPN nearest-x2 add, ReLU,
//                       GroupNorm partial sums.  Split-K writes raw partials instead.
//   k_conv_splitk_epilogue  fixed-order sum of the split-K partials + the same epilogue.
//   k_maxpool3x3s2, k_gn_finalize, k_gn_relu_up2, k_merge_head, k_up4_compress: HBM-bound
//                       streaming kernels, lanes along the contiguous (channel or x) axis.
#include "net_kernels.hpp"

namespace fpc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers (HIP's float4 struct copies can land in scratch)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// split_bf3 / pack_hi16: common.hpp (shared with conv_wgrad.hip)

// raw buffer descriptor over [base, base + 2 GB): offsets are 32-bit, an offset >= 2^31 reads zeros without
// touching memory (measured, tools_dev/dma_vs_mfma.hip) — the zero fill of the convolution padding
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}

// 16 bytes that another workgroup of the same launch reads (fused split-K): write-through stores / cache-bypassing loads
// at agent scope (sc1), so that neither side needs a whole-L2 write-back or invalidate; ordered by the drain + ticket
// of k_conv_igemm's epilogue.
typedef __attribute__((address_space(1))) unsigned long long gmem_u64;
__device__ __forceinline__ void store_wt128(float* p, f32x4 v) {
    const float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];      // (bit_cast of a vector ELEMENT expression misbehaves: scalars first)
    const unsigned long long lo = (unsigned long long)__builtin_bit_cast(unsigned, a0) | ((unsigned long long)__builtin_bit_cast(unsigned, a1) << 32);
    const unsigned long long hi = (unsigned long long)__builtin_bit_cast(unsigned, a2) | ((unsigned long long)__builtin_bit_cast(unsigned, a3) << 32);
    __hip_atomic_store((gmem_u64*)p, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((gmem_u64*)p + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x4 load_wt128(const float* p) {
    const unsigned long long lo = __hip_atomic_load((gmem_u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long hi = __hip_atomic_load((gmem_u64*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return f32x4{__builtin_bit_cast(float, (unsigned)lo), __builtin_bit_cast(float, (unsigned)(lo >> 32)),
                 __builtin_bit_cast(float, (unsigned)hi), __builtin_bit_cast(float, (unsigned)(hi >> 32))};
}

// a - b as two v_pk_add_f32 with negated second operand (the compiler splits a vector subtraction into four
// v_sub_f32).  Same IEEE result as the scalar subtraction.
__device__ __forceinline__ f32x4 sub_pk(f32x4 a, f32x4 b) {
    f32x2 al = __builtin_shufflevector(a, a, 0, 1), ah = __builtin_shufflevector(a, a, 2, 3);
    f32x2 bl = __builtin_shufflevector(b, b, 0, 1), bh = __builtin_shufflevector(b, b, 2, 3);
    f32x2 rl, rh;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(rl) : "v"(al), "v"(bl));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(rh) : "v"(ah), "v"(bh));
    return __builtin_shufflevector(rl, rh, 0, 1, 2, 3);
}

// Scalar forms for code that runs between bf16 matrix instructions: there a packed f32 instruction costs more than the two
// scalar ones it replaces (MI355X_MICROARCH.md, "price of one filler beside MFMAs"), and the SLP vectorizer would pack
// plain C++ arithmetic again — hence inline asm, one instruction per element.  Same IEEE results as the packed forms.
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
__device__ __forceinline__ f32x4 fma_s4(float s, f32x4 b, f32x4 a) {      // s * b + a, fused
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) { float x = b[k], y = a[k], z; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(z) : "v"(s), "v"(x), "v"(y)); r[k] = z; }
    return r;
}

// LDS operand rows are 32 floats (128 B) with NO padding: the 16-byte chunk c of row r lives in slot c ^ (r & 7), which
// keeps both the staging writes and the fragment reads conflict-free (8 consecutive rows hit 8 distinct slots of
// each half of the 64-bank window).  32 KB for the 64x64 tiling, 48 KB for 64x128 / 128x64 (36-float padded rows:
// 36.9 / 55.3 KB), i.e. room for 3 instead of 2 of the latter per CU.  (A fifth 64x64 workgroup per CU — registers
// squeezed to 96 — did not shorten the 1200-workgroup launches: their workgroups share the matrix pipe.)
constexpr int kLdsRow = kConvBK;


// ------------------------------------------------------------------------------------------
// implicit-GEMM convolution
// (FPC_IGEMM_DMA_B: net_kernels.hpp — fpc_conv2d's packing depends on it)
#ifndef FPC_IGEMM_PRIO
#define FPC_IGEMM_PRIO 1
#endif
#ifndef FPC_IGEMM_INTERLEAVE
#define FPC_IGEMM_INTERLEAVE 1
#endif

// One K-step of operands, global -> registers.  Thread (sr, sq) owns rows sr + 32*i and the float4 at
// column 4*sq of the 32-wide K-step.  (Macros, not functions: hipcc keeps by-reference register
// arrays in scratch.)
//
// Every load is a buffer_load (descriptor + 32-bit lane offset + scalar offset): beside a SIMD partner that
// issues MFMAs back to back a global_load with a 64-bit VGPR address waits like a vector-ALU instruction — one
// slot per MFMA, starved by a pure MFMA loop — while the buffer form issues in 9 cycles
// (tools_dev/dma_vs_mfma.hip).  MODE 0 (Cin % 32 == 0: a K-step is 32 channels of ONE tap) keeps the whole
// address generation on the scalar unit: per row a constant lane offset and an inverted validity mask over the
// taps (bit t = 1: tap t of this row is padding), so the zero fill is  offset | ((mask >> tap) << 31)  — two
// vector instructions per row and K-step; the tap walk (c0, kw, kh) advances with scalar compares.  MODE 0
// loads must be issued in K-step order (they are: ks0, ks0+1, ...).
#define FPC_CONV_LOAD(KS, ra, rb)                                                                                     \
    do {                                                                                                      \
        const int ks_ = (KS);                                                                                 \
        if (!DMAB) { _Pragma("unroll") for (int i = 0; i < BR; ++i) rb[i] = buf_load4(rs_w, wvo[i], ks_ * (kConvBK * 4)); } \
        if (MODE == 0) {                                                                                      \
            _Pragma("unroll") for (int i = 0; i < AR; ++i)                                                    \
                ra[i] = buf_load4(rs_in, ((anm[i] >> ld_tap) << 31) | avo[i], ld_soff);                       \
            ld_c0 += kConvBK; ld_soff += kConvBK * 4;                                                         \
            if (ld_c0 >= Cin) {                                                                               \
                ld_c0 = 0; ++ld_tap; ++ld_kw;                                                                 \
                if (ld_kw == Kw) { ld_kw = 0; ++ld_kh; }                                                      \
                ld_soff = (ld_kh * ish + ld_kw * isw) * 4;                                                    \
            }                                                                                                 \
        } else if (MODE == 2) {                                                                               \
            /* Cin % 4 == 0, channel-last: this lane's float4 is 4 channels of ONE tap (the 7x7 stem on */   \
            /* the NHWC4 image: 8 taps per K-step)                                                       */   \
            int kq = ks_ * kConvBK + 4 * sq;                                                                  \
            bool kv = kq < K;                                                                                 \
            int tap = kq / Cin, c0 = kq - tap * Cin;                                                          \
            int kh = tap / Kw, kw = tap - kh * Kw;                                                            \
            long long koff = (long long)kh * in_sh + (long long)kw * in_sw + c0;                              \
            _Pragma("unroll") for (int i = 0; i < AR; ++i) {                                                  \
                int hi = a_hi0[i] + kh, wi = a_wi0[i] + kw;                                                   \
                bool ok = kv && hi >= 0 && hi < Hi && wi >= 0 && wi < Wi;                                     \
                ra[i] = ok ? *reinterpret_cast<const f32x4*>(P.in + a_off[i] + koff)                          \
                           : f32x4{0.f, 0.f, 0.f, 0.f};                                                       \
            }                                                                                                 \
        } else {                                                                                              \
            /* any Cin / any input strides */                                                                 \
            _Pragma("unroll") for (int i = 0; i < AR; ++i) ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};           \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                   \
                int k = ks_ * kConvBK + 4 * sq + e;                                                           \
                bool kv = k < K;                                                                              \
                int tap = k / Cin, ci = k - tap * Cin;                                                        \
                int kh = tap / Kw, kw = tap - kh * Kw;                                                        \
                long long koff = (long long)kh * in_sh + (long long)kw * in_sw + (long long)ci * in_sc;       \
                _Pragma("unroll") for (int i = 0; i < AR; ++i) {                                              \
                    int hi = a_hi0[i] + kh, wi = a_wi0[i] + kw;                                               \
                    bool ok = kv && hi >= 0 && hi < Hi && wi >= 0 && wi < Wi;                                 \
                    float v = ok ? P.in[a_off[i] + koff] : 0.f;                                               \
                    if (e == 0) ra[i].x = v;                                                                  \
                    if (e == 1) ra[i].y = v;                                                                  \
                    if (e == 2) ra[i].z = v;                                                                  \
                    if (e == 3) ra[i].w = v;                                                                  \
                }                                                                                             \
            }                                                                                                 \
        }                                                                                                     \
    } while (0)

#define FPC_CONV_STORE(BUF, ra, rb)                                                                           \
    do {                                                                                                      \
        float* As_ = lds + (BUF) * (BM + BN) * kLdsRow;                                                       \
        float* Bs_ = As_ + BM * kLdsRow;                                                                      \
        _Pragma("unroll") for (int i = 0; i < AR; ++i)                                                        \
            *reinterpret_cast<f32x4*>(As_ + (sr + 32 * i) * kLdsRow + swz_w) = ra[i];                        \
        _Pragma("unroll") for (int i = 0; i < BR; ++i)                                                        \
            *reinterpret_cast<f32x4*>(Bs_ + (sr + 32 * i) * kLdsRow + swz_w) = rb[i];                        \
    } while (0)

// Split-precision staging: three bf16 planes per operand in LDS.  Plane rows are 32 bf16 = 64 bytes = four 16-byte chunks
// (one MFMA operand each); chunk c of row r sits in slot c ^ ((r >> 2) & 1) so that the eight rows a ds_read_b128 group
// touches hit eight different 16-byte positions of the bank window.  The ACTIVATION rows are split here, on the way from
// the f32 registers of one K-step; the WEIGHT rows arrive already split (k_pack_weight_bf3: three bf16 planes behind the
// f32 image) and go global -> LDS by LDS-DMA, FPC_CONV_DMA_B — no registers, no vector instructions, no ds_write.
#define FPC_CONV_STORE_BF3(BUF, ra, rb)                                                                       \
    do {                                                                                                      \
        char* st_ = reinterpret_cast<char*>(lds) + (BUF) * (BM + BN) * 192;                                   \
        _Pragma("unroll") for (int i = 0; i < AR + (DMAB ? 0 : BR); ++i) {                                    \
            const int row_ = (i < AR ? 0 : BM) + sr + 32 * (i < AR ? i : i - AR);                             \
            u32x2 p1_, p2_, p3_;                                                                              \
            split_bf3(i < AR ? ra[i < AR ? i : 0] : rb[i < AR ? 0 : i - AR], p1_, p2_, p3_);                  \
            char* d_ = st_ + row_ * 64 + bf3_w;                                                               \
            *reinterpret_cast<u32x2*>(d_) = p1_;                                                              \
            *reinterpret_cast<u32x2*>(d_ + (BM + BN) * 64) = p2_;                                             \
            *reinterpret_cast<u32x2*>(d_ + 2 * (BM + BN) * 64) = p3_;                                         \
        }                                                                                                     \
    } while (0)

// Epilogue of one lane's 4 rows (p0 + 8k) x 4 channels (n .. n+3) of a 32-row tile: folded BatchNorm / bias, residual,
// FPN nearest-x2 add, ReLU, store, and the GroupNorm partial sums of the tile's 32 rows per channel.
struct EpiGeom { int b, HoWo, Wo, Cout, Hu, Wu, P32, relu, lane; };
// The epilogue's own global reads of one lane's 4 rows x 4 channels (Cout % 4 == 0): scale / shift, residual, top-down addend.
// k_conv_igemm requests them BEFORE a tile's LDS transpose (round 5): issued inside conv_epilogue they sat behind the transpose's
// s_waitcnt (a memory clobber the compiler cannot move loads across), one exposed L2 round trip per 32 x 32 tile.  Absent
// operands come back as 1 / 0, so the epilogue uses them unconditionally.
struct EpiPre { f32x4 sc, sh, res[4], up[4]; };
__device__ __forceinline__ void conv_epilogue_prefetch(EpiPre& e, const ConvPtrs& P, const EpiGeom& g, int p0, int n) {
    e.sc = f32x4{1.f, 1.f, 1.f, 1.f}; e.sh = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) { e.res[k] = f32x4{0.f, 0.f, 0.f, 0.f}; e.up[k] = e.res[k]; }
    if ((g.Cout & 3) != 0 || n >= g.Cout) return;
    if (P.scale) e.sc = *reinterpret_cast<const f32x4*>(P.scale + n);
    if (P.shift) e.sh = *reinterpret_cast<const f32x4*>(P.shift + n);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = p0 + 8 * k;
        if (p >= g.HoWo) continue;
        if (P.res) e.res[k] = *reinterpret_cast<const f32x4*>(P.res + ((size_t)g.b * g.HoWo + p) * g.Cout + n);
        if (P.up) {
            const int ho = p / g.Wo, wo = p - ho * g.Wo;
            e.up[k] = *reinterpret_cast<const f32x4*>(P.up + (((size_t)g.b * g.Hu + (ho >> 1)) * g.Wu + (wo >> 1)) * g.Cout + n);
        }
    }
}
__device__ __forceinline__ void conv_epilogue(const ConvPtrs& P, const EpiGeom& g, f32x4 v0, f32x4 v1, f32x4 v2, f32x4 v3, int p0, int n,
                                              const EpiPre& pre) {
    const int b = g.b, HoWo = g.HoWo, Wo = g.Wo, Cout = g.Cout, Hu = g.Hu, Wu = g.Wu;
    const f32x4 v[4] = {v0, v1, v2, v3};
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if ((Cout & 3) == 0) {
        const bool nv = n < Cout;
        const f32x4 sc = pre.sc, sh = pre.sh;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int p = p0 + 8 * k;
            if (!(nv && p < HoWo)) continue;
            f32x4 x = v[k];
            if (P.scale) x = x * sc;
            x = x + sh;
            size_t o = ((size_t)b * HoWo + p) * Cout + n;
            if (P.res) x += pre.res[k];
            if (P.up) x += pre.up[k];
            if (g.relu) { x[0] = fmaxf(x[0], 0.f); x[1] = fmaxf(x[1], 0.f); x[2] = fmaxf(x[2], 0.f); x[3] = fmaxf(x[3], 0.f); }
            *reinterpret_cast<f32x4*>(P.out + o) = x;
            s1 += x;
            s2 += x * x;
        }
    } else {                                           // any Cout: scalar accesses
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int p = p0 + 8 * k;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int ne = n + e;
                if (!(ne < Cout && p < HoWo)) continue;
                float x = v[k][e];
                if (P.scale) x = x * P.scale[ne];
                if (P.shift) x = x + P.shift[ne];
                size_t o = ((size_t)b * HoWo + p) * Cout + ne;
                if (P.res) x += P.res[o];
                if (P.up) {
                    int ho = p / Wo, wo = p - ho * Wo;
                    x += P.up[(((size_t)b * Hu + (ho >> 1)) * Wu + (wo >> 1)) * Cout + ne];
                }
                if (g.relu) x = fmaxf(x, 0.f);
                P.out[o] = x;
                s1[e] += x;
                s2[e] += x * x;
            }
        }
    }
    if (P.gn_part) {
        // column sums over the tile's 32 rows: lanes with equal (lane & 7) hold the same channels
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[e] += __shfl_xor(s1[e], o, 64);
                s2[e] += __shfl_xor(s2[e], o, 64);
            }
        }
        if (g.lane < 8) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < Cout) {
                    float* gp = P.gn_part + (((size_t)b * g.P32 + p0 / 32) * Cout + n + e) * 2;
                    gp[0] = s1[e]; gp[1] = s2[e];
                }
        }
    }
}

// (the 128x128 tiling keeps 64 accumulator + 64 staging registers per lane: one workgroup per CU, no spills)
template <int BM, int BN, int MODE, bool BF3 = false>
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128) ? 1 : 2) void k_conv_igemm(const ConvArgs a) {
    constexpr int TM = BM / 64, TN = BN / 64;     // 32x32 tiles per wave
    constexpr int AR = BM / 32, BR = BN / 32;     // float4 rows staged per thread
    constexpr bool DMAB = BF3 && FPC_IGEMM_DMA_B; // split precision: weight planes pre-split, staged by LDS-DMA
    // f32 operands: 2 stages x (BM + BN) rows x 128 B;  split precision: 2 stages x 3 planes x (BM + BN) rows x 64 B
    __shared__ __attribute__((aligned(16))) float lds[BF3 ? 2 * (BM + BN) * 48 : 2 * (BM + BN) * kLdsRow];
    __shared__ int s_last;
    static_assert(!BF3 || MODE == 0, "split precision rides on the fast loader");
#ifdef FPC_STAMP_IGEMM      // diagnostic build (tools_dev/igemm_stamps.py): phase stamps per wave into a.dbg
    const long long st0 = clock64();
#endif

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int Cin = a.Cin, Kw = a.Kw, K = a.K, Hi = a.Hi, Wi = a.Wi, Wo = a.Wo, Cout = a.Cout, Npad = a.Npad;
    const int Kpad = a.Kpad, stride = a.stride, pad = a.pad, mtiles = a.mtiles, ntiles = a.ntiles, nB = a.B;
    const int nsplit = a.nsplit, ksteps = a.ksteps;
    const long long in_sb = a.in_sb, in_sh = a.in_sh, in_sw = a.in_sw, in_sc = a.in_sc;

    // Block order: the WEIGHT SLICE (group, n tile, k split) varies fastest, then the m tile.  Workgroups
    // are dealt round-robin over the 8 XCDs, so blockIdx % 8 — hence the weight slice, whenever the slice
    // count is 1, 2, 4 or a multiple of 8 — is fixed per XCD: an XCD's private L2 then holds ~1 MB of
    // weights that every one of its workgroups re-reads, instead of all slices thrashing all L2s
    // (measured before: 45-55 % L2 misses, ~5 TB/s of fabric reads for 85 MB of unique operands).
    int bid = blockIdx.x;
    const int sp = bid % nsplit; bid /= nsplit;
    const int nt = bid % ntiles; bid /= ntiles;
    const int grp = bid % a.groups; bid /= a.groups;
    const int mt = bid % mtiles;
    const int b = bid / mtiles;
    ConvPtrs P = a.p[0];
    if (grp == 1) P = a.p[1];
    if (grp == 2) P = a.p[2];
    if (grp == 3) P = a.p[3];
    const int HoWo = a.Ho * Wo;
    const int m0 = mt * BM, n0 = nt * BN;
    int ks0 = 0, ks1 = ksteps;
    if (nsplit > 1) {
        int per = (ksteps + nsplit - 1) / nsplit;
        ks0 = sp * per;
        ks1 = min(ksteps, ks0 + per);
    }

    const int sr = t >> 3, sq = t & 7;
    // swizzled 16-byte slot (in floats) of this thread's staging writes, and of its fragment reads per k-group:
    // lanes 0-31 carry k = kk*8 + e (chunk 2kk), lanes 32-63 k = kk*8 + 4 + e (chunk 2kk + 1); rows sr + 32i / li + 32i
    const int swz_w = 4 * (sq ^ (sr & 7));
    // split precision: byte offset inside a 64-byte plane row of this thread's four k (write) and of this lane's
    // eight k per 16-deep MFMA (read): chunk = k / 8, slot = chunk ^ ((row >> 2) & 1); rows sr + 32i / li + 32i
    const int bf3_w = (((sq >> 1) ^ ((sr >> 2) & 1)) << 4) + ((sq & 1) << 3);
    const int bf3_r[2] = {((0 + lh) ^ ((li >> 2) & 1)) << 4, ((2 + lh) ^ ((li >> 2) & 1)) << 4};
    const int swz_r[4] = {4 * ((0 + lh) ^ (li & 7)), 4 * ((2 + lh) ^ (li & 7)), 4 * ((4 + lh) ^ (li & 7)), 4 * ((6 + lh) ^ (li & 7))};
    long long a_off[AR];     // MODE 1, 2: element offset of (b, hi0, wi0, 0)
    int a_hi0[AR], a_wi0[AR];
    unsigned avo[AR], anm[AR];   // MODE 0: byte offset of (ho*stride, wo*stride, 4*sq) from the shifted base; padding mask
    const int ish = (int)in_sh, isw = (int)in_sw;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        int p = m0 + sr + 32 * i;
        bool ok = p < HoWo;
        int ho = ok ? p / Wo : 0, wo = ok ? p - ho * Wo : 0;
        a_hi0[i] = ok ? ho * stride - pad : -0x40000000;   // rows past the image: always out of bounds
        a_wi0[i] = wo * stride - pad;
        a_off[i] = (long long)b * in_sb + (long long)a_hi0[i] * in_sh + (long long)a_wi0[i] * in_sw;
        if (MODE == 0) {
            avo[i] = (unsigned)((ho * stride * ish + wo * stride * isw + 4 * sq) * 4);
            unsigned vw = 0, m = 0;                        // valid columns (bit kw), valid taps (bit kh*Kw + kw)
            const int lpx = a.lanepx ? sq : 0;             // lane-pixel layout: this lane's 16 bytes are pixel wi0 + sq
            for (int kw = 0; kw < Kw; ++kw) vw |= (unsigned)(a_wi0[i] + kw + lpx >= 0 && a_wi0[i] + kw + lpx < Wi) << kw;
            for (int kh = 0; kh < a.Kh; ++kh)
                if (ok && a_hi0[i] + kh >= 0 && a_hi0[i] + kh < Hi) m |= vw << (kh * Kw);
            anm[i] = ~m;
        }
    }
    // MODE 0: the descriptor starts `pad` rows and columns before the image (valid taps never reach below P.in)
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(P.in + (long long)b * in_sb - ((long long)pad * in_sh + (long long)pad * in_sw));
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(P.w);
    unsigned wvo[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) wvo[i] = (unsigned)(((n0 + sr + 32 * i) * Kpad + 4 * sq) * 4);
    // split precision: this wave's LDS-DMA pieces of a K-step's weight planes.  A piece = 16 rows x 64 B of one plane (1 KB,
    // lane l -> row l >> 2, slot l & 3); 3 * BN / 16 pieces per K-step, piece q = wave + 4 i is plane q / (BN / 16), row
    // group q % (BN / 16).  The swizzle is the choice of the chunk each lane fetches.  SGPR base + 32-bit lane offset, the
    // base advances 64 B per K-step (see the Winograd kernel for why not a 64-bit lane address).
    constexpr int kRG = BN / 16, kNPB = 3 * kRG / 4;
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    unsigned bvo[kNPB];
    const char* w3b = nullptr;
    if constexpr (DMAB) {
#pragma unroll
        for (int i = 0; i < kNPB; ++i) {
            const int q = swave + 4 * i, pl = q / kRG, rg = q - pl * kRG;
            const int r = rg * 16 + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 1);
            bvo[i] = (unsigned)((((size_t)pl * Npad + n0 + r) * Kpad + c * 8) * 2);
        }
        w3b = reinterpret_cast<const char*>(P.w + (size_t)Npad * Kpad) + (size_t)ks0 * (kConvBK * 2);
    }
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#define FPC_CONV_DMA_B(BUF)                                                                                   \
    do {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < kNPB; ++i) {                                                    \
            const int q_ = swave + 4 * i, pl_ = q_ / kRG, rg_ = q_ - pl_ * kRG;                               \
            asm volatile("s_mov_b32 m0, %0\n s_nop 0\n global_load_lds_dwordx4 %1, %2\n"                      \
                         :: "s"((unsigned)(size_t)(__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(lds) + \
                                 (BUF) * (BM + BN) * 192 + pl_ * (BM + BN) * 64 + (BM + rg_ * 16) * 64)),       \
                            "v"(bvo[i]), "s"(w3b) : "memory", "m0");                                          \
        }                                                                                                     \
        w3b += kConvBK * 2;                                                                                   \
    } while (0)
    // the pieces above have landed; the NYOUNG vector-memory operations issued after them may stay in flight
#define FPC_CONV_DMA_WAIT(YOUNGER)                                                                            \
    do {                                                                                                      \
        if (YOUNGER) { if constexpr (AR == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); } \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                 \
    } while (0)
    static_assert(AR == 2 || AR == 4, "FPC_CONV_DMA_WAIT counts AR activation loads");
    // tap walk of the NEXT K-step to load (scalar): channel offset, tap index and coordinates, byte offset of the tap
    int ld_c0, ld_tap, ld_kh, ld_kw, ld_soff;
    {
        int k0 = ks0 * kConvBK;
        ld_tap = k0 / Cin; ld_c0 = k0 - ld_tap * Cin;
        ld_kh = ld_tap / Kw; ld_kw = ld_tap - ld_kh * Kw;
        ld_soff = (ld_kh * ish + ld_kw * isw + ld_c0) * 4;
    }

    // two register sets: the loads of K-step k+2 are issued while step k is computed and step k+1
    // waits in registers, so every global load has two compute phases to land (HBM / L2 latency
    // under load exceeds one phase of 16..64 MFMAs)
    f32x4 ra0[AR], rb0[BR], ra1[AR], rb1[BR];
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Fragments of the next 8-deep k group are read from LDS BEFORE the current group's MFMAs are issued
    // (sched_barrier keeps hipcc from sinking the reads to their first use): the ~128-cycle LDS latency
    // is then hidden behind 8..32 MFMAs instead of stalling the wave four times per K-step.
#define FPC_CONV_FRAG(KK, FA, FB)                                                                             \
    do {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) FA[i] =                                                \
            *reinterpret_cast<const f32x4*>(As + i * 32 * kLdsRow + swz_r[KK]);                               \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) FB[j] =                                                \
            *reinterpret_cast<const f32x4*>(Bs + j * 32 * kLdsRow + swz_r[KK]);                               \
    } while (0)
    /* lanes 0-31 carry k = kk*8 + e, lanes 32-63 carry k = kk*8 + 4 + e: each MFMA sums two k */
#define FPC_CONV_MFMA(FA, FB)                                                                                 \
    do {                                                                                                      \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                         \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                    \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i][e], FB[j][e], acc[i][j], 0, 0, 0); \
    } while (0)
#define FPC_CONV_COMPUTE(BUF)                                                                                  \
    do {                                                                                                      \
        const float* As = lds + (BUF) * (BM + BN) * kLdsRow + (wm * (BM / 2) + li) * kLdsRow;                 \
        const float* Bs = lds + (BUF) * (BM + BN) * kLdsRow + BM * kLdsRow + (wn * (BN / 2) + li) * kLdsRow;  \
        f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];                                                             \
        FPC_CONV_FRAG(0, fa0, fb0);                                                                           \
        FPC_CONV_FRAG(1, fa1, fb1);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        FPC_CONV_MFMA(fa0, fb0);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        FPC_CONV_FRAG(2, fa0, fb0);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        FPC_CONV_MFMA(fa1, fb1);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        FPC_CONV_FRAG(3, fa1, fb1);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        FPC_CONV_MFMA(fa0, fb0);                                                                              \
        FPC_CONV_MFMA(fa1, fb1);                                                                              \
        __builtin_amdgcn_s_setprio(0);                                                                        \
    } while (0)

    // split precision: per 16-deep k group three planes per operand, six MFMAs per 32x32 tile
#define FPC_BF3_FRAG(KK, FA, FB)                                                                              \
    do {                                                                                                      \
        _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) {                                                    \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) FA[p_][i] = __builtin_bit_cast(bf16x8,             \
                *reinterpret_cast<const u32x4*>(Ab + p_ * (BM + BN) * 64 + i * 32 * 64 + bf3_r[KK]));         \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) FB[p_][j] = __builtin_bit_cast(bf16x8,             \
                *reinterpret_cast<const u32x4*>(Bb + p_ * (BM + BN) * 64 + j * 32 * 64 + bf3_r[KK]));         \
        }                                                                                                     \
    } while (0)
#define FPC_BF3_MFMA(FA, FB)                                                                                  \
    do {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                        \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                  \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[2][i], FB[0][j], acc[i][j], 0, 0, 0);  \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][i], FB[2][j], acc[i][j], 0, 0, 0);  \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1][i], FB[1][j], acc[i][j], 0, 0, 0);  \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1][i], FB[0][j], acc[i][j], 0, 0, 0);  \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][i], FB[1][j], acc[i][j], 0, 0, 0);  \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0][i], FB[0][j], acc[i][j], 0, 0, 0);  \
            }                                                                                                 \
    } while (0)
#define FPC_CONV_COMPUTE_BF3(BUF)                                                                             \
    do {                                                                                                      \
        const char* Ab = reinterpret_cast<const char*>(lds) + (BUF) * (BM + BN) * 192 + (wm * (BM / 2) + li) * 64;        \
        const char* Bb = reinterpret_cast<const char*>(lds) + (BUF) * (BM + BN) * 192 + (BM + wn * (BN / 2) + li) * 64;   \
        bf16x8 ga0[3][TM], gb0[3][TN], ga1[3][TM], gb1[3][TN];                                                \
        FPC_BF3_FRAG(0, ga0, gb0);                                                                            \
        FPC_BF3_FRAG(1, ga1, gb1);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        __builtin_amdgcn_s_setprio(FPC_IGEMM_PRIO);                                                                        \
        FPC_BF3_MFMA(ga0, gb0);                                                                               \
        FPC_BF3_MFMA(ga1, gb1);                                                                               \
        __builtin_amdgcn_s_setprio(0);                                                                        \
    } while (0)
    // Split precision with DMA-staged weights: the matrix block of LDS buffer BUF with the split of the NEXT step's
    // activation registers (ra -> the three planes of buffer BUF ^ 1) cut into 6 micro-steps per row — and, and-subtract,
    // and, subtract, pack, three 8-byte LDS stores — that are issued one after each MFMA (or every second one), in the matrix
    // instructions' shadow instead of as a block of 44 vector instructions + 6 ds_write_b64 behind them.  Runs unconditionally:
    // past the last step it splits stale registers into a buffer nobody reads (the epilogue's patches come after a barrier).
    // Not for the 128 x 128 tile: fully unrolled there the block needs more than 512 registers.
#define FPC_CONV_COMPUTE_STORE_BF3(BUF, ra)                                                                   \
    do {                                                                                                      \
        const char* Ab = reinterpret_cast<const char*>(lds) + (BUF) * (BM + BN) * 192 + (wm * (BM / 2) + li) * 64;        \
        const char* Bb = reinterpret_cast<const char*>(lds) + (BUF) * (BM + BN) * 192 + (BM + wn * (BN / 2) + li) * 64;   \
        char* st_ = reinterpret_cast<char*>(lds) + ((BUF) ^ 1) * (BM + BN) * 192;                             \
        bf16x8 ga[2][3][TM], gb[2][3][TN];                                                                    \
        FPC_BF3_FRAG(0, ga[0], gb[0]);                                                                        \
        FPC_BF3_FRAG(1, ga[1], gb[1]);                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        __builtin_amdgcn_s_setprio(FPC_IGEMM_PRIO);                                                                        \
        constexpr int kNM = 12 * TM * TN, kNS = 6 * AR;      /* MFMAs, split micro-steps */                    \
        constexpr int kPa[6] = {2, 0, 1, 1, 0, 0}, kPb[6] = {0, 2, 1, 0, 1, 0};                               \
        unsigned sb1_[AR][4], sb2_[AR][4];                                                                    \
        float sr_[AR][4], sq_[AR][4];                                                                         \
        u32x2 sp1_[AR], sp2_[AR], sp3_[AR];                                                                   \
        _Pragma("unroll") for (int m_ = 0; m_ < kNM; ++m_) {                                                  \
            const int kk_ = m_ / (6 * TM * TN), t_ = (m_ / 6) % (TM * TN), c_ = m_ % 6;                       \
            const int i_ = t_ / TN, j_ = t_ % TN;                                                             \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[kk_][kPa[c_]][i_], gb[kk_][kPb[c_]][j_], acc[i_][j_], 0, 0, 0); \
            _Pragma("unroll") for (int ns_ = m_ * kNS / kNM; ns_ < (m_ + 1) * kNS / kNM; ++ns_) {             \
                const int r_ = ns_ / 6, ph_ = ns_ % 6;                                                        \
                if (ph_ == 0) { _Pragma("unroll") for (int e = 0; e < 4; ++e) { const float x_ = ra[r_][e]; sb1_[r_][e] = __builtin_bit_cast(unsigned, x_) & 0xFFFF0000u; } } \
                if (ph_ == 1) { _Pragma("unroll") for (int e = 0; e < 4; ++e) { const float x_ = ra[r_][e]; sr_[r_][e] = x_ - __builtin_bit_cast(float, sb1_[r_][e]); } } \
                if (ph_ == 2) { _Pragma("unroll") for (int e = 0; e < 4; ++e) sb2_[r_][e] = __builtin_bit_cast(unsigned, sr_[r_][e]) & 0xFFFF0000u; } \
                if (ph_ == 3) { _Pragma("unroll") for (int e = 0; e < 4; ++e) sq_[r_][e] = sr_[r_][e] - __builtin_bit_cast(float, sb2_[r_][e]); } \
                if (ph_ == 4) {                                                                               \
                    const float x0_ = ra[r_][0], x1_ = ra[r_][1], x2_ = ra[r_][2], x3_ = ra[r_][3];           \
                    sp1_[r_] = u32x2{pack_hi16(x0_, x1_), pack_hi16(x2_, x3_)};                               \
                    sp2_[r_] = u32x2{pack_hi16(sr_[r_][0], sr_[r_][1]), pack_hi16(sr_[r_][2], sr_[r_][3])};   \
                    sp3_[r_] = u32x2{pack_hi16(sq_[r_][0], sq_[r_][1]), pack_hi16(sq_[r_][2], sq_[r_][3])};   \
                }                                                                                             \
                if (ph_ == 5) {                                                                               \
                    char* d_ = st_ + (sr + 32 * r_) * 64 + bf3_w;                                             \
                    *reinterpret_cast<u32x2*>(d_) = sp1_[r_];                                                 \
                    *reinterpret_cast<u32x2*>(d_ + (BM + BN) * 64) = sp2_[r_];                                \
                    *reinterpret_cast<u32x2*>(d_ + 2 * (BM + BN) * 64) = sp3_[r_];                            \
                }                                                                                             \
            }                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                \
        }                                                                                                     \
        __builtin_amdgcn_s_setprio(0);                                                                        \
    } while (0)
#define FPC_STORE_ANY(BUF, ra, rb) do { if (BF3) FPC_CONV_STORE_BF3(BUF, ra, rb); else FPC_CONV_STORE(BUF, ra, rb); } while (0)
#define FPC_COMPUTE_ANY(BUF) do { if (BF3) FPC_CONV_COMPUTE_BF3(BUF); else FPC_CONV_COMPUTE(BUF); } while (0)

    // Split precision: the weight planes of step k + 1 are DMA'd into the other LDS buffer at the start of step k (that buffer
    // was last read in step k - 1) and must have landed at the barrier that ends step k; the activation loads issued after
    // them (step k + 2) stay in flight across the barrier (counted vmcnt).
    if (ks0 < ks1) {
        if constexpr (DMAB) FPC_CONV_DMA_B(0);
        FPC_CONV_LOAD(ks0, ra0, rb0);
        FPC_STORE_ANY(0, ra0, rb0);
    }
    if (ks0 + 1 < ks1) FPC_CONV_LOAD(ks0 + 1, ra0, rb0);
    if constexpr (DMAB) FPC_CONV_DMA_WAIT(ks0 + 1 < ks1);
    __syncthreads();
#ifdef FPC_STAMP_IGEMM
    const long long st1 = clock64();
#endif
    for (int ks = ks0; ks < ks1; ks += 2) {
        // even phase: LDS buffer 0 holds step ks, set 0 holds ks+1
        if constexpr (DMAB) { if (ks + 1 < ks1) FPC_CONV_DMA_B(1); }
        if (ks + 2 < ks1) FPC_CONV_LOAD(ks + 2, ra1, rb1);
        if constexpr (DMAB && FPC_IGEMM_INTERLEAVE && BM * BN < 128 * 128) FPC_CONV_COMPUTE_STORE_BF3(0, ra0);
        else {
            FPC_COMPUTE_ANY(0);
            if (ks + 1 < ks1) FPC_STORE_ANY(1, ra0, rb0);
        }
        if constexpr (DMAB) FPC_CONV_DMA_WAIT(ks + 2 < ks1);
        __syncthreads();
        if (ks + 1 >= ks1) break;
        // odd phase: LDS buffer 1 holds step ks+1, set 1 holds ks+2
        if constexpr (DMAB) { if (ks + 2 < ks1) FPC_CONV_DMA_B(0); }
        if (ks + 3 < ks1) FPC_CONV_LOAD(ks + 3, ra0, rb0);
        if constexpr (DMAB && FPC_IGEMM_INTERLEAVE && BM * BN < 128 * 128) FPC_CONV_COMPUTE_STORE_BF3(1, ra1);
        else {
            FPC_COMPUTE_ANY(1);
            if (ks + 2 < ks1) FPC_STORE_ANY(0, ra1, rb1);
        }
        if constexpr (DMAB) FPC_CONV_DMA_WAIT(ks + 3 < ks1);
        __syncthreads();
    }
#pragma clang diagnostic pop

#ifdef FPC_STAMP_IGEMM
    const long long st2 = clock64();
#endif
    // C/D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5),
    // i.e. a lane holds 16 rows of ONE channel.  Each wave transposes its tile through a private LDS
    // patch (the operand buffers are free after the last barrier) so that a lane owns 4 consecutive
    // channels of 4 rows: every global access of the epilogue is then 16 bytes per lane and a row's
    // 32 channels are one 128-byte segment (scalar 4-byte stores made the 1x1 laterals store-issue bound).
    constexpr int kTS = 36;                                   // floats per transposed row
    float* tp = lds + wave * 32 * kTS;
    const int trow = lane >> 3, tc4 = (lane & 7) * 4;         // rows trow + 8k (k < 4), channels tc4 .. tc4+3
    const int Wu = Wo >> 1, Hu = a.Ho >> 1;
    // split-K: partials of split s at ws0 + s * ws_split; padded rows / columns exist
    const size_t ws_split = (size_t)nB * mtiles * BM * Npad;
    float* ws0 = nsplit > 1 ? a.splitk_ws + (((size_t)grp * nsplit * nB + b) * ((size_t)mtiles * BM)) * Npad : nullptr;
    float* ws = ws0 ? ws0 + sp * ws_split : nullptr;
    const bool fused = a.fused != 0;
    const EpiGeom eg{b, HoWo, Wo, Cout, Hu, Wu, mtiles * (BM / 32), a.relu, lane};
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            // (this tile's epilogue operands are requested first: they land while the tile goes through the LDS patch)
            EpiPre pre;
            if (!ws) conv_epilogue_prefetch(pre, P, eg, m0 + wm * (BM / 2) + i * 32 + trow, n0 + wn * (BN / 2) + j * 32 + tc4);
#pragma unroll
            for (int r = 0; r < 16; ++r) tp[((r & 3) + 8 * (r >> 2) + 4 * lh) * kTS + li] = acc[i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same wave: LDS ops complete in order
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(tp + (trow + 8 * k) * kTS + tc4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the next tile overwrites the patch
            const int prow = m0 + wm * (BM / 2) + i * 32;
            const int n = n0 + wn * (BN / 2) + j * 32 + tc4;
            if (ws) {                                          // split-K: raw partial sums
                if (fused) {                                   // read by another workgroup of this launch: write-through
#pragma unroll
                    for (int k = 0; k < 4; ++k) store_wt128(ws + (size_t)(prow + trow + 8 * k) * Npad + n, v[k]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        *reinterpret_cast<f32x4*>(ws + (size_t)(prow + trow + 8 * k) * Npad + n) = v[k];
                }
                continue;
            }
            conv_epilogue(P, eg, v[0], v[1], v[2], v[3], prow + trow, n, pre);
        }
    // Fused split-K: every workgroup has written its raw partial tile through to memory; it drains, and takes a ticket
    // of its output tile.  The one that draws the last ticket sums ALL partials in split order (its own included: the
    // sum does not depend on who arrives last; the same order as k_conv_splitk_epilogue) and applies the epilogue.
    if (ws && fused) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) {
            int* tk = a.tickets + (((size_t)grp * nB + b) * mtiles + mt) * ntiles + nt;
            const int got = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got == nsplit - 1) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the next launch
            s_last = got == nsplit - 1;
        }
        __syncthreads();
        if (s_last) {
#pragma unroll 1
            for (int ij = 0; ij < TM * TN; ++ij) {
                const int i = ij / TN, j = ij - i * TN;
                const int prow = m0 + wm * (BM / 2) + i * 32;
                const int n = n0 + wn * (BN / 2) + j * 32 + tc4;
                const float* src = ws0 + (size_t)(prow + trow) * Npad + n;
                EpiPre pre;
                conv_epilogue_prefetch(pre, P, eg, prow + trow, n);      // beside the partial sums' loads
                f32x4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = load_wt128(src + (size_t)(8 * k) * Npad);
#pragma unroll 2
                for (int q = 1; q < nsplit; ++q) {
                    f32x4 u[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) u[k] = load_wt128(src + q * ws_split + (size_t)(8 * k) * Npad);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += u[k];
                }
                conv_epilogue(P, eg, v[0], v[1], v[2], v[3], prow + trow, n, pre);
            }
        }
    }
#ifdef FPC_STAMP_IGEMM
    if (a.dbg && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long* o = (long long*)a.dbg + ((size_t)blockIdx.x * 4 + wave) * 4;
        o[0] = st1 - st0; o[1] = st2 - st1; o[2] = clock64() - st2; o[3] = ks1 - ks0;
    }
#endif
}

// Sums the split-K partials in split order and applies the epilogue.
// grid (P32 * nchunk, B, G): one workgroup = 32 rows x 128 channels; thread = 4 rows x 4 channels, so a
// split costs every thread four independent 16-byte loads (unrolled over splits for more in flight).
__global__ __launch_bounds__(256) void k_conv_splitk_epilogue(const ConvArgs a) {
    __shared__ float s_sum[8][128][2];
    ConvPtrs P = a.p[0];
    if (blockIdx.z == 1) P = a.p[1];
    if (blockIdx.z == 2) P = a.p[2];
    if (blockIdx.z == 3) P = a.p[3];
    const int t = threadIdx.x, rg = t >> 5, cq = t & 31;
    const int nchunk = (a.Cout + 127) >> 7;
    const int b = blockIdx.y, tile = blockIdx.x / nchunk, nc = (blockIdx.x - tile * nchunk) * 128;
    const int HoWo = a.Ho * a.Wo, Mp = a.mtiles * a.bm, nsplit = a.nsplit, Npad = a.Npad, Cout = a.Cout;
    const int Wu = a.Wo >> 1, Hu = a.Ho >> 1;
    const int P32 = a.mtiles * a.bm / 32;
    const int n = nc + 4 * cq;
    const bool nv = n < Cout;     // Cout % 4 == 0 on this path
    f32x4 v[4];
    bool rv[4];
    const float* src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int p = tile * 32 + rg + 8 * j;
        rv[j] = nv && p < HoWo;
        v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        src[j] = a.splitk_ws + (((size_t)blockIdx.z * nsplit * a.B + b) * (size_t)Mp + (rv[j] ? p : 0)) * Npad + (nv ? n : 0);
    }
    const size_t sstride = (size_t)a.B * Mp * Npad;
#pragma unroll 4
    for (int sp = 0; sp < nsplit; ++sp) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += *reinterpret_cast<const f32x4*>(src[j] + sp * sstride);
    }
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (nv && P.scale) sc = *reinterpret_cast<const f32x4*>(P.scale + n);
    if (nv && P.shift) sh = *reinterpret_cast<const f32x4*>(P.shift + n);
    f32x4 cs1 = {0.f, 0.f, 0.f, 0.f}, cs2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!rv[j]) continue;
        int p = tile * 32 + rg + 8 * j;
        f32x4 x = v[j];
        if (P.scale) x = x * sc;
        x = x + sh;
        size_t o = ((size_t)b * HoWo + p) * Cout + n;
        if (P.res) x += *reinterpret_cast<const f32x4*>(P.res + o);
        if (P.up) {
            int ho = p / a.Wo, wo = p - ho * a.Wo;
            x += *reinterpret_cast<const f32x4*>(P.up + (((size_t)b * Hu + (ho >> 1)) * Wu + (wo >> 1)) * Cout + n);
        }
        if (a.relu) { x[0] = fmaxf(x[0], 0.f); x[1] = fmaxf(x[1], 0.f); x[2] = fmaxf(x[2], 0.f); x[3] = fmaxf(x[3], 0.f); }
        *reinterpret_cast<f32x4*>(P.out + o) = x;
        cs1 += x;
        cs2 += x * x;
    }
    if (P.gn_part) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { s_sum[rg][4 * cq + e][0] = cs1[e]; s_sum[rg][4 * cq + e][1] = cs2[e]; }
        __syncthreads();
        if (t < 128 && nc + t < Cout) {
            float u = 0.f, w = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) { u += s_sum[g][t][0]; w += s_sum[g][t][1]; }
            float* gp = P.gn_part + (((size_t)b * P32 + tile) * Cout + nc + t) * 2;
            gp[0] = u; gp[1] = w;
        }
    }
}

// ------------------------------------------------------------------------------------------
// max pooling 3x3 stride 2 pad 1, NHWC, one float4 of channels per thread

// XCD-banded grid-stride walk of `total` items in row order (round 5).  Workgroups go to the 8 XCDs round-robin by linear id; a
// kernel whose items read NEIGHBOURING input rows (pooling windows, bilinear taps) then makes every XCD's L2 fetch every input
// row.  With a grid that is a multiple of 8, XCD x = blockIdx.x % 8 walks the x-th contiguous eighth of the items with its own
// blocks, so an input row is fetched by one L2 (two at a band border).  lo / hi / step in items; 32-bit (launchers check).
struct XcdWalk { unsigned first, end, step; };
__device__ __forceinline__ XcdWalk xcd_walk(unsigned total) {
    if ((gridDim.x & 7) != 0) return XcdWalk{blockIdx.x * blockDim.x + threadIdx.x, total, gridDim.x * blockDim.x};
    const unsigned per = (total + 7) >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const unsigned lo = xcd * per, hi = min(lo + per, total);
    return XcdWalk{lo + slot * blockDim.x + threadIdx.x, hi, (gridDim.x >> 3) * blockDim.x};
}

__global__ __launch_bounds__(256) void k_maxpool3x3s2(const float* __restrict__ in, float* __restrict__ out, int B,
                                                      int Hi, int Wi, int C, int Ho, int Wo) {
    const unsigned C4 = C >> 2;
    const XcdWalk wk = xcd_walk((unsigned)B * Ho * Wo * C4);      // < 2^32: launch_maxpool3x3s2
    for (unsigned g = wk.first; g < wk.end; g += wk.step) {
        int c4 = (int)(g % C4);
        unsigned r = g / C4;
        int wo = (int)(r % Wo); r /= Wo;
        int ho = (int)(r % Ho);
        int b = (int)(r / Ho);
        float4 m = make_float4(-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf());
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            int hi = ho * 2 - 1 + dy;
            if (hi < 0 || hi >= Hi) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                int wi = wo * 2 - 1 + dx;
                if (wi < 0 || wi >= Wi) continue;
                float4 v = *reinterpret_cast<const float4*>(in + (((size_t)b * Hi + hi) * Wi + wi) * C + 4 * c4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *reinterpret_cast<float4*>(out + (size_t)g * 4) = m;
    }
}

// ------------------------------------------------------------------------------------------
// GroupNorm statistics: partial sums -> per (image, channel) affine  y = x*a + b
// grid (groups, B, sites * kMaxGroup), 64 threads; fp64 combine in fixed order.  One launch serves every site whose
// statistics are complete (the four first segmentation blocks of a frame: one launch instead of four).

__global__ __launch_bounds__(64) void k_gn_finalize(const GnFinArgs a) {
    const int g = blockIdx.x, b = blockIdx.y, z = blockIdx.z;
    const int cpg = a.C / a.groups;
    const int site = z / kMaxGroup;
    const int P = a.P[site];
    const long long count = a.count[site];
    const float* part = a.gn_part[z] + (size_t)b * P * a.C * 2;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < P * cpg; i += 64) {
        int tile = i / cpg, c = g * cpg + (i - tile * cpg);
        const float* q = part + ((size_t)tile * a.C + c) * 2;
        s1 += (double)q[0]; s2 += (double)q[1];
    }
    s1 = wave_reduce_add(s1);
    s2 = wave_reduce_add(s2);
    s1 = __shfl(s1, 0, 64); s2 = __shfl(s2, 0, 64);
    double mean = s1 / (double)count;
    double var = s2 / (double)count - mean * mean;
    if (var < 0.0) var = 0.0;
    float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    float fmean = (float)mean;
    if ((int)threadIdx.x < cpg) {
        int c = g * cpg + threadIdx.x;
        float ga = a.gamma[z][c], be = a.beta[z][c];
        float sa = rstd * ga;
        float* o = a.affine[z] + ((size_t)b * a.C + c) * 2;
        o[0] = sa;
        o[1] = be - fmean * sa;
    }
}

// Lerp / lerp_coord / lerp_scaled: net_kernels.hpp (shared with up4.hip)

__device__ __forceinline__ float4 gn_relu4(const float* p, float4 sa, float4 sb) {
    float4 v = *reinterpret_cast<const float4*>(p);
    v.x = fmaxf(v.x * sa.x + sb.x, 0.f); v.y = fmaxf(v.y * sa.y + sb.y, 0.f);
    v.z = fmaxf(v.z * sa.z + sb.z, 0.f); v.w = fmaxf(v.w * sa.w + sb.w, 0.f);
    return v;
}

__device__ __forceinline__ float4 bilerp4(float4 v00, float4 v01, float4 v10, float4 v11, Lerp ly, Lerp lx) {
    float4 r;
    r.x = ly.l0 * (lx.l0 * v00.x + lx.l1 * v01.x) + ly.l1 * (lx.l0 * v10.x + lx.l1 * v11.x);
    r.y = ly.l0 * (lx.l0 * v00.y + lx.l1 * v01.y) + ly.l1 * (lx.l0 * v10.y + lx.l1 * v11.y);
    r.z = ly.l0 * (lx.l0 * v00.z + lx.l1 * v01.z) + ly.l1 * (lx.l0 * v10.z + lx.l1 * v11.z);
    r.w = ly.l0 * (lx.l0 * v00.w + lx.l1 * v01.w) + ly.l1 * (lx.l0 * v10.w + lx.l1 * v11.w);
    return r;
}

// affine [B][C][2] interleaved (a, b): two float4 loads give (a0,b0,a1,b1),(a2,b2,a3,b3)
__device__ __forceinline__ void load_affine4(const float* aff, float4& sa, float4& sb) {
    float4 u = *reinterpret_cast<const float4*>(aff), v = *reinterpret_cast<const float4*>(aff + 4);
    sa = make_float4(u.x, u.z, v.x, v.z);
    sb = make_float4(u.y, u.w, v.y, v.w);
}

// GN + ReLU + x2 bilinear upsample (Conv3x3GNReLU with upsample=True), grid-stride, grid.y = job * kMaxGroup + group.
// A thread = one 2 x 2 output block x one channel quad (round 5; one output pixel before): the block's taps lie in a 3 x 3
// neighbourhood of the input (the align_corners scale is < 1/2), so 9 loads and GroupNorm + ReLU evaluations serve four
// outputs instead of 16, and the index arithmetic is 32-bit (the per-pixel form divided 64-bit indices three times per
// element).  Every output is the per-pixel expression on the same four operands, selected from the neighbourhood (as in
// k_merge_head): bit-identical.
__global__ __launch_bounds__(256) void k_gn_relu_up2(const GnUpArgs a) {
    const int z = blockIdx.y;
    const float* in = a.in[z];
    const float* aff = a.affine[z];
    float* out = a.out[z];
    const int job = z / kMaxGroup;
    const int ah = a.h[job], aw = a.w[job];
    const int C4 = a.C >> 2, H2 = 2 * ah, W2 = 2 * aw;
    const float sy = H2 > 1 ? (float)(ah - 1) / (float)(H2 - 1) : 0.f, sx = W2 > 1 ? (float)(aw - 1) / (float)(W2 - 1) : 0.f;
    const XcdWalk wk = xcd_walk((unsigned)a.B * ah * aw * C4);      // < 2^32: launch_gn_relu_up2; an XCD walks a band of rows
    for (unsigned g = wk.first; g < wk.end; g += wk.step) {
        const unsigned c4 = g % C4;
        unsigned r = g / C4;
        const int bx = (int)(r % aw); r /= aw;
        const int by = (int)(r % ah);
        const int b = (int)(r / ah);
        const int Y = 2 * by, X = 2 * bx;
        const Lerp lya = lerp_scaled(Y, ah, sy), lyb = lerp_scaled(Y + 1, ah, sy);
        const Lerp lxa = lerp_scaled(X, aw, sx), lxb = lerp_scaled(X + 1, aw, sx);
        const int rb = lya.i0, cb = lxa.i0;
        const bool sy1 = lyb.i0 != rb, sx1 = lxb.i0 != cb;  // the second row / column's first tap is the next one
        const int rr[3] = {rb, min(rb + 1, ah - 1), min(rb + 2, ah - 1)}, cc[3] = {cb, min(cb + 1, aw - 1), min(cb + 2, aw - 1)};
        float4 sa4, sb4;
        load_affine4(aff + ((size_t)b * a.C + 4 * c4) * 2, sa4, sb4);
        const f32x4 sa = {sa4.x, sa4.y, sa4.z, sa4.w}, sb = {sb4.x, sb4.y, sb4.z, sb4.w};      // native vectors: float4 structs selected by ?: land in scratch
        const float* base = in + (size_t)b * ah * aw * a.C + 4 * c4;
        auto gnr = [&](int i, int j) {
            f32x4 v = *reinterpret_cast<const f32x4*>(base + ((size_t)rr[i] * aw + cc[j]) * a.C) * sa + sb;
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            return v;
        };
        const f32x4 t00 = gnr(0, 0), t01 = gnr(0, 1), t02 = gnr(0, 2), t10 = gnr(1, 0), t11 = gnr(1, 1), t12 = gnr(1, 2),
                    t20 = gnr(2, 0), t21 = gnr(2, 1), t22 = gnr(2, 2);
        float* o = out + (((size_t)b * H2 + Y) * W2 + X) * a.C + 4 * c4;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const Lerp ly = dy ? lyb : lya, lx = dx ? lxb : lxa;
                const bool sr = dy && sy1, sc = dx && sx1;
                // rows (r0, r0 + 1) and columns (c0, c0 + 1) of the neighbourhood
                const f32x4 a0 = sr ? t10 : t00, a1 = sr ? t11 : t01, a2 = sr ? t12 : t02;      // upper tap row
                const f32x4 b0 = sr ? t20 : t10, b1 = sr ? t21 : t11, b2 = sr ? t22 : t12;      // lower tap row
                const f32x4 v00 = sc ? a1 : a0, v01 = sc ? a2 : a1, v10 = sc ? b1 : b0, v11 = sc ? b2 : b1;
                // bilerp4's expression, component-wise
                const f32x4 res = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
                *reinterpret_cast<f32x4*>(o + ((size_t)dy * W2 + dx) * a.C) = res;
            }
    }
}

// ------------------------------------------------------------------------------------------
// FPN merge ("add" of the four segmentation blocks' outputs, smp MergeBlock) fused with the last
// GN+ReLU(+x2 upsample) of each branch and with the 1x1 segmentation head:
//   merged = up2(relu(gn(t5))) + up2(relu(gn(t4))) + up2(relu(gn(t3))) + relu(gn(t2))   (sum order of sum([...]))
//   logits_lowres[pixel][ch] = bias[ch] + sum_c merged[pixel][c] * W[ch][c]
// One workgroup = 32 pixels of one image (an 8 x 4 patch where the map divides into patches); merged tile and head weights live in LDS.
// Dropout2d is the identity in eval mode.  grid (ceil(H2*W2/64), B, G).
constexpr int kMhPx = 32;
constexpr int kMhMaxC = 128;
constexpr int kMhMaxCh = 32;

#ifdef FPC_STAMP_MH      // diagnostic build: summed phase ticks of lane 0 / wave 0 of every workgroup
__device__ unsigned long long g_mh[6];
#define FPC_MH_STAMP(I) do { if (threadIdx.x == 0) { long long n_ = clock64(); atomicAdd(&g_mh[I], (unsigned long long)(n_ - mh_t)); mh_t = n_; } } while (0)
#else
#define FPC_MH_STAMP(I) do { } while (0)
#endif

__global__ __launch_bounds__(256) void k_merge_head(const MergeHeadArgs a) {
#ifdef FPC_STAMP_MH
    long long mh_t = clock64();
#endif
    __shared__ __attribute__((aligned(16))) float s_m[kMhPx][kMhMaxC + 4];
    __shared__ __attribute__((aligned(16))) float s_w[kMhMaxCh][kMhMaxC + 4];
    const int z = blockIdx.z, b = blockIdx.y;
    const int H2 = 2 * a.h, W2 = 2 * a.w, C = a.C, C4 = C >> 2;
    const int ch = a.ch[z], chp = a.chp[z];
    // C4 divides 256 (C = 128): a thread's channel quad is fixed — no division inside the loops
    const int c4f = threadIdx.x % C4, prow = threadIdx.x / C4, pstep = 256 / C4;
    for (int r = prow; r < kMhMaxCh; r += pstep)            // rows past ch: zero (the MFMA tile is 32 wide)
        *reinterpret_cast<f32x4*>(&s_w[r][4 * c4f]) =
            r < ch ? *reinterpret_cast<const f32x4*>(a.hw[z] + (size_t)r * C + 4 * c4f) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int p0 = blockIdx.x * kMhPx;
    // the thread's GroupNorm affines are loaded once
    f32x4 la[3], lb[3], ha, hb;
    {
        float4 sa, sb;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            load_affine4(a.a_lo[z][k] + ((size_t)b * C + 4 * c4f) * 2, sa, sb);
            la[k] = f32x4{sa.x, sa.y, sa.z, sa.w}; lb[k] = f32x4{sb.x, sb.y, sb.z, sb.w};
        }
        load_affine4(a.a_hi[z] + ((size_t)b * C + 4 * c4f) * 2, sa, sb);
        ha = f32x4{sa.x, sa.y, sa.z, sa.w}; hb = f32x4{sb.x, sb.y, sb.z, sb.w};
    }
    FPC_MH_STAMP(0);      // head weights -> LDS, affines
    auto gnr = [](const float* ptr, f32x4 sa, f32x4 sb) {
        f32x4 v = *reinterpret_cast<const f32x4*>(ptr) * sa + sb;
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
        return v;
    };
    // bilinear source scales (align_corners): computed once, same arithmetic as lerp_coord
    const float sy = H2 > 1 ? (float)(a.h - 1) / (float)(H2 - 1) : 0.f, sx = W2 > 1 ? (float)(a.w - 1) / (float)(W2 - 1) : 0.f;
    // the workgroup's 32 pixels: an 8 x 4 patch where the map divides into patches (its taps are 5 x 3 low-resolution pixels per
    // branch instead of the 17 x 2 of a 32-pixel row segment: less than half the L2 -> L1 bytes), else 32 consecutive pixels
    const bool patch = (W2 % 8 == 0) && (H2 % 4 == 0);
    const int tpr = W2 >> 3;                                 // patches per patch row
    const int y0 = patch ? 4 * (blockIdx.x / tpr) : p0 / W2, x0 = patch ? 8 * (blockIdx.x % tpr) : p0 - y0 * W2;      // workgroup-uniform
    if (patch && C4 == 32) {
        // a thread = one 2 x 2 output block of the patch x one channel quad.  The block's taps lie in a 3 x 3 low-resolution
        // neighbourhood (the x2 align_corners scale is < 1/2: the second row / column starts at most one tap later), so a
        // branch costs 9 loads and GroupNorm + ReLU evaluations for FOUR pixels instead of 16, and the horizontal lerp of a
        // tap row serves both output rows.  Every pixel's value is the expression of the per-pixel loop below on the same
        // operands, selected from the neighbourhood: bit-identical.
        const int blk = threadIdx.x >> 5, c4 = c4f;
        const int Y = y0 + 2 * (blk >> 2), X = x0 + 2 * (blk & 3);
        const Lerp lya = lerp_scaled(Y, a.h, sy), lyb = lerp_scaled(Y + 1, a.h, sy);
        const Lerp lxa = lerp_scaled(X, a.w, sx), lxb = lerp_scaled(X + 1, a.w, sx);
        const int rb = lya.i0, cb = lxa.i0;
        const bool sy1 = lyb.i0 != rb, sx1 = lxb.i0 != cb;  // the second row / column's first tap is the next one
        const int rr[3] = {rb, min(rb + 1, a.h - 1), min(rb + 2, a.h - 1)}, cc[3] = {cb, min(cb + 1, a.w - 1), min(cb + 2, a.w - 1)};
        f32x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i >> 1][i & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* base = a.t_lo[z][k] + (size_t)b * a.h * a.w * C + 4 * c4;
            f32x4 g[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) g[r][c] = gnr(base + ((size_t)rr[r] * a.w + cc[c]) * C, la[k], lb[k]);
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const Lerp lxx = dx ? lxb : lxa;
                const bool s = dx && sx1;
                f32x4 hrow[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) hrow[r] = lxx.l0 * (s ? g[r][1] : g[r][0]) + lxx.l1 * (s ? g[r][2] : g[r][1]);
                acc[0][dx] += lya.l0 * hrow[0] + lya.l1 * hrow[1];
                acc[1][dx] += lyb.l0 * (sy1 ? hrow[1] : hrow[0]) + lyb.l1 * (sy1 ? hrow[2] : hrow[1]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int dy = i >> 1, dx = i & 1;
            acc[dy][dx] += gnr(a.t_hi[z] + ((size_t)b * H2 * W2 + (size_t)(Y + dy) * W2 + X + dx) * C + 4 * c4, ha, hb);
            *reinterpret_cast<f32x4*>(&s_m[(2 * (blk >> 2) + dy) * 8 + 2 * (blk & 3) + dx][4 * c4]) = acc[dy][dx];
        }
    } else
#pragma unroll 2
    for (int pl = prow; pl < kMhPx; pl += pstep) {
        const int c4 = c4f;
        int p = patch ? (y0 + (pl >> 3)) * W2 + x0 + (pl & 7) : p0 + pl;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (p < H2 * W2) {
            int y = y0, x = x0 + pl;
            if (patch) { y = y0 + (pl >> 3); x = x0 + (pl & 7); }
            while (x >= W2) { x -= W2; ++y; }
            Lerp ly = lerp_scaled(y, a.h, sy), lx = lerp_scaled(x, a.w, sx);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float* base = a.t_lo[z][k] + (size_t)b * a.h * a.w * C + 4 * c4;
                f32x4 v00 = gnr(base + ((size_t)ly.i0 * a.w + lx.i0) * C, la[k], lb[k]);
                f32x4 v01 = gnr(base + ((size_t)ly.i0 * a.w + lx.i1) * C, la[k], lb[k]);
                f32x4 v10 = gnr(base + ((size_t)ly.i1 * a.w + lx.i0) * C, la[k], lb[k]);
                f32x4 v11 = gnr(base + ((size_t)ly.i1 * a.w + lx.i1) * C, la[k], lb[k]);
                acc += ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
            }
            acc += gnr(a.t_hi[z] + ((size_t)b * H2 * W2 + p) * C + 4 * c4, ha, hb);
        }
        *reinterpret_cast<f32x4*>(&s_m[pl][4 * c4]) = acc;
    }
    FPC_MH_STAMP(1);      // gather + GroupNorm + ReLU + bilinear merge
    __syncthreads();
    FPC_MH_STAMP(2);      // barrier
    // head (1x1 conv, C -> ch <= 32) on the matrix cores: the 32 px x 32 ch tile, K = C split over the four waves
    // (wave w owns channels [w C/4, (w+1) C/4)); A = merged activations, B = head weights, both read from LDS as
    // 16-byte fragments (2 x C/32 ds_read_b128 per wave — the scalar-FMA form needed ~900 wave-level LDS reads
    // per workgroup and was LDS-bound: 40 us for the four decoders).  Partials are summed in wave order.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int kq = C >> 2;                                   // K range of a wave (C % 32 == 0, checked by the launcher)
    for (int k0 = wv * kq; k0 < (wv + 1) * kq; k0 += 8) {
        f32x4 fa = *reinterpret_cast<const f32x4*>(&s_m[li][k0 + 4 * lh]);
        f32x4 fb = *reinterpret_cast<const f32x4*>(&s_w[li][k0 + 4 * lh]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc, 0, 0, 0);
    }
    FPC_MH_STAMP(3);      // MFMA head
    __syncthreads();                                         // all fragments read: s_m becomes the partial buffer
    float* part = &s_m[0][0];                                // [4 waves][32 px][33]: 4224 floats = sizeof(s_m)
    static_assert(kMhPx * (kMhMaxC + 4) >= 4 * 32 * 33, "partial buffer");
#pragma unroll
    for (int r = 0; r < 16; ++r) part[(wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = acc[r];
    __syncthreads();
    // the 32 pixels' outputs are one contiguous block of 32 * chp floats; 32 lanes per pixel (k < chp active)
    const int nvalid = patch ? kMhPx : min(kMhPx, H2 * W2 - p0);
    float* ob = a.out[z] + (size_t)b * H2 * W2 * chp;
    const int ok = threadIdx.x & 31;
    const float bias = ok < ch ? a.hb[z][ok] : 0.f;
    for (int pl = threadIdx.x >> 5; pl < nvalid; pl += 8) {
        if (ok >= chp) continue;
        float v = 0.f;
        if (ok < ch) {
            v = bias;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += part[(w * 32 + pl) * 33 + ok];
        }
        const int p = patch ? (y0 + (pl >> 3)) * W2 + x0 + (pl & 7) : p0 + pl;
        ob[(size_t)p * chp + ok] = v;
    }
    FPC_MH_STAMP(4);      // partial exchange + stores
#ifdef FPC_STAMP_MH
    if (threadIdx.x == 0) atomicAdd(&g_mh[5], 1ull);
#endif
}

#ifdef FPC_STAMP_MH
extern "C" int fpc_dbg_merge_head_stamps(unsigned long long* out6) {
    unsigned long long z[6] = {0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out6, HIP_SYMBOL(g_mh), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_mh), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

// ------------------------------------------------------------------------------------------
// x4 bilinear upsample (UpsamplingBilinear2d, align_corners=True) of the four heads' low-res
// logits to full resolution, the xyz -> xy / z channel split (pose_regressor.py:729-732) and
// class compression (pose_regressor.py:445-457, gpu_tensor_funcs.py:37-99) in one pass.
// One thread per output pixel, lanes along x: every full-res plane store is a coalesced
// 256-byte wave row; the low-res taps (5 MB in total) are served by L1/L2.

template <int MAXC>
__global__ __launch_bounds__(256) void k_up4_compress(const Up4Args a) {
    const int b = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= a.W) return;
    const int HW = a.H * a.W, C = a.C, G = C - 1;
    const size_t p = (size_t)y * a.W + x;
    Lerp ly = lerp_coord(y, a.hl, a.H), lx = lerp_coord(x, a.wl, a.W);
    const size_t t00 = ((size_t)b * a.hl + ly.i0) * a.wl + lx.i0, t01 = ((size_t)b * a.hl + ly.i0) * a.wl + lx.i1;
    const size_t t10 = ((size_t)b * a.hl + ly.i1) * a.wl + lx.i0, t11 = ((size_t)b * a.hl + ly.i1) * a.wl + lx.i1;
    auto tap = [&](const float* L, int stride, int c) {
        return ly.l0 * (lx.l0 * L[t00 * stride + c] + lx.l1 * L[t01 * stride + c]) +
               ly.l1 * (lx.l0 * L[t10 * stride + c] + lx.l1 * L[t11 * stride + c]);
    };
    // mask logits + arg-max of the log-softmax (first maximal index on ties), as class_compress.hip
    float v[MAXC];
    float mx = -__builtin_huge_valf();
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < C) {
            v[c] = tap(a.lm, a.pm, c);
            mx = fmaxf(mx, v[c]);
            if (a.o_mask) a.o_mask[((size_t)b * C + c) * HW + p] = v[c];
        }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < C) s += expf(v[c] - mx);
    float lse = logf(s);
    float best = (v[0] - mx) - lse;
    int cls = 0;
#pragma unroll
    for (int c = 1; c < MAXC; ++c)
        if (c < C) {
            float val = (v[c] - mx) - lse;
            if (val > best) { best = val; cls = c; }
        }
    a.cat_mask[(size_t)b * HW + p] = cls;
    if (a.fg_bits) {            // rows start on word boundaries (the host checked W % 64 == 0): a wave = one word
        const unsigned long long fgm = __ballot(cls != 0);
        if ((threadIdx.x & 63) == 0) a.fg_bits[(size_t)b * a.fg_stride + (p >> 6)] = fgm;
    }
    const int g = cls - 1;
    float q[4] = {0, 0, 0, 0}, sc[3] = {0, 0, 0}, vxy[2] = {0, 0}, zz = 0.f;
    for (int k = 0; k < G; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float u = tap(a.lq, a.pq, 4 * k + e);
            if (a.o_quat) a.o_quat[((size_t)b * 4 * G + 4 * k + e) * HW + p] = u;
            if (k == g) q[e] = u;
        }
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            float u = tap(a.ls, a.ps, 3 * k + e);
            if (a.o_scales) a.o_scales[((size_t)b * 3 * G + 3 * k + e) * HW + p] = u;
            if (k == g) sc[e] = u;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float u = tap(a.lt, a.pt, 3 * k + e);
            if (a.o_xy) a.o_xy[((size_t)b * 2 * G + 2 * k + e) * HW + p] = u;
            if (k == g) vxy[e] = u;
        }
        float u = tap(a.lt, a.pt, 3 * k + 2);
        if (a.o_z) a.o_z[((size_t)b * G + k) * HW + p] = u;
        if (k == g) zz = u;
    }
    float nq = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (nq == 0.0f) nq = 1.0f;
    float nv = sqrtf(vxy[0] * vxy[0] + vxy[1] * vxy[1]);
    if (nv == 0.0f) nv = 1.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) a.cq[((size_t)b * 4 + e) * HW + p] = q[e] / nq;
#pragma unroll
    for (int e = 0; e < 3; ++e) a.cs[((size_t)b * 3 + e) * HW + p] = sc[e];
#pragma unroll
    for (int e = 0; e < 2; ++e) a.cxy[((size_t)b * 2 + e) * HW + p] = vxy[e] / nv;
    a.cz[(size_t)b * HW + p] = zz;
}

// The same for the reference's 7 classes (bg + 6): every channel index is a compile-time constant, so the
// four low-res taps are fetched as 18 float4 (channel strides 8 / 24 / 20 / 20, 16-byte aligned pixels)
// instead of 67 scalars each, and all 67 interpolated values stay in registers.
__global__ __launch_bounds__(256) void k_up4_compress7(const Up4Args a) {
    const int b = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= a.W) return;
    constexpr int C = 7, G = 6;
    const int HW = a.H * a.W;
    const size_t p = (size_t)y * a.W + x;
    Lerp ly = lerp_coord(y, a.hl, a.H), lx = lerp_coord(x, a.wl, a.W);
    const size_t t00 = ((size_t)b * a.hl + ly.i0) * a.wl + lx.i0, t01 = ((size_t)b * a.hl + ly.i0) * a.wl + lx.i1;
    const size_t t10 = ((size_t)b * a.hl + ly.i1) * a.wl + lx.i0, t11 = ((size_t)b * a.hl + ly.i1) * a.wl + lx.i1;
    auto quad = [&](const float* L, int stride, int q) {
        f32x4 v00 = *reinterpret_cast<const f32x4*>(L + t00 * stride + 4 * q);
        f32x4 v01 = *reinterpret_cast<const f32x4*>(L + t01 * stride + 4 * q);
        f32x4 v10 = *reinterpret_cast<const f32x4*>(L + t10 * stride + 4 * q);
        f32x4 v11 = *reinterpret_cast<const f32x4*>(L + t11 * stride + 4 * q);
        return ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
    };
    float vm[8], vq[24], vt[20], vs[20];
#pragma unroll
    for (int q = 0; q < 2; ++q) { f32x4 r = quad(a.lm, 8, q); vm[4 * q] = r[0]; vm[4 * q + 1] = r[1]; vm[4 * q + 2] = r[2]; vm[4 * q + 3] = r[3]; }
#pragma unroll
    for (int q = 0; q < 6; ++q) { f32x4 r = quad(a.lq, 24, q); vq[4 * q] = r[0]; vq[4 * q + 1] = r[1]; vq[4 * q + 2] = r[2]; vq[4 * q + 3] = r[3]; }
#pragma unroll
    for (int q = 0; q < 5; ++q) { f32x4 r = quad(a.lt, 20, q); vt[4 * q] = r[0]; vt[4 * q + 1] = r[1]; vt[4 * q + 2] = r[2]; vt[4 * q + 3] = r[3]; }
#pragma unroll
    for (int q = 0; q < 5; ++q) { f32x4 r = quad(a.ls, 20, q); vs[4 * q] = r[0]; vs[4 * q + 1] = r[1]; vs[4 * q + 2] = r[2]; vs[4 * q + 3] = r[3]; }
    // arg-max of the log-softmax, first maximal index on ties (as class_compress.hip)
    float mx = vm[0];
#pragma unroll
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, vm[c]);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) sum += expf(vm[c] - mx);
    float lse = logf(sum);
    float best = (vm[0] - mx) - lse;
    int cls = 0;
#pragma unroll
    for (int c = 1; c < C; ++c) {
        float val = (vm[c] - mx) - lse;
        if (val > best) { best = val; cls = c; }
    }
    a.cat_mask[(size_t)b * HW + p] = cls;
    if (a.fg_bits) {            // rows start on word boundaries (the host checked W % 64 == 0): a wave = one word
        const unsigned long long fgm = __ballot(cls != 0);
        if ((threadIdx.x & 63) == 0) a.fg_bits[(size_t)b * a.fg_stride + (p >> 6)] = fgm;
    }
    if (a.o_mask) {
        // the 67 full-resolution logit planes (82 MB per frame) are the caller's output: nothing on this path reads them back,
        // so they go out as streaming stores (the categorical planes below are read by the aggregation next: ordinary stores)
#pragma unroll
        for (int c = 0; c < C; ++c) __builtin_nontemporal_store(vm[c], a.o_mask + ((size_t)b * C + c) * HW + p);
#pragma unroll
        for (int c = 0; c < 4 * G; ++c) __builtin_nontemporal_store(vq[c], a.o_quat + ((size_t)b * 4 * G + c) * HW + p);
#pragma unroll
        for (int c = 0; c < 3 * G; ++c) __builtin_nontemporal_store(vs[c], a.o_scales + ((size_t)b * 3 * G + c) * HW + p);
#pragma unroll
        for (int k = 0; k < G; ++k) {
            __builtin_nontemporal_store(vt[3 * k], a.o_xy + ((size_t)b * 2 * G + 2 * k) * HW + p);
            __builtin_nontemporal_store(vt[3 * k + 1], a.o_xy + ((size_t)b * 2 * G + 2 * k + 1) * HW + p);
            __builtin_nontemporal_store(vt[3 * k + 2], a.o_z + ((size_t)b * G + k) * HW + p);
        }
    }
    float q[4] = {0, 0, 0, 0}, sc[3] = {0, 0, 0}, vxy[2] = {0, 0}, zz = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k)
        if (k == cls - 1) {
            q[0] = vq[4 * k]; q[1] = vq[4 * k + 1]; q[2] = vq[4 * k + 2]; q[3] = vq[4 * k + 3];
            sc[0] = vs[3 * k]; sc[1] = vs[3 * k + 1]; sc[2] = vs[3 * k + 2];
            vxy[0] = vt[3 * k]; vxy[1] = vt[3 * k + 1]; zz = vt[3 * k + 2];
        }
    float nq = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (nq == 0.0f) nq = 1.0f;
    float nv = sqrtf(vxy[0] * vxy[0] + vxy[1] * vxy[1]);
    if (nv == 0.0f) nv = 1.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) a.cq[((size_t)b * 4 + e) * HW + p] = q[e] / nq;
#pragma unroll
    for (int e = 0; e < 3; ++e) a.cs[((size_t)b * 3 + e) * HW + p] = sc[e];
#pragma unroll
    for (int e = 0; e < 2; ++e) a.cxy[((size_t)b * 2 + e) * HW + p] = vxy[e] / nv;
    a.cz[(size_t)b * HW + p] = zz;
}

// ------------------------------------------------------------------------------------------
// parameter repacking (once per plan)

// OIHW [Cout][Cin][Kh][Kw] -> OHWI rows [Npad][Kpad], k = (kh*Kwp + kw)*Cinp + ci, zero padded.  Cinp >= Cin: channel
// padding of the input layout (4 for the RGB stem); Kwp >= Kw: taps per kernel row in the layout (8 for the stem,
// whose K-step is one kernel row = 8 consecutive 4-channel pixels)
__global__ __launch_bounds__(256) void k_pack_weight(const float* __restrict__ w, float* __restrict__ out, int Cout,
                                                     int Cin, int Cinp, int Kh, int Kw, int Kwp, int Npad, int Kpad) {
    long long total = (long long)Npad * Kpad;
    int K = Cinp * Kh * Kwp;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        int k = (int)(g % Kpad), n = (int)(g / Kpad);
        float v = 0.f;
        if (n < Cout && k < K) {
            int tap = k / Cinp, ci = k - tap * Cinp;
            int kh = tap / Kwp, kw = tap - kh * Kwp;
            if (ci < Cin && kw < Kw) v = w[(((size_t)n * Cin + ci) * Kh + kh) * Kw + kw];
        }
        out[g] = v;
    }
}

// The same [Npad][Kpad] image split EXACTLY into three bf16 planes (x = b1 + b2 + b3 by truncation, as split_bf3):
// out[(plane * Npad + n) * Kpad + k] — the weight operand of the split-precision direct convolution, staged by LDS-DMA.
__global__ __launch_bounds__(256) void k_pack_weight_bf3(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout,
                                                       int Cin, int Cinp, int Kh, int Kw, int Kwp, int Npad, int Kpad) {
    const int K = Cinp * Kh * Kwp;
    long long total = (long long)Npad * Kpad;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        int n = (int)(g / Kpad), k = (int)(g - (long long)n * Kpad);
        float v = 0.f;
        if (n < Cout && k < K) {
            int tap = k / Cinp, ci = k - tap * Cinp;
            int kh = tap / Kwp, kw = tap - kh * Kwp;
            if (ci < Cin && kw < Kw) v = w[(((size_t)n * Cin + ci) * Kh + kh) * Kw + kw];
        }
        const unsigned b1 = __builtin_bit_cast(unsigned, v) & 0xFFFF0000u;
        const float r = v - __builtin_bit_cast(float, b1);
        const unsigned b2 = __builtin_bit_cast(unsigned, r) & 0xFFFF0000u;
        const float q = r - __builtin_bit_cast(float, b2);
        out[g] = (unsigned short)(b1 >> 16);
        out[total + g] = (unsigned short)(b2 >> 16);
        out[2 * total + g] = (unsigned short)(__builtin_bit_cast(unsigned, q) >> 16);
    }
}

// image NCHW [B,3,H,W] -> NHWC4 [B,H,W,4] (4th channel 0): 16-byte pixels for the stem's loader
__global__ __launch_bounds__(256) void k_nchw3_to_nhwc4(const float* __restrict__ x, float* __restrict__ out, int B,
                                                        int HW) {
    long long total = (long long)B * HW;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        long long b = g / HW, p = g - b * HW;
        const float* s = x + b * 3 * HW + p;
        *reinterpret_cast<f32x4*>(out + g * 4) = f32x4{s[0], s[HW], s[2 * (long long)HW], 0.f};
    }
}

// eval-mode BatchNorm as y = x*scale + shift (torch: (x - mean) / sqrt(var + eps) * gamma + beta)
__global__ void k_fold_bn(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                          const float* __restrict__ var, float eps, int C, float* __restrict__ scale,
                          float* __restrict__ shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float inv = 1.0f / sqrtf(var[c] + eps);
    float s = gamma[c] * inv;
    scale[c] = s;
    shift[c] = beta[c] - mean[c] * s;
}

// ------------------------------------------------------------------------------------------
// launch wrappers

template <int BM, int BN>
static void launch_conv_t(const ConvArgs& a, int groups, hipStream_t s) {
    dim3 grid(a.mtiles * a.B * a.ntiles * a.nsplit * groups);
    if (a.generic == 0 && a.bf3)
        hipLaunchKernelGGL((k_conv_igemm<BM, BN, 0, true>), grid, dim3(256), 0, s, a);
    else if (a.generic == 0)
        hipLaunchKernelGGL((k_conv_igemm<BM, BN, 0>), grid, dim3(256), 0, s, a);
    else if (a.generic == 2)
        hipLaunchKernelGGL((k_conv_igemm<BM, BN, 2>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((k_conv_igemm<BM, BN, 1>), grid, dim3(256), 0, s, a);
}

// a.generic: 0 = Cin % 32 == 0 channel-last (fast path), 2 = Cin % 4 == 0 channel-last (per-lane tap decode),
//            1 = anything (scalar gathers)
int launch_conv(const ConvArgs& a, int groups, hipStream_t s) {
    if (groups < 1 || groups > kMaxGroup || a.Npad % a.bn != 0 || a.Kpad % kConvBK != 0) return FPC_EINVAL;
    if (a.generic == 0 && (a.Cin % kConvBK != 0 || a.in_sc != 1 || a.Kh * a.Kw > 32 ||
                           ((long long)a.Hi + 2 * a.pad) * a.in_sh * 4 >= (1LL << 31)))
        return FPC_EINVAL;      // tap mask is 32 bits, lane offsets are 31 bits
    if ((long long)a.Npad * a.Kpad * 4 >= (1LL << 31)) return FPC_EINVAL;
    if (a.nsplit > 1 && a.fused && (!a.tickets || (long long)groups * a.B * a.mtiles * a.ntiles > kConvTickets || a.Cout % 4 != 0))
        return FPC_EINVAL;
    if (a.generic == 2 && (a.Cin % 4 != 0 || a.in_sc != 1 || a.in_sw % 4 != 0 || a.in_sh % 4 != 0 || a.in_sb % 4 != 0))
        return FPC_EINVAL;
    if (a.bm == 128 && a.bn == 128) launch_conv_t<128, 128>(a, groups, s);
    else if (a.bm == 128 && a.bn == 64) launch_conv_t<128, 64>(a, groups, s);
    else if (a.bm == 64 && a.bn == 128) launch_conv_t<64, 128>(a, groups, s);
    else if (a.bm == 64 && a.bn == 64) launch_conv_t<64, 64>(a, groups, s);
    else return FPC_EINVAL;
    return check_launch();
}

int launch_conv_splitk_epilogue(const ConvArgs& a, int groups, hipStream_t s) {
    if (a.Cout % 4 != 0) return FPC_EINVAL;
    hipLaunchKernelGGL(k_conv_splitk_epilogue, dim3((a.mtiles * a.bm / 32) * cdiv(a.Cout, 128), a.B, groups), dim3(256), 0, s,
                       a);
    return check_launch();
}

static int stream_grid(long long work_items) {
    long long g = (work_items + 255) / 256;
    g = g < 1 ? 1 : (g > 4096 ? 4096 : g);
    return (int)(g >= 8 ? (g + 7) / 8 * 8 : g);      // a multiple of 8: the XCD-banded walks (xcd_walk) need it, the others do not mind
}

int launch_maxpool3x3s2(const float* in, float* out, int B, int Hi, int Wi, int C, int Ho, int Wo, hipStream_t s) {
    // (xcd_walk's 32-bit `first + k * step` must not wrap: one grid stride — at most 4096 x 256 items — of headroom below 2^32)
    if (C % 4 != 0 || (long long)B * Ho * Wo * (C / 4) >= (1LL << 32) - 4096LL * 256 - 8) return FPC_EINVAL;
    hipLaunchKernelGGL(k_maxpool3x3s2, dim3(stream_grid((long long)B * Ho * Wo * (C / 4))), dim3(256), 0, s, in, out, B,
                       Hi, Wi, C, Ho, Wo);
    return check_launch();
}

int launch_gn_finalize(const GnFinArgs& a, int groups, hipStream_t s) {
    if (a.C % a.groups != 0 || a.C / a.groups > 64) return FPC_EINVAL;
    if (a.sites < 1 || a.sites > kMaxGnSites || groups != kMaxGroup) return FPC_EINVAL;
    hipLaunchKernelGGL(k_gn_finalize, dim3(a.groups, a.B, a.sites * kMaxGroup), dim3(64), 0, s, a);
    return check_launch();
}

int launch_gn_relu_up2(const GnUpArgs& a, int groups, hipStream_t s) {
    if (a.C % 4 != 0) return FPC_EINVAL;
    if (a.jobs < 1 || a.jobs > kMaxUpJobs || groups != kMaxGroup) return FPC_EINVAL;
    long long most = 0;
    for (int j = 0; j < a.jobs; ++j) most = std::max(most, (long long)a.B * a.h[j] * a.w[j] * (a.C / 4));      // 2 x 2 blocks x channel quads
    if (most >= (1LL << 31)) return FPC_EINVAL;
    hipLaunchKernelGGL(k_gn_relu_up2, dim3(stream_grid(most), a.jobs * kMaxGroup), dim3(256), 0, s, a);
    return check_launch();
}

int launch_merge_head(const MergeHeadArgs& a, int groups, hipStream_t s) {
    if (a.C > kMhMaxC || a.C % 32 != 0 || 256 % (a.C / 4) != 0) return FPC_EINVAL;      // the head splits C over 4 waves in 8-channel steps
    for (int z = 0; z < groups; ++z)
        if (a.ch[z] > kMhMaxCh || a.chp[z] > kMhMaxCh || a.chp[z] < a.ch[z]) return FPC_EINVAL;
    hipLaunchKernelGGL(k_merge_head, dim3(cdiv(4 * a.h * a.w, kMhPx), a.B, groups), dim3(256), 0, s, a);
    return check_launch();
}

int launch_up4_compress(const Up4Args& a, hipStream_t s) {
    if (a.C < 2 || a.C > 32 || a.H > 65535 || a.B > 65535) return FPC_EINVAL;
    if (a.fg_bits && a.W % 64 != 0) return FPC_EINVAL;
    dim3 grid(cdiv(a.W, 256), a.H, a.B);
    const bool seven = a.C == 7 && a.pm == 8 && a.pq == 24 && a.pt == 20 && a.ps == 20;
    const bool all_or_none = (a.o_mask != nullptr) == (a.o_quat != nullptr) && (a.o_mask != nullptr) == (a.o_scales != nullptr) &&
                             (a.o_mask != nullptr) == (a.o_xy != nullptr) && (a.o_mask != nullptr) == (a.o_z != nullptr);
    if (seven && a.W % 4 == 0 && ((long long)a.H * a.W) % 256 == 0 && (long long)a.H * a.W < (1LL << 31) && all_or_none)
        launch_up4_compress7x4(a, s);
    else if (seven)
        hipLaunchKernelGGL(k_up4_compress7, dim3(cdiv(a.W, 128), a.H, a.B), dim3(128), 0, s, a);
    else if (a.C <= 8) hipLaunchKernelGGL(k_up4_compress<8>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_up4_compress<32>, grid, dim3(256), 0, s, a);
    return check_launch();
}


// ------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) convolution (3x3, stride 1, pad 1) on the f32 matrix cores: 2.25x fewer
// multiply-adds than the direct form, all in f32.
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A  per 4x4 input patch d / 2x2 output tile, summed over Cin:
//   16 independent GEMMs  M_xi[tile][co] = sum_ci V_xi[tile][ci] * U_xi[ci][co].
// One workgroup = an 8x4 patch of tiles (16x8 output pixels) x 64 output channels x all 16 xi.
// Wave w owns transform row i = w (xi = 4i..4i+3): 4 xi x 2 column tiles of 32 = 8 MFMA accumulators.
// Per K-step of 8 input channels the RAW 18x10 input region and the pre-transformed weights
// (fpc::k_wino_pack, 32 KB contiguous per step) are staged global -> registers -> LDS; each lane builds
// its four V_xi fragments from 8 LDS reads + 8 vector adds, so the transformed input never exists in
// memory.  The output transform runs through LDS (wave i holds row i of M) and feeds the same
// epilogue as the direct kernel (BatchNorm / bias, residual, ReLU, GroupNorm partial sums).
#ifndef FPC_WINO_PRIO_HI
#define FPC_WINO_PRIO_HI 1
#endif
#ifndef FPC_WINO_STAGGER
#define FPC_WINO_STAGGER 1
#endif
#ifndef FPC_WINO_B3_OLDER
#define FPC_WINO_B3_OLDER 1
#endif
#ifndef FPC_WINO_IN_OLDER
#define FPC_WINO_IN_OLDER 0
#endif
#ifndef FPC_WINO_LATE_AT
#define FPC_WINO_LATE_AT 0
#endif
constexpr int kWinoTX = 8;                              // tile patch per workgroup: 8 wide, NW tall (NW = 4 or 8 waves)
constexpr int kWinoRW = 2 * kWinoTX + 2;                // input region width 18
constexpr int kWinoIS = 8;                              // floats per staged position (one 32-byte K-step slice)
constexpr int kWinoBN = 64;
constexpr int kWinoLdsW = 16 * kWinoBN * 8;             // 8192 floats (32 KB) per weight buffer

// NW = 4: 32 tiles (16x8 output pixels) per workgroup, 2 workgroups per CU.
// NW = 8: 64 tiles (16x16 pixels), waves 4-7 work on the lower half of the patch with the SAME weights in
//         LDS: 11.8 instead of 6.7 multiply-adds per staged byte.  The kernels sit on the ~10 B/clk/CU the
//         global -> LDS path delivers (measured: MFMA busy 52 % at 6.7 MAC/B), so this is the lever for
//         the large maps; the 4-wave form keeps more workgroups for the small ones.
// WP ("wave private", NW = 4): no barrier inside the K loop.  A wave needs only ITS four xi rows of the
// weight image (exactly the 8 KB it DMAs itself) and 5 or 8 rows of the input region, which it stages
// into a private LDS patch; every fragment of a K-step is pulled into registers first, so the single
// LDS buffer can be refilled (DMA + ds_write) under that step's 32 MFMAs.  Ablation of the barrier form:
// the two barriers per step cost 17 % of the kernel, they also force the four waves into lockstep.
// P3 (NW = 8): every operand goes global -> LDS by DMA (out-of-image positions read a zero page), three
// LDS stages, the loads of step k+2 are issued before step k's MFMAs and the wave waits with a COUNTED
// vmcnt (the newest batch stays in flight across the raw s_barrier) — guide "Pipelining across barriers".
// DBG: diagnostic instantiations that stamp the K-loop phases with s_memtime (tools_dev/wino_stamps.py).
// BF3 (NW = 8, barrier form): split-precision products.  The transformed input tile is split EXACTLY into three bf16
// pieces per value between the MFMAs, the weights arrive pre-split (k_wino_pack_bf3: a 32 KB {b1, b2} image in the f32
// image's own layout + a 16 KB {b3} image per K-step), and the 8-channel K-step of a 32 x 32 tile is THREE
// v_mfma_f32_32x32x16_bf16 (slots: a1 b1, a1 b2 | a2 b1, a2 b2 | a1 b3, a3 b1 for the lane half's four channels;
// dropped products < 2^-23 of the term, f32 accumulation) instead of four v_mfma_f32_32x32x2_f32: 96 instead of 256
// matrix cycles.  118 KB of LDS: one 8-wave workgroup per CU, as the f32 8-wave form.
template <int NW, bool WP, bool P3, bool DBG = false, bool BF3 = false>
__global__ __launch_bounds__(64 * NW, BF3 ? 1 : 2) void k_conv_wino(const WinoArgs a) {
    constexpr int TY = NW;                                  // tile rows of the patch
    constexpr int RH = 2 * TY + 2, POS = kWinoRW * RH;      // staged input region
    constexpr int NT = 8 * NW;                              // tiles per workgroup
    constexpr int WPI = 8 * kWinoRW * kWinoIS;              // WP: floats of a wave's private input patch (8 rows)
    constexpr int IP3 = 512 * kWinoIS;                      // P3: floats per input stage (16 pieces of 1 KB, 324 positions used)
    // barrier form: 1 KB input pieces per stage, floats per input buffer.  PERM (the BF3 form): the 16-byte units of the region
    // are PERMUTED in LDS so that the fragment reads are conflict-free — position (ry, rx), channel half hf lives in unit
    //   (((a >> 2) * 3 + (q >> 2)) * 8 + (ry & 1) * 4 + (rx & 1) * 2 + hf) * 16 + 4 * (q & 3) + (a & 3),   a = ry >> 1, q = rx >> 1:
    // the 16 lanes of a ds_read_b128 group are 4 tile rows x 4 tile columns at fixed row / column parity, i.e. 16 different
    // (a & 3, q & 3) = 16 different 16-byte bank slots (the row-major image put them on 4: every read 4-way, 1024 of the
    // ~2300 LDS cycles of a K-step).  The DMA lands lane-linear pieces, so the permutation is only the choice of the global
    // address each lane fetches; 72 x 16 units = 18 pieces with 648 of 1152 lanes active.
    constexpr bool PERM = BF3;
    constexpr int NPI = PERM ? 18 : (POS + 31) / 32, LINP = NPI * 256;
    // input pieces per wave and K-step: piece (wave + IST i), i < NIN.  NW = 8: 18 (PERM) / 11 pieces over the 8 waves, or —
    // FPC_WINO_IN_OLDER — over the four older waves only (see the {b3} pieces)
    constexpr bool INO = PERM && FPC_WINO_IN_OLDER;
    constexpr int NIN = INO ? 5 : (PERM ? 3 : 2), IST = INO ? 4 : NW;
    static_assert(!PERM || NW == 8, "permuted input image is laid out for the 18 x 18 region");
    constexpr int kWB = BF3 ? 12288 : kWinoLdsW;            // floats per weight buffer (BF3: 32 KB {b1, b2} + 16 KB {b3})
    constexpr int kLdsFloats = WP ? (kWinoLdsW + 4 * WPI) : P3 ? (3 * IP3 + 3 * kWinoLdsW) : (2 * LINP + 2 * kWB);
    static_assert(!BF3 || (NW == 8 && !WP && !P3), "split precision rides on the 8-wave barrier form");
    static_assert(!WP || NW == 4, "wave-private form is written for 4 waves");
    static_assert(!P3 || (NW == 8 && !WP), "three-stage DMA form is written for 8 waves");
    static_assert(kLdsFloats >= 2 * 4 * NT * 32, "output transform needs 2*4*NT*32 floats");
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
    const long long t_entry = DBG ? clock64() : 0;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int wi = wv & 3, half = wv >> 2;                  // transform row of this wave, tile-row group
    const int li = lane & 31, lh = lane >> 5;
    const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, HW = H * W;
    const int nkb = Cin >> 3;
    // weight slice (group, 64-channel block) fastest: fixed per XCD under round-robin dispatch, so each
    // XCD's L2 keeps the 16*64*Cin*4 bytes of transformed weights all its workgroups stream (see k_conv_igemm)
    int bid = blockIdx.x;
    const int nnb = Cout / kWinoBN;
    const int nb = bid % nnb; bid /= nnb;
    const int grp = bid % a.groups; bid /= a.groups;
    const int bx = bid % a.tbx; bid /= a.tbx;
    const int by = bid % a.tby;
    const int b = bid / a.tby;
    ConvPtrs P = a.p[0];
    if (grp == 1) P = a.p[1];
    if (grp == 2) P = a.p[2];
    if (grp == 3) P = a.p[3];
    const int ty0 = by * TY, tx0 = bx * kWinoTX;
    const int y_in0 = 2 * ty0 - 1, x_in0 = 2 * tx0 - 1;

    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][nt][r] = 0.f;
    // row pair (ra, rb) and sign of B^T row wi:  0: d0-d2   1: d1+d2   2: d2-d1   3: d1-d3
    const float sgn = (wi == 1) ? 1.f : -1.f;

    if constexpr (WP) {
        // ---- wave-private staging
        float* const Wp = lds + wi * 2048;                        // this wave's xi rows [4][64][8]
        float* const Ip = lds + kWinoLdsW + wi * WPI;             // this wave's input rows [<=8][18][8]
        const float* wsrc = P.w + (size_t)nb * nkb * kWinoLdsW + wi * 2048 + 4 * lane;
        const bool outer = (wi == 0 || wi == 3);                  // 5 region rows (every other one), else rows 1..8
        const int nrows = outer ? 5 : 8;
        long long i_src[5];
        int i_dst[5];
        bool i_ok[5], i_use[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            int f = lane + 64 * i;
            int q = f >> 1, hf = f & 1;
            int lr = q / kWinoRW, rx = q - lr * kWinoRW;
            i_use[i] = lr < nrows;
            int ry = outer ? 2 * lr + (wi == 3 ? 1 : 0) : lr + 1;  // region row of private row lr
            int y = y_in0 + ry, x = x_in0 + rx;
            i_ok[i] = i_use[i] && y >= 0 && y < H && x >= 0 && x < W;
            i_src[i] = ((long long)b * HW + (long long)y * W + x) * Cin + 4 * hf;
            i_dst[i] = q * kWinoIS + 4 * hf;
        }
        const int tyl = li >> 3, txl = li & 7;
        // private rows holding patch rows (2*tyl + ra) and (2*tyl + rb)
        const int lra = outer ? tyl : (wi == 1 ? 2 * tyl : 2 * tyl + 1);
        const int lrb = outer ? tyl + 1 : (wi == 1 ? 2 * tyl + 1 : 2 * tyl);
        const int in_a = (lra * kWinoRW + 2 * txl) * kWinoIS + 4 * lh;
        const int in_b = (lrb * kWinoRW + 2 * txl) * kWinoIS + 4 * lh;
        int w_frag[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            int co = nt * 32 + li;
            w_frag[nt] = co * 8 + 4 * (lh ^ ((co >> 3) & 1));
        }
        f32x4 ri[5];
#define FPC_WP_ISSUE(KB)                                                                                      \
    do {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds(                       \
            (const __attribute__((address_space(1))) void*)(wsrc + (size_t)(KB) * kWinoLdsW + 256 * i),      \
            (__attribute__((address_space(3))) void*)(Wp + i * 256), 16, 0, 0);                               \
        _Pragma("unroll") for (int i = 0; i < 5; ++i) ri[i] =                                                 \
            i_ok[i] ? *reinterpret_cast<const f32x4*>(P.in + i_src[i] + 8 * (KB)) : f32x4{0.f, 0.f, 0.f, 0.f}; \
    } while (0)
#define FPC_WP_LAND()                                                                                         \
    do {                                                                                                      \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
        _Pragma("unroll") for (int i = 0; i < 5; ++i)                                                         \
            if (i_use[i]) *reinterpret_cast<f32x4*>(Ip + i_dst[i]) = ri[i];                                   \
    } while (0)
        FPC_WP_ISSUE(0);
        FPC_WP_LAND();
        for (int kb = 0; kb < nkb; ++kb) {
            // every fragment of this K-step into registers
            f32x4 e[4], v[4], u[4][2];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 da = *reinterpret_cast<const f32x4*>(Ip + in_a + c * kWinoIS);
                f32x4 db = *reinterpret_cast<const f32x4*>(Ip + in_b + c * kWinoIS);
                e[c] = da + sgn * db;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u[j][0] = *reinterpret_cast<const f32x4*>(Wp + j * 512 + w_frag[0]);
                u[j][1] = *reinterpret_cast<const f32x4*>(Wp + j * 512 + w_frag[1]);
            }
            v[0] = sub_pk(e[0], e[2]); v[1] = e[1] + e[2]; v[2] = sub_pk(e[2], e[1]); v[3] = sub_pk(e[1], e[3]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // LDS reads complete: the patch may be refilled
            __builtin_amdgcn_sched_barrier(0);
            if (kb + 1 < nkb) FPC_WP_ISSUE(kb + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j][q], u[j][0][q], acc[j][0], 0, 0, 0);
                    acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j][q], u[j][1][q], acc[j][1], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (kb + 1 < nkb) FPC_WP_LAND();
        }
#undef FPC_WP_ISSUE
#undef FPC_WP_LAND
    } else if constexpr (P3) {
        // Measured on the barrier form (s_memtime stamps): while the co-resident wave of a SIMD issues its 32
        // MFMAs back to back, THIS wave's vector instructions (address maths, the input transform) get the
        // vector ALU only at MFMA boundaries — 1600 + 1350 cycles per K-step for ~60 VALU instructions beside
        // 2200 cycles of MFMA issue.  So here every vector instruction that is not an MFMA sits in the shadow
        // of the wave's OWN MFMAs: the next step's fragments are read and transformed between the MFMAs of
        // the second half of the current step, DMA addressing is scalar, and the single barrier of a step sits
        // in the middle of its MFMA block.
        float* const lds_w = lds + 3 * IP3;
        const int swv = __builtin_amdgcn_readfirstlane(wv);
        const float* wbase = P.w + (size_t)nb * nkb * kWinoLdsW + (swv * 4 * 256);     // wave-uniform
        const unsigned lane16 = 4u * lane;                                            // floats
        const float* isrc[2];
        int istep[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int q = (swv + 8 * i) * 32 + (lane >> 1), hf = lane & 1;     // LDS position of this lane's 16 bytes
            int ry = q / kWinoRW, rx = q - ry * kWinoRW;
            int y = y_in0 + ry, x = x_in0 + rx;
            bool ok = q < POS && y >= 0 && y < H && x >= 0 && x < W;
            isrc[i] = ok ? P.in + ((long long)b * HW + (long long)y * W + x) * Cin + 4 * hf : a.zeros;
            istep[i] = ok ? 8 : 0;
        }
#define FPC_P3_ISSUE(KB, BUF)                                                                                 \
    do {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds(                       \
            (const __attribute__((address_space(1))) void*)(wbase + (size_t)(KB) * kWinoLdsW + 256 * i + lane16), \
            (__attribute__((address_space(3))) void*)(lds_w + (BUF) * kWinoLdsW + (swv * 4 + i) * 256), 16, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                       \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)isrc[i],          \
                (__attribute__((address_space(3))) void*)(lds + (BUF) * IP3 + (swv + 8 * i) * 256), 16, 0, 0); \
            isrc[i] += istep[i];                                                                              \
        }                                                                                                     \
    } while (0)
        const int tyl = (li >> 3) + 4 * half, txl = li & 7;
        const int ra = (wi == 0) ? 0 : (wi == 2 ? 2 : 1);
        const int rb = (wi == 0) ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
        const int in_a = ((2 * tyl + ra) * kWinoRW + 2 * txl) * kWinoIS + 4 * lh;
        const int in_b = ((2 * tyl + rb) * kWinoRW + 2 * txl) * kWinoIS + 4 * lh;
        int w_frag[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            int co = nt * 32 + li;
            w_frag[nt] = ((4 * wi) * kWinoBN + co) * 8 + 4 * (lh ^ ((co >> 3) & 1));
        }
        const f32x4 sg4 = {sgn, sgn, sgn, sgn};
#define FPC_P3_MFMA8(J, U0, U1)                                                                               \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                           \
        acc[J][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[J][q], U0[q], acc[J][0], 0, 0, 0);                 \
        acc[J][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[J][q], U1[q], acc[J][1], 0, 0, 0);                 \
    }
        // prologue: stages 0 and 1 in flight, fragments of step 0 transformed
        FPC_P3_ISSUE(0, 0);
        if (nkb > 1) FPC_P3_ISSUE(1, 1);
        if (nkb > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        f32x4 v[4];
        {
            f32x4 e[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                e[c] = __builtin_elementwise_fma(sg4, *reinterpret_cast<const f32x4*>(lds + in_b + c * kWinoIS),
                                                 *reinterpret_cast<const f32x4*>(lds + in_a + c * kWinoIS));
            v[0] = sub_pk(e[0], e[2]); v[1] = e[1] + e[2]; v[2] = sub_pk(e[2], e[1]); v[3] = sub_pk(e[1], e[3]);
        }
        long long stamp[6] = {0, 0, 0, 0, 0, 0};
        const bool dbg = DBG && a.dbg != nullptr;
#define FPC_STAMP(I) do { if (DBG && dbg) { long long now_ = clock64(); stamp[I] += now_ - tprev; tprev = now_; } } while (0)
        long long tprev = dbg ? clock64() : 0;
        const long long c_begin = tprev, r_begin = dbg ? wall_clock64() : 0;
        int cur = 0;
        for (int kb = 0; kb < nkb; ++kb) {
            int nxt = cur + 1 == 3 ? 0 : cur + 1;
            int nx2 = nxt + 1 == 3 ? 0 : nxt + 1;
            const float* Wb = lds_w + cur * kWinoLdsW;
            const float* In = lds + nxt * IP3;
            // ---- first half: xi 0, 1 of this step
            f32x4 u0 = *reinterpret_cast<const f32x4*>(Wb + w_frag[0]);
            f32x4 u1 = *reinterpret_cast<const f32x4*>(Wb + w_frag[1]);
            f32x4 p0 = *reinterpret_cast<const f32x4*>(Wb + w_frag[0] + 1 * kWinoBN * 8);
            f32x4 p1 = *reinterpret_cast<const f32x4*>(Wb + w_frag[1] + 1 * kWinoBN * 8);
            __builtin_amdgcn_s_setprio(1);
            FPC_P3_MFMA8(0, u0, u1)
            __builtin_amdgcn_sched_barrier(0);
            u0 = *reinterpret_cast<const f32x4*>(Wb + w_frag[0] + 2 * kWinoBN * 8);
            u1 = *reinterpret_cast<const f32x4*>(Wb + w_frag[1] + 2 * kWinoBN * 8);
            FPC_P3_MFMA8(1, p0, p1)
            __builtin_amdgcn_sched_barrier(0);
            p0 = *reinterpret_cast<const f32x4*>(Wb + w_frag[0] + 3 * kWinoBN * 8);
            p1 = *reinterpret_cast<const f32x4*>(Wb + w_frag[1] + 3 * kWinoBN * 8);
            FPC_STAMP(0);      // first half: 16 MFMA issued
            // ---- middle: stage kb+1 (issued a whole step ago) must have landed; everyone is past step kb-1,
            //      so stage (kb+2)%3 — read last in step kb-1 — may be refilled
            if (kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FPC_STAMP(1);      // wait for this wave's DMA pieces
            __builtin_amdgcn_s_barrier();
            FPC_STAMP(2);      // barrier
            if (kb + 2 < nkb) FPC_P3_ISSUE(kb + 2, nx2);
            // ---- second half: xi 2, 3, with the NEXT step's input fragments read and transformed in between
            f32x4 da[4], db[4], e[4];
            const bool more = kb + 1 < nkb;
            if (more) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    da[c] = *reinterpret_cast<const f32x4*>(In + in_a + c * kWinoIS);
                    db[c] = *reinterpret_cast<const f32x4*>(In + in_b + c * kWinoIS);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            FPC_P3_MFMA8(2, u0, u1)
            __builtin_amdgcn_sched_barrier(0);
            f32x4 w0 = v[0], w1 = v[1], w2 = v[2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[3][q], p0[q], acc[3][0], 0, 0, 0);
                acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[3][q], p1[q], acc[3][1], 0, 0, 0);
                if (more) {      // 8 vector instructions in the shadow of this MFMA pair
                    if (q == 0) { e[0] = __builtin_elementwise_fma(sg4, db[0], da[0]); e[1] = __builtin_elementwise_fma(sg4, db[1], da[1]); }
                    if (q == 1) { e[2] = __builtin_elementwise_fma(sg4, db[2], da[2]); e[3] = __builtin_elementwise_fma(sg4, db[3], da[3]); }
                    if (q == 2) { w0 = sub_pk(e[0], e[2]); w1 = e[1] + e[2]; }
                    if (q == 3) { w2 = sub_pk(e[2], e[1]); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
            if (more) { v[3] = sub_pk(e[1], e[3]); v[0] = w0; v[1] = w1; v[2] = w2; }
            FPC_STAMP(3);      // DMA issue + second half (16 MFMA + next step's fragments)
            cur = nxt;
        }
        if (dbg && lane == 0) {
            long long* o = a.dbg + ((size_t)blockIdx.x * NW + wv) * 8;
            o[0] = stamp[0]; o[1] = stamp[1] + stamp[2]; o[2] = stamp[3];
            o[3] = clock64() - c_begin; o[4] = wall_clock64() - r_begin; o[5] = nkb; o[6] = c_begin - t_entry;
        }
#undef FPC_STAMP
#undef FPC_P3_ISSUE
#undef FPC_P3_MFMA8
    } else {
    // ---- barrier form.  Both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no
    // ds_write).  Weights: the K-step image (32 KB, already in its LDS layout), every wave moves 32/NW
    // pieces of 1 KB; the four pieces of a group share ONE address register pair and ONE M0 value and differ
    // in the instruction's immediate offset, which the hardware adds to the global AND to the LDS address
    // (measured: tools_dev/glds_offset.hip).  Input: the staged region as 1 KB pieces of 32 positions,
    // out-of-image positions read the zero page.  The K loop is unrolled by two so that every LDS address
    // of a step is register + immediate.  Per K-step this leaves ~8 address instructions beside the
    // transform's 16 packed ones (the register-staged form had ~60, and a wave's vector instructions crawl
    // while its SIMD partner issues MFMAs — see the P3 comment above).
    constexpr int NPIECE = 32 / NW, NG = NPIECE / 4;
    const int swv = __builtin_amdgcn_readfirstlane(wv);
    float* const lds_w = lds + 2 * LINP;
    // Staging addresses are SGPR base + 32-bit VGPR offset: beside a SIMD partner that issues MFMAs back to
    // back, a global_load* with a 64-bit VGPR address waits like a vector-ALU instruction (one per MFMA; a pure
    // MFMA partner starves it: 1550 cycles against 12 for the SGPR-base form — tools_dev/dma_vs_mfma.hip).
    // The bases advance on the scalar unit, the lane offsets never change: no vector instruction per K-step.
    const float* wsb = P.w + (size_t)nb * nkb * kWB + swv * NPIECE * 256;           // wave-uniform: this wave's pieces of step 0
    // BF3: the {b3} image's 16 pieces.  FPC_WINO_B3_OLDER: all of them go to the older half (waves 0-3, four each) — in the
    // staggered loop the younger half's staging burst + matrix block is the longer chain (stamps: 1093 + 1797 cycles against
    // 659 + 1668), so the older half takes 10-11 of a step's pieces and the younger 6-7 instead of 8-9 each.
    const float* wsb3 = P.w + (size_t)nb * nkb * kWB + 8192 + (FPC_WINO_B3_OLDER ? (swv & 3) * 1024 : swv * 512);
    const float* isb = P.in + (size_t)b * HW * Cin;                                // image base, + 8 floats per step
    const unsigned wvo = 16u * lane;                                               // bytes
    unsigned ivo[NIN];
    bool iok[NIN];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        int ry, rx, hf;
        bool inreg;
        if constexpr (PERM) {
            const int slot = (swv + IST * i) * 64 + lane;             // 16-byte unit this lane's DMA data lands in
            const int blk = slot >> 4, res = slot & 15, g = blk >> 3;
            const int ah = (g / 3) * 4 + (res & 3), qh = (g % 3) * 4 + (res >> 2);
            hf = blk & 1;
            ry = 2 * ah + ((blk >> 2) & 1); rx = 2 * qh + ((blk >> 1) & 1);
            inreg = swv + IST * i < NPI && ah <= 8 && qh <= 8 && (!INO || swv < 4);
        } else {
            const int q = (swv + NW * i) * 32 + (lane >> 1);          // LDS position of this lane's 16 bytes
            hf = lane & 1;
            ry = q / kWinoRW; rx = q - ry * kWinoRW;
            inreg = q < POS && (i == 0 || swv + NW < NPI);
        }
        int y = y_in0 + ry, x = x_in0 + rx;
        iok[i] = inreg && y >= 0 && y < H && x >= 0 && x < W;
        ivo[i] = iok[i] ? (unsigned)((((size_t)y * W + x) * Cin + 4 * hf) * sizeof(float)) : 0u;
    }
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#define FPC_LDS_ADDR(PTR) ((unsigned)(size_t)(__attribute__((address_space(3))) void*)(PTR))
    // weights of one K-step -> weight buffer BUF: four 1 KB pieces per group share base, offset register and M0
    // (the immediate offset moves the global AND the LDS address, tools_dev/glds_offset.hip)
#define FPC_WB_ISSUE_W(BUF)                                                                                   \
    do {                                                                                                      \
        _Pragma("unroll") for (int g = 0; g < NG; ++g)                                                        \
            asm volatile("s_mov_b32 m0, %0\n s_nop 0\n"                                                       \
                         "global_load_lds_dwordx4 %1, %2\n global_load_lds_dwordx4 %1, %2 offset:1024\n"      \
                         "global_load_lds_dwordx4 %1, %2 offset:2048\n global_load_lds_dwordx4 %1, %2 offset:3072\n" \
                         :: "s"(FPC_LDS_ADDR(lds_w + (BUF) * kWB + (swv * NPIECE + 4 * g) * 256)), "v"(wvo),       \
                            "s"(wsb + 1024 * g) : "memory", "m0");                                            \
        if constexpr (BF3 && !FPC_WINO_B3_OLDER)                                                              \
            asm volatile("s_mov_b32 m0, %0\n s_nop 0\n"                                                       \
                         "global_load_lds_dwordx4 %1, %2\n global_load_lds_dwordx4 %1, %2 offset:1024\n"      \
                         :: "s"(FPC_LDS_ADDR(lds_w + (BUF) * kWB + 8192 + swv * 512)), "v"(wvo), "s"(wsb3) : "memory", "m0"); \
        if constexpr (BF3 && FPC_WINO_B3_OLDER)                                                               \
            if (swv < 4)                                                                                      \
                asm volatile("s_mov_b32 m0, %0\n s_nop 0\n"                                                   \
                             "global_load_lds_dwordx4 %1, %2\n global_load_lds_dwordx4 %1, %2 offset:1024\n"  \
                             "global_load_lds_dwordx4 %1, %2 offset:2048\n global_load_lds_dwordx4 %1, %2 offset:3072\n" \
                             :: "s"(FPC_LDS_ADDR(lds_w + (BUF) * kWB + 8192 + swv * 1024)), "v"(wvo), "s"(wsb3) : "memory", "m0"); \
    } while (0)
    // input region of one K-step -> input buffer BUF (in-image lanes only)
#define FPC_WB_ISSUE_IN(BUF)                                                                                  \
    do {                                                                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < NIN; ++i_)                                                    \
            if (iok[i_]) asm volatile("s_mov_b32 m0, %0\n s_nop 0\n global_load_lds_dwordx4 %1, %2\n"         \
                                      :: "s"(FPC_LDS_ADDR(lds + (BUF) * LINP + (swv + IST * i_) * 256)), "v"(ivo[i_]), "s"(isb) : "memory", "m0"); \
    } while (0)

    // ---- fragment addressing
    const int tyl = (li >> 3) + 4 * half, txl = li & 7;        // this lane's tile inside the patch
    // row pair (ra, rb) and sign of B^T row wi:  0: d0-d2   1: d1+d2   2: d2-d1   3: d1-d3
    const int ra = (wi == 0) ? 0 : (wi == 2 ? 2 : 1);
    const int rb = (wi == 0) ? 2 : (wi == 1 ? 2 : (wi == 2 ? 1 : 3));
    // float offsets of the four positions (2 * txl + c, c = 0..3) of region rows 2 * tyl + ra / rb: [c >> 1] + (c & 1) * in_cs
    int in_a[2], in_b[2];
    constexpr int in_cs = PERM ? 2 * 16 * 4 : kWinoIS;
    if constexpr (PERM) {
        auto unit = [&](int r, int ch) {      // row 2 * tyl + r, column 2 * (txl + ch)
            const int ah = tyl + (r >> 1), qh = txl + ch;
            return ((((ah >> 2) * 3 + (qh >> 2)) * 8 + (r & 1) * 4 + lh) * 16 + 4 * (qh & 3) + (ah & 3)) * 4;
        };
        in_a[0] = unit(ra, 0); in_a[1] = unit(ra, 1); in_b[0] = unit(rb, 0); in_b[1] = unit(rb, 1);
    } else {
        in_a[0] = ((2 * tyl + ra) * kWinoRW + 2 * txl) * kWinoIS + 4 * lh; in_a[1] = in_a[0] + 2 * kWinoIS;
        in_b[0] = ((2 * tyl + rb) * kWinoRW + 2 * txl) * kWinoIS + 4 * lh; in_b[1] = in_b[0] + 2 * kWinoIS;
    }
    int w_frag[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        int co = nt * 32 + li;
        w_frag[nt] = ((4 * wi) * kWinoBN + co) * 8 + 4 * (lh ^ ((co >> 3) & 1));   // halves swapped on odd 8-channel groups (k_wino_pack)
    }
    const f32x4 sg4 = {sgn, sgn, sgn, sgn};

    // Software pipeline (measured on this kernel with s_memtime stamps, tools_dev/wino_stamps.py):
    //  * a burst of LDS-DMA issues holds a wave ~1500 cycles per K-step (the CU's L2 -> LDS path moves ~62 B/clk
    //    whatever the instruction form, tools_dev/dma_rate.hip) and costs as much when the instructions are
    //    spread between the wave's MFMAs (~100 cycles each there) — so the burst stays a phase of its own,
    //    beside the SIMD partner's MFMA phase;
    //  * vector ALU instructions of a wave whose partner issues MFMAs back to back advance one per MFMA
    //    (~64 cycles each), but cost 2-4 cycles between the wave's OWN MFMAs — so the input transform of step
    //    k+1 (8 LDS reads, 16 packed instructions) runs inside step k's MFMA block.
    // Step k therefore: [DMA: weights k+1 -> W[cur^1], input k+2 -> I[cur]] [32 MFMAs of step k on v and
    // W[cur], with the fragments of step k+1 read from I[cur^1] and transformed in between] [wait, barrier].
    // Input k+2 may overwrite I[cur]: step k's fragments were read from it during step k-1.  The body has no
    // branch: past the last step the sources stop advancing, so the final steps stage (and transform) the
    // last step's operands once more into buffers nobody reads.
    FPC_WB_ISSUE_W(0);      // (the weight buffers are not touched by the zero fill below: the first 48 KB are on their way meanwhile)
    // out-of-image positions are never written by the DMA (inactive lanes): a patch that reaches over the image border zeroes
    // both input buffers once.  An interior patch (workgroup-uniform) skips the fill and its barrier: every position the
    // fragment reads touch is rewritten by every step's DMA (the permuted image's padding units are never read).
    if (y_in0 < 0 || x_in0 < 0 || y_in0 + RH > H || x_in0 + kWinoRW > W) {
        for (int i = t; i < 2 * LINP / 4; i += 64 * NW) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    FPC_WB_ISSUE_IN(0);
    if (nkb > 1) { wsb += kWB; wsb3 += kWB; isb += 8; }
    FPC_WB_ISSUE_IN(1);
    if (nkb > 2) isb += 8;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x4 v[4];      // transformed input fragments of the current step: lanes 0-31 carry ci = q, lanes 32-63 ci = 4 + q
    {
        f32x4 e[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            e[c] = __builtin_elementwise_fma(sg4, *reinterpret_cast<const f32x4*>(lds + in_b[c >> 1] + (c & 1) * in_cs),
                                             *reinterpret_cast<const f32x4*>(lds + in_a[c >> 1] + (c & 1) * in_cs));
        v[0] = sub_pk(e[0], e[2]); v[1] = e[1] + e[2]; v[2] = sub_pk(e[2], e[1]); v[3] = sub_pk(e[1], e[3]);
    }
    // BF3: the three bf16 pieces of the transformed fragments (four channels per piece and xi), and the {b3} fragment offsets
    u32x2 pa[4][3];
    int w3_frag[2];
    int w_fragh[2] = {w_frag[0], w_frag[1]};      // BF3: the same offsets, opaque (see the operand tuples in the K loop)
    if constexpr (BF3) {
#pragma unroll
        for (int j = 0; j < 4; ++j) split_bf3(v[j], pa[j][0], pa[j][1], pa[j][2]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            int co = nt * 32 + li;
            w3_frag[nt] = 8192 + ((4 * wi) * kWinoBN + co) * 4 + 2 * (lh ^ ((co >> 4) & 1));      // halves swapped per 16 channels (k_wino_pack_bf3)
        }
        // opaque copies: no forwarding from the 16-byte reads, and no pairing of the two {b3} reads into one
        // ds_read2 (its results would sit in adjacent registers and have to be moved into their operand tuples)
        asm volatile("" : "+v"(w_fragh[0]), "+v"(w_fragh[1]), "+v"(w3_frag[1]));
    }
    __syncthreads();       // I[0] is refilled by step 0's DMA
    long long stamp[6] = {0, 0, 0, 0, 0, 0};
    const bool dbg = DBG && a.dbg != nullptr;
#define FPC_STAMP(I) do { if (DBG && dbg) { long long now_ = clock64(); stamp[I] += now_ - tprev; tprev = now_; } } while (0)
    long long tprev = dbg ? clock64() : 0;
    const long long c_begin = tprev, r_begin = dbg ? wall_clock64() : 0;
    int cur = 0;
#pragma unroll 1
    for (int kb = 0; kb < nkb; ++kb) {
        // BF3, stagger (FPC_WINO_STAGGER): SIMD partners are waves w and w + 4.  The younger half stages first and computes
        // second, the older half computes first and stages afterwards, so that on every SIMD one wave's matrix block runs
        // beside the other's staging burst instead of both doing the same phase in lockstep.  Same buffers, same hazards:
        // W[cur ^ 1] and I[cur] were last read in step kb - 1, whichever half writes them in step kb.
        const bool stage_first = !(BF3 && FPC_WINO_STAGGER) || swv >= 4;
        if (stage_first) {
            FPC_WB_ISSUE_W(cur ^ 1);
            FPC_WB_ISSUE_IN(cur);
        }
        FPC_STAMP(0);      // issue of the staging loads
        const float* In = lds + (cur ^ 1) * LINP;
        const float* Wb = lds_w + cur * kWB;
        f32x4 da[4], db[4], e[4], vn[4];
        // MFMA issue ahead of the co-resident workgroup's staging.  BF3 (one workgroup per CU): the SIMD partners are waves w
        // and w + 4 of this workgroup, and the younger one (w + 4) loses the arbitration in the staging phase AND here
        // (stamps: 1520 + 1900 cycles per K-step against 1000 + 1500, the older waves then wait 1200 at the barrier) —
        // FPC_WINO_PRIO_HI for the younger half in this block evens the two out.
        if (BF3 && swv >= 4) __builtin_amdgcn_s_setprio(FPC_WINO_PRIO_HI);
        else __builtin_amdgcn_s_setprio(1);
        if constexpr (BF3) {
            // Operand tuples without register moves: {b1, b2} is one 16-byte read; {b3, b1} is built from two 8-byte reads
            // that land in the two halves of one register tuple (the b1 half read again through an offset the compiler cannot
            // see through, or it would forward the 16-byte read and copy) — 8 LDS instructions more, 32 vector instructions fewer per K-step
            // of a wave, in a block that is bound by vector issue (190 vector instructions beside 24 matrix instructions).
            u32x4 u0 = *reinterpret_cast<const u32x4*>(Wb + w_frag[0]), u1 = *reinterpret_cast<const u32x4*>(Wb + w_frag[1]);
            u32x2 t0 = *reinterpret_cast<const u32x2*>(Wb + w3_frag[0]), t1 = *reinterpret_cast<const u32x2*>(Wb + w3_frag[1]);
            u32x2 h0 = *reinterpret_cast<const u32x2*>(Wb + w_fragh[0]), h1 = *reinterpret_cast<const u32x2*>(Wb + w_fragh[1]);
            asm volatile("" ::: "memory");      // keeps these reads from being paired (ds_read2) with xi 1's: paired results sit in adjacent registers
            u32x2 pn[4][3];
#define FPC_WINO_BF3_MFMA(A, B0, B1)                                                                               \
    do {                                                                                                      \
        acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B0), acc[j][0], 0, 0, 0); \
        acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B1), acc[j][1], 0, 0, 0); \
    } while (0)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4 n0 = u0, n1 = u1;
                u32x2 m0 = t0, m1 = t1, g0 = h0, g1 = h1;
                if (j < 3) {
                    n0 = *reinterpret_cast<const u32x4*>(Wb + w_frag[0] + (j + 1) * kWinoBN * 8);
                    n1 = *reinterpret_cast<const u32x4*>(Wb + w_frag[1] + (j + 1) * kWinoBN * 8);
                    m0 = *reinterpret_cast<const u32x2*>(Wb + w3_frag[0] + (j + 1) * kWinoBN * 4);
                    m1 = *reinterpret_cast<const u32x2*>(Wb + w3_frag[1] + (j + 1) * kWinoBN * 4);
                    g0 = *reinterpret_cast<const u32x2*>(Wb + w_fragh[0] + (j + 1) * kWinoBN * 8);
                    g1 = *reinterpret_cast<const u32x2*>(Wb + w_fragh[1] + (j + 1) * kWinoBN * 8);
                }
                if (j < 2) {
#pragma unroll
                    for (int c = 2 * j; c < 2 * j + 2; ++c) {
                        da[c] = *reinterpret_cast<const f32x4*>(In + in_a[c >> 1] + (c & 1) * in_cs);
                        db[c] = *reinterpret_cast<const f32x4*>(In + in_b[c >> 1] + (c & 1) * in_cs);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 A0 = {pa[j][0][0], pa[j][0][1], pa[j][0][0], pa[j][0][1]};
                const u32x4 A1 = {pa[j][1][0], pa[j][1][1], pa[j][1][0], pa[j][1][1]};
                const u32x4 A2 = {pa[j][0][0], pa[j][0][1], pa[j][2][0], pa[j][2][1]};
                const u32x4 C0 = {t0[0], t0[1], h0[0], h0[1]}, C1 = {t1[0], t1[1], h1[0], h1[1]};
                FPC_WINO_BF3_MFMA(A0, u0, u1);      // a1 b1 + a1 b2
                // the next step's fragments in the shadow of this wave's own MFMAs: transform at xi 0 / 1, split at xi 2 / 3
                if (j == 0) { e[0] = fma_s4(sgn, db[0], da[0]); e[1] = fma_s4(sgn, db[1], da[1]); }
                if (j == 1) { e[2] = fma_s4(sgn, db[2], da[2]); e[3] = fma_s4(sgn, db[3], da[3]); }
                if (j == 2) split_bf3(vn[0], pn[0][0], pn[0][1], pn[0][2]);
                if (j == 3) split_bf3(vn[2], pn[2][0], pn[2][1], pn[2][2]);
                __builtin_amdgcn_sched_barrier(0);
                FPC_WINO_BF3_MFMA(A1, u0, u1);      // a2 b1 + a2 b2
                if (j == 1) { vn[0] = sub_s4(e[0], e[2]); vn[1] = add_s4(e[1], e[2]); }
                if (j == 2) split_bf3(vn[1], pn[1][0], pn[1][1], pn[1][2]);
                if (j == 3) split_bf3(vn[3], pn[3][0], pn[3][1], pn[3][2]);
                __builtin_amdgcn_sched_barrier(0);
                FPC_WINO_BF3_MFMA(A2, C0, C1);      // a1 b3 + a3 b1
                if (j == 1) { vn[2] = sub_s4(e[2], e[1]); vn[3] = sub_s4(e[1], e[3]); }
                __builtin_amdgcn_sched_barrier(0);
                u0 = n0; u1 = n1; t0 = m0; t1 = m1; h0 = g0; h1 = g1;
                // stagger: the compute-first half stages after xi FPC_WINO_LATE_AT of its matrix block (3 = after the block): the
                // pieces land while the rest of the block runs (3415 -> 3170 cycles per K-step against staging after the block, 3123
                // after xi 0 once the older half also carries the {b3} pieces; the stage-first half staging inside its block as
                // well: 3500-4100)
                if (j == FPC_WINO_LATE_AT && FPC_WINO_LATE_AT < 3 && !stage_first) {
                    FPC_WB_ISSUE_W(cur ^ 1);
                    FPC_WB_ISSUE_IN(cur);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#undef FPC_WINO_BF3_MFMA
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { pa[j][0] = pn[j][0]; pa[j][1] = pn[j][1]; pa[j][2] = pn[j][2]; }
        } else {
        f32x4 u0 = *reinterpret_cast<const f32x4*>(Wb + w_frag[0]);
        f32x4 u1 = *reinterpret_cast<const f32x4*>(Wb + w_frag[1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // weight fragments of xi j+1 are requested before xi j's MFMAs (LDS latency behind 8 MFMAs)
            f32x4 n0 = u0, n1 = u1;
            if (j < 3) {
                n0 = *reinterpret_cast<const f32x4*>(Wb + w_frag[0] + (j + 1) * kWinoBN * 8);
                n1 = *reinterpret_cast<const f32x4*>(Wb + w_frag[1] + (j + 1) * kWinoBN * 8);
            }
            if (j < 2) {                    // raw fragments of the next step, two channel pairs per xi block
#pragma unroll
                for (int c = 2 * j; c < 2 * j + 2; ++c) {
                    da[c] = *reinterpret_cast<const f32x4*>(In + in_a[c >> 1] + (c & 1) * in_cs);
                    db[c] = *reinterpret_cast<const f32x4*>(In + in_b[c >> 1] + (c & 1) * in_cs);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j][q], u0[q], acc[j][0], 0, 0, 0);
                acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j][q], u1[q], acc[j][1], 0, 0, 0);
                // 2 packed instructions in the shadow of this MFMA pair
                if (j < 2 && q >= 2) e[2 * j + q - 2] = __builtin_elementwise_fma(sg4, db[2 * j + q - 2], da[2 * j + q - 2]);
                if (j == 2 && q == 0) vn[0] = sub_pk(e[0], e[2]);
                if (j == 2 && q == 1) vn[1] = e[1] + e[2];
                if (j == 2 && q == 2) vn[2] = sub_pk(e[2], e[1]);
                if (j == 2 && q == 3) vn[3] = sub_pk(e[1], e[3]);
                __builtin_amdgcn_sched_barrier(0);
            }
            u0 = n0; u1 = n1;
        }
        __builtin_amdgcn_s_setprio(0);
        v[0] = vn[0]; v[1] = vn[1]; v[2] = vn[2]; v[3] = vn[3];
        }
        FPC_STAMP(2);      // MFMA issue + the next step's fragments (not completion)
        if (!stage_first && !(BF3 && FPC_WINO_LATE_AT < 3)) {
            FPC_WB_ISSUE_W(cur ^ 1);
            FPC_WB_ISSUE_IN(cur);
        }
        wsb += kb + 2 < nkb ? kWB : 0;
        wsb3 += kb + 2 < nkb ? kWB : 0;
        isb += kb + 3 < nkb ? 8 : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA pieces have landed
        FPC_STAMP(3);
        __syncthreads();                                      // everybody's have; this step's reads are done
        FPC_STAMP(4);      // barrier
        cur ^= 1;
    }
    if (dbg && lane == 0) {
        long long* o = a.dbg + ((size_t)blockIdx.x * NW + wv) * 8;
        o[0] = stamp[0]; o[1] = stamp[3] + stamp[4]; o[2] = stamp[2];      // issue of the staging loads | wait + barrier | MFMA block
        o[3] = clock64() - c_begin;            // shader-clock ticks of the whole K loop
        o[6] = c_begin - t_entry;              // kernel entry -> K loop
        o[4] = wall_clock64() - r_begin;       // 100 MHz reference ticks of the same span
        o[5] = nkb;
    }
#undef FPC_STAMP
#undef FPC_WB_ISSUE_W
#undef FPC_WB_ISSUE_IN
#undef FPC_LDS_ADDR
#pragma clang diagnostic pop

    }

    const long long t_kend = DBG ? clock64() : 0;
    // ---- output transform.  Column part inside the wave: z0 = m0 + m1 + m2, z1 = m1 - m2 - m3;
    // row part across the four transform-row waves through LDS: y0 = z[0] + z[1] + z[2], y1 = z[1] - z[2] - z[3].
    // LDS image Z[row i][cc][tile NT][co 32], one 32-channel half (nt) at a time.
    const int ot = t >> 3, oc4 = (t & 7) * 4;                 // output stage: thread = one tile x 4 channels
    const int oty = ty0 + (ot >> 3), otx = tx0 + (ot & 7);
    // The epilogue's own global reads — folded-BatchNorm scale / shift and the residual of this thread's 2 x 2 outputs, for both
    // 32-channel halves — are requested BEFORE the first barrier of the output transform: behind it they were issued after the
    // second barrier of each half and their latency stood exposed twice per workgroup (one workgroup per CU: nothing overlaps it;
    // a residual convolution of ResNet-34's layer1 took 266 us against 205 us without residual at batch 32).
    f32x4 e_sc[2], e_sh[2], e_res[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = nb * kWinoBN + nt * 32 + oc4;
        e_sc[nt] = P.scale ? *reinterpret_cast<const f32x4*>(P.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
        e_sh[nt] = P.shift ? *reinterpret_cast<const f32x4*>(P.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int y = 2 * oty + (q >> 1), x = 2 * otx + (q & 1);
            e_res[nt][q] = (P.res && y < H && x < W) ? *reinterpret_cast<const f32x4*>(P.res + ((size_t)b * HW + (size_t)y * W + x) * Cout + n)
                                                     : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int m = half * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float m0 = acc[0][nt][r], m1 = acc[1][nt][r], m2 = acc[2][nt][r], m3 = acc[3][nt][r];
            lds[((wi * 2 + 0) * NT + m) * 32 + li] = m0 + m1 + m2;
            lds[((wi * 2 + 1) * NT + m) * 32 + li] = m1 - m2 - m3;
        }
        __syncthreads();
        const int n = nb * kWinoBN + nt * 32 + oc4;
        f32x4 z[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) z[i][cc] = *reinterpret_cast<const f32x4*>(lds + ((i * 2 + cc) * NT + ot) * 32 + oc4);
        const f32x4 sc = e_sc[nt], sh = e_sh[nt];
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                int y = 2 * oty + rr, x = 2 * otx + cc;
                if (y >= H || x >= W) continue;
                f32x4 val = rr == 0 ? z[0][cc] + z[1][cc] + z[2][cc] : z[1][cc] - z[2][cc] - z[3][cc];
                if (P.scale) val = val * sc;
                val = val + sh;
                size_t o = ((size_t)b * HW + (size_t)y * W + x) * Cout + n;
                if (P.res) val += e_res[nt][2 * rr + cc];
                if (a.relu) { val[0] = fmaxf(val[0], 0.f); val[1] = fmaxf(val[1], 0.f); val[2] = fmaxf(val[2], 0.f); val[3] = fmaxf(val[3], 0.f); }
                *reinterpret_cast<f32x4*>(P.out + o) = val;
                s1 += val;
                s2 += val * val;
            }
        if (P.gn_part) {
            // per-channel sums of this workgroup's outputs.  A wave holds 8 tiles (lane bits 3-5) of 8 channel quads
            // (lane bits 0-2): butterfly over the tile bits, then the NW waves' sums through LDS in wave order
            // (a serial walk of the NT tile slots by 32 threads cost ~4000 cycles per 32-channel half).
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { s1[k] += __shfl_xor(s1[k], o, 64); s2[k] += __shfl_xor(s2[k], o, 64); }
            }
            __syncthreads();
            float* red = lds;                                  // [NW waves][32 ch][2]
            if (lane < 8) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { red[(wv * 32 + oc4 + k) * 2] = s1[k]; red[(wv * 32 + oc4 + k) * 2 + 1] = s2[k]; }
            }
            __syncthreads();
            if (t < 32) {
                float u1 = 0.f, u2 = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) { u1 += red[(w * 32 + t) * 2]; u2 += red[(w * 32 + t) * 2 + 1]; }
                int Pn = a.tbx * a.tby;
                float* g = P.gn_part + (((size_t)b * Pn + by * a.tbx + bx) * Cout + nb * kWinoBN + nt * 32 + t) * 2;
                g[0] = u1; g[1] = u2;
            }
        }
    }
    if (DBG && a.dbg != nullptr && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        a.dbg[((size_t)blockIdx.x * NW + wv) * 8 + 7] = clock64() - t_kend;      // K loop end -> last store acknowledged
    }
}

// OIHW 3x3 weights -> U = G g G^T, packed [Cout/64][Cin/8][16 xi][64 co][8 ci] (one K-step image = 32 KB)
__global__ __launch_bounds__(256) void k_wino_pack(const float* __restrict__ w, float* __restrict__ out, int Cout,
                                                   int Cin) {
    long long total = (long long)Cout * Cin;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        int ci = (int)(g % Cin), co = (int)(g / Cin);
        const float* k = w + ((size_t)co * Cin + ci) * 9;
        float gg[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float g0 = k[c], g1 = k[3 + c], g2 = k[6 + c];
            gg[0][c] = g0;
            gg[1][c] = 0.5f * (g0 + g1 + g2);
            gg[2][c] = 0.5f * (g0 - g1 + g2);
            gg[3][c] = g2;
        }
        int nb = co >> 6, col = co & 63, kb = ci >> 3, cil = ci & 7;
        // the step image IS the LDS image: halves (4 channels) swapped on odd 8-channel groups
        float* dst = out + (((size_t)nb * (Cin >> 3) + kb) * 16) * 512 + col * 8 + (cil ^ (4 * ((col >> 3) & 1)));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float r0 = gg[i][0], r1 = gg[i][1], r2 = gg[i][2];
            dst[(4 * i + 0) * 512] = r0;
            dst[(4 * i + 1) * 512] = 0.5f * (r0 + r1 + r2);
            dst[(4 * i + 2) * 512] = 0.5f * (r0 - r1 + r2);
            dst[(4 * i + 3) * 512] = r2;
        }
    }
}

// The same U, every value split exactly into three bf16 pieces (truncation split, as split_bf3), packed per K-step as
// [Cout/64][Cin/8][ 16 xi x 64 co x {half slot: b1 x 4 ch, b2 x 4 ch} (32 KB, the f32 image's addressing) | 16 xi x 64 co x
// {half slot: b3 x 4 ch} (16 KB) ]; half slots swapped on odd 8-channel (main) / 16-channel ({b3}) groups of co.
__global__ __launch_bounds__(256) void k_wino_pack_bf3(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout,
                                                       int Cin) {
    long long total = (long long)Cout * Cin;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        int ci = (int)(g % Cin), co = (int)(g / Cin);
        const float* k = w + ((size_t)co * Cin + ci) * 9;
        float gg[4][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float g0 = k[c], g1 = k[3 + c], g2 = k[6 + c];
            gg[0][c] = g0;
            gg[1][c] = 0.5f * (g0 + g1 + g2);
            gg[2][c] = 0.5f * (g0 - g1 + g2);
            gg[3][c] = g2;
        }
        const int nb = co >> 6, col = co & 63, kb = ci >> 3, cil = ci & 7, hw = cil >> 2, e = cil & 3;
        unsigned short* img = out + ((size_t)nb * (Cin >> 3) + kb) * (12288 * 2);
        unsigned short* dm = img + col * 16 + 8 * (hw ^ ((col >> 3) & 1)) + e;                      // + xi * 1024;  b2 at + 4
        unsigned short* d3 = img + 8192 * 2 + col * 8 + 4 * (hw ^ ((col >> 4) & 1)) + e;            // + xi * 512
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float r0 = gg[i][0], r1 = gg[i][1], r2 = gg[i][2];
            const float u[4] = {r0, 0.5f * (r0 + r1 + r2), 0.5f * (r0 - r1 + r2), r2};
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                const int xi = 4 * i + jx;
                const float x = u[jx];
                const unsigned xb = __builtin_bit_cast(unsigned, x) & 0xFFFF0000u;
                const float r = x - __builtin_bit_cast(float, xb);
                const unsigned rb = __builtin_bit_cast(unsigned, r) & 0xFFFF0000u;
                const float q = r - __builtin_bit_cast(float, rb);
                dm[xi * 1024] = (unsigned short)(xb >> 16);
                dm[xi * 1024 + 4] = (unsigned short)(rb >> 16);
                d3[xi * 512] = (unsigned short)(__builtin_bit_cast(unsigned, q) >> 16);
            }
        }
    }
}

// 256 bytes of zeros that live in the code object (zero-initialised at load, never written): the zero page of the all-DMA and
// split-precision Winograd forms for callers that have no plan workspace (fpc_conv2d: one memset per call before)
__device__ __attribute__((aligned(256))) float g_fpc_zero_page[64];
const float* zero_page() {      // looked up per call: the address belongs to the CURRENT device
    void* q = nullptr;
    return hipGetSymbolAddress(&q, HIP_SYMBOL(g_fpc_zero_page)) == hipSuccess ? static_cast<const float*>(q) : nullptr;
}

int launch_conv_wino(const WinoArgs& a, int groups, hipStream_t s) {
    if (groups < 1 || groups > kMaxGroup || a.Cin % 8 != 0 || a.Cout % kWinoBN != 0 || (a.waves != 4 && a.waves != 8))
        return FPC_EINVAL;
    dim3 grid(a.tbx * a.tby * a.B * (a.Cout / kWinoBN) * groups);
    if (a.variant != 1 && !a.zeros) return FPC_EINVAL;
    if ((long long)a.H * a.W * a.Cin * (long long)sizeof(float) >= (1LL << 32)) return FPC_EINVAL;   // 32-bit lane offsets inside one image
    if (a.waves == 8 && a.variant == 3) {
        if (a.dbg) hipLaunchKernelGGL((k_conv_wino<8, false, false, true, true>), grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((k_conv_wino<8, false, false, false, true>), grid, dim3(512), 0, s, a);
    } else if (a.waves == 8 && a.variant == 2) {
        if (a.dbg) hipLaunchKernelGGL((k_conv_wino<8, false, true, true>), grid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((k_conv_wino<8, false, true>), grid, dim3(512), 0, s, a);
    } else if (a.waves == 8 && a.dbg) hipLaunchKernelGGL((k_conv_wino<8, false, false, true>), grid, dim3(512), 0, s, a);
    else if (a.waves == 8) hipLaunchKernelGGL((k_conv_wino<8, false, false>), grid, dim3(512), 0, s, a);
    else if (a.variant == 1) hipLaunchKernelGGL((k_conv_wino<4, true, false>), grid, dim3(256), 0, s, a);
    else if (a.dbg) hipLaunchKernelGGL((k_conv_wino<4, false, false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_conv_wino<4, false, false>), grid, dim3(256), 0, s, a);
    return check_launch();
}

int launch_wino_pack(const float* w_oihw, float* packed, int Cout, int Cin, hipStream_t s) {
    if (Cin % 8 != 0 || Cout % kWinoBN != 0) return FPC_EINVAL;
    hipLaunchKernelGGL(k_wino_pack, dim3(stream_grid((long long)Cout * Cin)), dim3(256), 0, s, w_oihw, packed, Cout, Cin);
    return check_launch();
}

// split-precision image: 24 * Cout * Cin floats (every byte is written)
int launch_wino_pack_bf3(const float* w_oihw, float* packed, int Cout, int Cin, hipStream_t s) {
    if (Cin % 8 != 0 || Cout % kWinoBN != 0) return FPC_EINVAL;
    hipLaunchKernelGGL(k_wino_pack_bf3, dim3(stream_grid((long long)Cout * Cin)), dim3(256), 0, s, w_oihw,
                       reinterpret_cast<unsigned short*>(packed), Cout, Cin);
    return check_launch();
}

int launch_pack_weight(const float* w, float* packed, int Cout, int Cin, int Cinp, int Kh, int Kw, int Kwp, int Npad,
                       int Kpad, hipStream_t s) {
    if (Kwp < Kw || Cinp < Cin) return FPC_EINVAL;
    hipLaunchKernelGGL(k_pack_weight, dim3(stream_grid((long long)Npad * Kpad)), dim3(256), 0, s, w, packed, Cout, Cin,
                       Cinp, Kh, Kw, Kwp, Npad, Kpad);
    return check_launch();
}

int launch_pack_weight_bf3(const float* w, float* packed, int Cout, int Cin, int Cinp, int Kh, int Kw, int Kwp, int Npad,
                           int Kpad, hipStream_t s) {
    if (Kwp < Kw || Cinp < Cin) return FPC_EINVAL;
    hipLaunchKernelGGL(k_pack_weight_bf3, dim3(stream_grid((long long)Npad * Kpad)), dim3(256), 0, s, w,
                       reinterpret_cast<unsigned short*>(packed + (size_t)Npad * Kpad), Cout, Cin, Cinp, Kh, Kw, Kwp, Npad, Kpad);
    return check_launch();
}

int launch_nchw3_to_nhwc4(const float* x, float* out, int B, int HW, hipStream_t s) {
    hipLaunchKernelGGL(k_nchw3_to_nhwc4, dim3(stream_grid((long long)B * HW)), dim3(256), 0, s, x, out, B, HW);
    return check_launch();
}

int launch_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int C,
                   float* scale, float* shift, hipStream_t s) {
    hipLaunchKernelGGL(k_fold_bn, dim3(cdiv(C, 256)), dim3(256), 0, s, gamma, beta, mean, var, eps, C, scale, shift);
    return check_launch();
}

}  // namespace fpc
