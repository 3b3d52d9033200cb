// net_kernels.hpp — argument structs and launch wrappers of the backbone kernels
// (net_kernels.hip), used by the engine (net.hip).  gfx950 only.
#pragma once
#include "common.hpp"

namespace fpc {

#ifndef FPC_IGEMM_DMA_B
#define FPC_IGEMM_DMA_B 1      // split-precision k_conv_igemm: B rows staged from the bf16 planes by LDS-DMA (0: from the f32 image through registers)
#endif

constexpr int kMaxGroup = 4;      // the four FPN decoders run as one grouped launch
constexpr int kConvBK = 32;       // K-step of the implicit GEMM (floats)
constexpr int kConvNAlign = 128;  // packed weight rows are padded to a multiple of this
constexpr int kConvTickets = 16384;   // arrival counters a plan's workspace holds (fused split-K needs one per output tile)

struct ConvPtrs {
    const float* in;      // input activation
    const float* w;       // packed weights [Npad][Kpad] (OHWI, zero padded)
    float* out;           // NHWC [B,Ho,Wo,Cout]
    const float* scale;   // per-channel multiplier (folded BatchNorm) or null
    const float* shift;   // per-channel addend (folded BatchNorm / conv bias) or null
    const float* res;     // residual, same shape as out, or null
    const float* up;      // [B,Ho/2,Wo/2,Cout]: nearest-x2 upsampled and added (FPN top-down) or null
    float* gn_part;       // [B][P][Cout][2] per-32-row-tile column sums / sums of squares, or null
};

struct ConvArgs {
    ConvPtrs p[kMaxGroup];
    float* splitk_ws;     // [G][nsplit][B][mtiles*BM][Npad] raw partial sums (nsplit > 1)
    int* tickets;         // fused split-K: one arrival counter per (group, image, m tile, n tile), zero between launches
    int fused;            // nsplit > 1: 1 = the last workgroup to arrive at a tile sums the partials and applies the epilogue
                          // inside k_conv_igemm; 0 = k_conv_splitk_epilogue does (a second launch)
    int B, Hi, Wi, Cin, Ho, Wo, Cout, Npad, Kh, Kw, stride, pad, K, Kpad;
    long long in_sb, in_sh, in_sw, in_sc;   // element strides of the input
    int relu, nsplit, mtiles, ntiles, ksteps, bm, bn, generic, groups;
    int bf3;              // MODE 0 only: split-precision matrix products (three bf16 planes per operand, six MFMAs per tile)
    int lanepx;           // MODE 0 only: 1 = a K-step is 8 consecutive pixels of `Cin / 8` channels (one kernel ROW of the
                          // stem: Kw = 1, Cin = 32 virtual channels); horizontal padding is then per lane
    const float* wino_w[kMaxGroup];   // host side only: Winograd-packed weights per group (or null)
    const float* zeros;               // host side only: 64 zero floats for the all-DMA Winograd form (or null)
    void* dbg;                        // host side only: diagnostic stamp buffer for k_conv_wino (or null)
};

// FPN lateral 1x1 convolutions of all decoders as one pixel-resident product (lateral.hip): K = Cin in {64, 128}
struct LatArgs {
    const float* in;                        // [B][Ho*Wo][K] channel-last, contiguous; shared by the groups
    const unsigned short* wpl[kMaxGroup];   // k_pack_weight_bf3's planes: [3][Npad][Kpad] bf16
    float* out[kMaxGroup];                  // [B][Ho*Wo][Cout]
    const float* shift[kMaxGroup];          // conv bias or null
    const float* up[kMaxGroup];             // [B][Ho/2][Wo/2][Cout], nearest-x2 upsampled and added; all null or none
    int B, Ho, Wo, Cout, Npad, Kpad, groups, relu;
    unsigned bias_mask;                     // set by launch_lateral1x1: 0 = no bias (shift then points at readable memory)
    int parts;                              // workgroups per 128-pixel tile: each walks (groups * Cout / 32) / parts weight tiles
};
int launch_lateral1x1(const LatArgs& a, hipStream_t s);

// the 7x7 / stride-2 / pad-3 stem on the NHWC4 image as a weight-resident product (stem.hip)
struct StemArgs {
    const float* in;             // [B][Hi][Wi][4] f32 (4th channel 0), contiguous
    const unsigned short* wpl;   // k_pack_weight_bf3's planes [3][Npad][Kpad = 224], k = (kh * 8 + tap) * 4 + channel
    float* out;                  // [B][Ho][Wo][Cout = 64]
    const float* scale;          // folded BatchNorm, or null
    const float* shift;
    int B, Hi, Wi, Ho, Wo, Cout, Npad, Kpad, relu;
    int grid;                    // persistent workgroups (0: one per CU of an MI355X)
};
int launch_stem7x7(const StemArgs& a, hipStream_t s);

constexpr int kMaxGnSites = 4;    // segmentation sites finalized by one launch (x kMaxGroup decoders each)
struct GnFinArgs {
    const float* gn_part[kMaxGnSites * kMaxGroup];   // [site * kMaxGroup + decoder]: [B][P][C][2]
    const float* gamma[kMaxGnSites * kMaxGroup];
    const float* beta[kMaxGnSites * kMaxGroup];
    float* affine[kMaxGnSites * kMaxGroup];          // [B][C][2]: y = x*a + b
    int P[kMaxGnSites];                // partial rows per image of the site's convolution plan
    long long count[kMaxGnSites];      // elements per (image, group) = Ho*Wo*C/groups
    int B, C, groups, sites;
    float eps;
};

constexpr int kMaxUpJobs = 2;     // upsample jobs (different map sizes) of one launch (x kMaxGroup decoders each)
struct GnUpArgs {
    const float* in[kMaxUpJobs * kMaxGroup];        // [job * kMaxGroup + decoder]: [B,h,w,C]
    const float* affine[kMaxUpJobs * kMaxGroup];    // [B][C][2]
    float* out[kMaxUpJobs * kMaxGroup];             // [B,2h,2w,C]
    int h[kMaxUpJobs], w[kMaxUpJobs];
    int B, C, jobs;
};

struct MergeHeadArgs {
    const float* t_lo[kMaxGroup][3];   // three [B,h,w,C] pre-GroupNorm maps, upsampled x2 after GN+ReLU
    const float* a_lo[kMaxGroup][3];   // their affines [B][C][2]
    const float* t_hi[kMaxGroup];      // [B,2h,2w,C] pre-GroupNorm map at the merge resolution
    const float* a_hi[kMaxGroup];
    const float* hw[kMaxGroup];        // head weight [Ch][C]
    const float* hb[kMaxGroup];        // head bias [Ch]
    float* out[kMaxGroup];             // low-res logits NHWC [B,2h,2w,chp]
    int ch[kMaxGroup];                 // real head channels
    int chp[kMaxGroup];                // padded (multiple of 4) channel stride of out
    int B, h, w, C;
};

// the merge + head in two passes (merge_split.hip): LOW sums the three upsampled branches at their own resolution and applies the
// head; HI applies it to the p2 branch and adds bias + the x2 upsample of LOW's result
struct HeadPartArgs {
    const float* t[kMaxGroup][3];      // pre-GroupNorm maps [B,H,W,C] of this pass (LOW: three, HI: one)
    const float* aff[kMaxGroup][3];    // their affines [B][C][2]
    const float* hw[kMaxGroup];        // head weight [Ch][C]
    const float* hb[kMaxGroup];        // head bias [Ch] (HI)
    const float* lsum[kMaxGroup];      // HI: pass LOW's output [B,hl,wl,chp]
    float* out[kMaxGroup];             // LOW: [B,H,W,chp] without bias; HI: the low-resolution logits NHWC [B,H,W,chp]
    int ch[kMaxGroup], chp[kMaxGroup];
    int B, H, W, C, hl, wl;
};
int launch_head_part(const HeadPartArgs& a, bool hi, int groups, hipStream_t s);

struct Up4Args {
    const float* lm; const float* lq; const float* lt; const float* ls;   // low-res NHWC logits (mask, quat, xyz, scales)
    int pm, pq, pt, ps;                                                   // their channel strides
    float* o_mask; float* o_quat; float* o_scales; float* o_xy; float* o_z;   // full-res NCHW logits (nullable as a set)
    long long* cat_mask; float* cq; float* cs; float* cxy; float* cz;          // categorical outputs
    int B, hl, wl, H, W, C;                                               // C classes incl. background
    unsigned long long* fg_bits; size_t fg_stride;                        // nullable: foreground bit words [B][fg_stride] (W % 64 == 0)
};

// Winograd F(2x2,3x3) convolution (3x3, stride 1, pad 1, NHWC, Cin % 8 == 0, Cout % 64 == 0)
struct WinoArgs {
    ConvPtrs p[kMaxGroup];   // .w = Winograd-packed weights [Cout/64][Cin/8][16][64][8]; .up unused
    long long* dbg;          // diagnostic builds: per (workgroup, wave) cycle sums of the K-loop phases, or null
    const float* zeros;      // >= 16 bytes of zeros, 16-byte aligned (source of out-of-image DMA pieces; variant 2)
    int variant;             // 0: barrier form, 1: wave-private barrier-free K loop (4 waves), 2: all-DMA 3-stage (8 waves),
                             // 3: split-precision barrier form (8 waves; .w = the k_wino_pack_bf3 image)
    int groups;
    int waves;               // 4: 8x4 tile patch per workgroup; 8: 8x8 patch (512 threads)
    int B, H, W, Cin, Cout, relu, tbx, tby;   // tbx = ceil(ceil(W/2)/8), tby = ceil(ceil(H/2)/waves) tile patches
};
int launch_conv_wino(const WinoArgs& a, int groups, hipStream_t s);
const float* zero_page();      // 64 zero floats in the code object (per device context), or null
int launch_wino_pack(const float* w_oihw, float* packed, int Cout, int Cin, hipStream_t s);
int launch_wino_pack_bf3(const float* w_oihw, float* packed, int Cout, int Cin, hipStream_t s);
// wino128.hip: the split-precision form with 128 output channels per workgroup (8 x 4 tile patch, 4 waves; Cout % 128 == 0;
// .w = the k_wino_pack_c128 fragment-order image, .waves = 4, tby = ceil(ceil(H/2)/4))
int launch_conv_wino_c128(const WinoArgs& a, int groups, hipStream_t s);
int launch_wino_pack_c128(const float* w_oihw, float* packed, int Cout, int Cin, hipStream_t s);
// wino_w4.hip: the split-precision form as four waves of 512 registers (8 x 8 tile patch x 64 channels, weights straight into the
// operand registers; .w = the k_wino_pack_bf3 image, tby = ceil(ceil(H/2)/8))
int launch_conv_wino_w4(const WinoArgs& a, int groups, hipStream_t s);
// wino_h2.hip: the four-wave form on two fp16 pieces per operand (range-limited, see the file; .w = the k_wino_pack_h2 image
// incl. its two-float tail: 16 * Cout * Cin + 2 floats)
int launch_conv_wino_h2(const WinoArgs& a, int groups, hipStream_t s);
int launch_wino_pack_h2(const float* w_oihw, float* packed, int Cout, int Cin, hipStream_t s);
int launch_absmax_bits(const float* w, long long n, unsigned* out_bits, hipStream_t s);      // max |w| as the bits of a non-negative float (atomicMax into *out_bits)
// wino_h3.hip: k_conv_wino_h2 with three of the four piece products, over pairs of K-steps (Cin a multiple of 16); its own image:
// 16 * Cout * Cin + 2 floats
int launch_conv_wino_h3(const WinoArgs& a, int groups, hipStream_t s);
int launch_wino_pack_h3(const float* w_oihw, float* packed, int Cout, int Cin, hipStream_t s);
int launch_conv(const ConvArgs& a, int groups, hipStream_t s);
int launch_conv_splitk_epilogue(const ConvArgs& a, int groups, hipStream_t s);
int launch_maxpool3x3s2(const float* in, float* out, int B, int Hi, int Wi, int C, int Ho, int Wo, hipStream_t s);
int launch_gn_finalize(const GnFinArgs& a, int groups, hipStream_t s);
int launch_gn_relu_up2(const GnUpArgs& a, int groups, hipStream_t s);
int launch_merge_head(const MergeHeadArgs& a, int groups, hipStream_t s);
int launch_up4_compress(const Up4Args& a, hipStream_t s);
void launch_up4_compress7x4(const Up4Args& a, hipStream_t s);   // up4.hip; preconditions checked by launch_up4_compress

// bilinear source coordinate, align_corners=True, torch's arithmetic (UpSample.cuh):
// src = dst * (in-1)/(out-1) in f32; i0 = (int)src; l1 = src - i0; i1 = i0 + (i0 < in-1)
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_coord(int dst, int in, int out) {
    float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    float src = scale * (float)dst;
    Lerp L;
    L.i0 = (int)src;
    L.i1 = L.i0 + (L.i0 < in - 1 ? 1 : 0);
    L.l1 = src - (float)L.i0;
    L.l0 = 1.f - L.l1;
    return L;
}

// lerp_coord with the scale (in - 1) / (out - 1) computed by the caller once
__device__ __forceinline__ Lerp lerp_scaled(int dst, int in, float scale) {
    float src = scale * (float)dst;
    Lerp L;
    L.i0 = (int)src;
    L.i1 = L.i0 + (L.i0 < in - 1 ? 1 : 0);
    L.l1 = src - (float)L.i0;
    L.l0 = 1.f - L.l1;
    return L;
}

int launch_pack_weight(const float* w_oihw, float* packed, int Cout, int Cin, int Cinp, int Kh, int Kw, int Kwp, int Npad,
                       int Kpad, hipStream_t s);
// + the three bf16 planes of the same image behind it (packed + Npad * Kpad floats; 1.5 x Npad * Kpad floats more):
// the weight operand of the split-precision direct convolution.  conv_packed_floats() = the room both need.
int launch_pack_weight_bf3(const float* w_oihw, float* packed, int Cout, int Cin, int Cinp, int Kh, int Kw, int Kwp, int Npad,
                           int Kpad, hipStream_t s);
inline size_t conv_packed_floats(int Npad, int Kpad) { return ((size_t)Npad * Kpad * 5 / 2 + 63) / 64 * 64; }
int launch_nchw3_to_nhwc4(const float* x, float* out, int B, int HW, hipStream_t s);
int launch_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int C,
                   float* scale, float* shift, hipStream_t s);

}  // namespace fpc
