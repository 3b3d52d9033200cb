// preprocess.hip — the input side of the path: NOCS colour frame (u8, H x W x 3) -> the
// network's f32 [3,H,W] tensor, as F/tools/dataset.py:249-262 does on the host in numpy:
//     image = preprocessing_fn(image)        smp.encoders.preprocess_input for the encoder's "imagenet" settings
//                                            [upstream, not in /root/reference]: float64; x / 255 when x.max() > 1 and the
//                                            input range is [0,1]; - mean; / std
//     image = image.transpose(2, 0, 1)       tools/transforms/general.py:7-8
//     image /= np.max(np.abs(image))         dataset.py:256-257 (still float64)
//     image = img_as_float32(image)          dataset.py:262: ONE rounding to f32
// Per channel the map u8 -> f64 is monotone, so max|.| follows from each channel's min and max byte:
//   k_pre_minmax   streams the frame once (16 bytes per lane), per-channel min / max by wave reduction + atomics
//   k_pre_apply    builds the 3 x 256 table  f32( ((x / 255) - mean) / std / m )  in fp64 per workgroup (LDS) and maps
//                  the frame through it: 4 pixels (12 bytes) in, three float4 plane stores out.  Bit-exact with numpy:
//                  the same IEEE double operations, one rounding.
// HBM: reads 2 x 3 HW bytes, writes 12 HW bytes per frame (the second read is served by L2 for a 0.9 MB frame).
#include <algorithm>

#include "common.hpp"

namespace fpc {

struct PreArgs { double mean[3], stdv[3]; int scale255; };   // scale255: input range [0,1] (smp's /255 rule applies)

// ws i32 [B][8]: [c] = min of channel c, [4 + c] = min of (255 - x) of channel c  (one 0xFF memset initialises both)
__global__ __launch_bounds__(256) void k_pre_minmax(const uint8_t* __restrict__ img, int HW, int B, uint32_t* __restrict__ ws) {
    const int b = blockIdx.y;
    const uint8_t* p = img + (size_t)b * HW * 3;
    unsigned mn[3] = {255u, 255u, 255u}, mx[3] = {0u, 0u, 0u};
    const int nvec = (HW * 3) / 48;                              // 48 bytes = 16 pixels: three aligned 16-byte loads
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += gridDim.x * blockDim.x) {
        const uint4* q = reinterpret_cast<const uint4*>(p + (size_t)v * 48);
        const uint4 a = q[0], c = q[1], d = q[2];
        const unsigned w[12] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            const unsigned x = (w[i >> 2] >> ((i & 3) * 8)) & 255u;
            mn[i % 3] = min(mn[i % 3], x); mx[i % 3] = max(mx[i % 3], x);
        }
    }
    if (blockIdx.x == 0)                                         // tail bytes (HW * 3 not a multiple of 48)
        for (int i = nvec * 48 + threadIdx.x; i < HW * 3; i += blockDim.x) {
            const unsigned x = p[i];
            mn[i % 3] = min(mn[i % 3], x); mx[i % 3] = max(mx[i % 3], x);
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            mn[c] = min(mn[c], (unsigned)__shfl_xor((int)mn[c], o, kWave));
            mx[c] = max(mx[c], (unsigned)__shfl_xor((int)mx[c], o, kWave));
        }
        if ((threadIdx.x & (kWave - 1)) == 0) {
            atomicMin(&ws[b * 8 + c], mn[c]);
            atomicMin(&ws[b * 8 + 4 + c], 255u - mx[c]);
        }
    }
}

__global__ __launch_bounds__(256) void k_pre_apply(const uint8_t* __restrict__ img, int HW, int B, PreArgs a,
                                                   const uint32_t* __restrict__ ws, float* __restrict__ out) {
    __shared__ float s_tab[3][256];
    __shared__ double s_m;
    const int b = blockIdx.y;
    // per-image: is x / 255 applied (smp: x.max() > 1), and the largest |value| m
    const unsigned mn[3] = {ws[b * 8 + 0], ws[b * 8 + 1], ws[b * 8 + 2]};
    const unsigned mx[3] = {255u - ws[b * 8 + 4], 255u - ws[b * 8 + 5], 255u - ws[b * 8 + 6]};
    const bool div255 = a.scale255 && max(mx[0], max(mx[1], mx[2])) > 1u;
    auto f64 = [&](int c, unsigned x) -> double {
        double v = (double)x;
        if (div255) v = div_ieee(v, 255.0);
        return div_ieee(v - a.mean[c], a.stdv[c]);
    };
    if (threadIdx.x == 0) {
        double m = 0.0;
        for (int c = 0; c < 3; ++c) m = fmax(m, fmax(fabs(f64(c, mn[c])), fabs(f64(c, mx[c]))));
        s_m = m;
    }
    __syncthreads();
    const double m = s_m;
    for (int i = threadIdx.x; i < 768; i += blockDim.x) s_tab[i >> 8][i & 255] = (float)div_ieee(f64(i >> 8, i & 255), m);
    __syncthreads();
    const uint8_t* p = img + (size_t)b * HW * 3;
    float* o = out + (size_t)b * 3 * HW;
    const int ngrp = HW / 4;                                     // 4 pixels = 12 bytes in, one float4 per plane out
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < ngrp; g += gridDim.x * blockDim.x) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p + (size_t)g * 12);
        const unsigned w0 = q[0], w1 = q[1], w2 = q[2];
        unsigned by[12];
#pragma unroll
        for (int i = 0; i < 4; ++i) { by[i] = (w0 >> (8 * i)) & 255u; by[4 + i] = (w1 >> (8 * i)) & 255u; by[8 + i] = (w2 >> (8 * i)) & 255u; }
#pragma unroll
        for (int c = 0; c < 3; ++c)
            *reinterpret_cast<float4*>(o + (size_t)c * HW + (size_t)g * 4) =
                make_float4(s_tab[c][by[c]], s_tab[c][by[3 + c]], s_tab[c][by[6 + c]], s_tab[c][by[9 + c]]);
    }
    if (blockIdx.x == 0)                                         // tail pixels (HW not a multiple of 4)
        for (int px = ngrp * 4 + threadIdx.x; px < HW; px += blockDim.x)
            for (int c = 0; c < 3; ++c) o[(size_t)c * HW + px] = s_tab[c][p[(size_t)px * 3 + c]];
}

}  // namespace fpc

using namespace fpc;

extern "C" size_t fpc_preprocess_workspace_bytes(int B) { return align_up((size_t)(B > 0 ? B : 1) * 8 * sizeof(uint32_t), 256); }

extern "C" int fpc_preprocess_u8(const uint8_t* img_hwc, int B, int H, int W, const double* mean3, const double* std3,
                                 int input_range_01, float* out_nchw, void* ws, size_t ws_bytes, fpc_stream_t stream) {
    if (B < 0 || H < 1 || W < 1 || (int64_t)H * W > (1 << 28)) return FPC_EINVAL;
    if (B == 0) return FPC_OK;
    if (!img_hwc || !mean3 || !std3 || !out_nchw || !ws) return FPC_EINVAL;
    if (((uintptr_t)ws & 255) != 0 || ws_bytes < fpc_preprocess_workspace_bytes(B)) return FPC_EWORKSPACE;
    if (B > 65535) return FPC_EINVAL;
    // 16-byte vector accesses: frames and planes must keep the alignment of their bases
    const int HW = H * W;
    if (((uintptr_t)img_hwc & 15) || ((uintptr_t)out_nchw & 15) || (B > 1 && ((HW * 3) % 16 || HW % 4))) return FPC_EINVAL;
    for (int c = 0; c < 3; ++c) if (!(std3[c] != 0.0)) return FPC_EINVAL;
    clear_hip_error();
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(ws, 0xFF, (size_t)B * 8 * sizeof(uint32_t), s);
    if (e != hipSuccess) { set_hip_error(e); return FPC_ELAUNCH; }
    PreArgs a;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; }
    a.scale255 = input_range_01 ? 1 : 0;
    const int gx = std::min(256, cdiv(HW * 3 / 48 + 1, 256));
    hipLaunchKernelGGL(k_pre_minmax, dim3(gx, B), dim3(256), 0, s, img_hwc, HW, B, (uint32_t*)ws);
    const int ga = std::min(512, cdiv(HW / 4 + 1, 256));
    hipLaunchKernelGGL(k_pre_apply, dim3(ga, B), dim3(256), 0, s, img_hwc, HW, B, a, (const uint32_t*)ws, out_nchw);
    return check_launch();
}
