// class_compress.hip — Model.class_compression + gtf.class_compress
// (F/lib/pose_regressor.py:445-457, F/lib/gpu_tensor_funcs.py:37-99) in one pass:
// per pixel arg-max of log-softmax over the C mask logits, gather of the winning
// class's 4+3+2+1 regression channels (background -> 0), L2 normalisation of the
// quaternion and of the xy vote.  The reference materialises a one-hot [B,C,H,W]
// mask and [B,C-1,A,H,W] fp64 temporaries per head; here every input plane that is
// touched is read once and the 10 output planes + the i64 mask are written once.
// HBM-bound: lanes own consecutive pixels, so every plane access is a coalesced
// 256-byte wave row.
#include "common.hpp"

namespace fpc {

template <int MAXC>
__global__ __launch_bounds__(256) void k_class_compress(
    const float* __restrict__ ml, const float* __restrict__ quat, const float* __restrict__ scales,
    const float* __restrict__ xy, const float* __restrict__ z, const int64_t* __restrict__ cm_in, int C, int HW,
    int64_t* __restrict__ cat_mask, float* __restrict__ oq, float* __restrict__ os, float* __restrict__ oxy,
    float* __restrict__ oz, unsigned long long* __restrict__ fg_bits, size_t fg_stride) {
    int b = blockIdx.y;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
        int cls;
        if (cm_in) {
            long long c = cm_in[(size_t)b * HW + p];
            cls = (c < 0 || c >= C) ? 0 : (int)c;
        } else {
            const float* m = ml + (size_t)b * C * HW + p;
            float v[MAXC];
            float mx = m[0];
            v[0] = mx;
#pragma unroll
            for (int c = 1; c < MAXC; ++c)
                if (c < C) { v[c] = m[(size_t)c * HW]; mx = fmaxf(mx, v[c]); }
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) s += expf(v[c] - mx);
            float lse = logf(s);
            float best = (v[0] - mx) - lse;
            cls = 0;
#pragma unroll
            for (int c = 1; c < MAXC; ++c)
                if (c < C) {
                    float val = (v[c] - mx) - lse;
                    if (val > best) { best = val; cls = c; }
                }
        }
        cat_mask[(size_t)b * HW + p] = cls;
        if (fg_bits) {      // a wave's pixels are 64 consecutive ones starting on a word boundary; lanes past HW are not in the loop
            const unsigned long long fgm = __ballot(cls != 0);
            if ((threadIdx.x & 63) == 0) fg_bits[(size_t)b * fg_stride + (p >> 6)] = fgm;
        }
        float q0 = 0, q1 = 0, q2 = 0, q3 = 0, s0 = 0, s1 = 0, s2 = 0, v0 = 0, v1 = 0, zz = 0;
        if (cls > 0) {
            int g = cls - 1, G = C - 1;
            const float* qp = quat + ((size_t)b * 4 * G + 4 * g) * HW + p;
            q0 = qp[0]; q1 = qp[(size_t)HW]; q2 = qp[(size_t)2 * HW]; q3 = qp[(size_t)3 * HW];
            const float* sp = scales + ((size_t)b * 3 * G + 3 * g) * HW + p;
            s0 = sp[0]; s1 = sp[(size_t)HW]; s2 = sp[(size_t)2 * HW];
            const float* vp = xy + ((size_t)b * 2 * G + 2 * g) * HW + p;
            v0 = vp[0]; v1 = vp[(size_t)HW];
            zz = z[((size_t)b * G + g) * HW + p];
        }
        float nq = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
        if (nq == 0.0f) nq = 1.0f;
        float nv = sqrtf(v0 * v0 + v1 * v1);
        if (nv == 0.0f) nv = 1.0f;
        float* o = oq + (size_t)b * 4 * HW + p;
        o[0] = q0 / nq; o[(size_t)HW] = q1 / nq; o[(size_t)2 * HW] = q2 / nq; o[(size_t)3 * HW] = q3 / nq;
        o = os + (size_t)b * 3 * HW + p;
        o[0] = s0; o[(size_t)HW] = s1; o[(size_t)2 * HW] = s2;
        o = oxy + (size_t)b * 2 * HW + p;
        o[0] = v0 / nv; o[(size_t)HW] = v1 / nv;
        oz[(size_t)b * HW + p] = zz;
    }
}

}  // namespace fpc

using namespace fpc;

extern "C" int fpc_class_compress(const float* mask_logits, const float* quat, const float* scales, const float* xy,
                                  const float* z, const int64_t* cat_mask_in, int B, int C, int HW,
                                  int64_t* cat_mask, float* oq, float* os, float* oxy, float* oz,
                                  fpc_stream_t stream) {
    return fpc_class_compress_bits(mask_logits, quat, scales, xy, z, cat_mask_in, B, C, HW, cat_mask, oq, os, oxy, oz, nullptr, stream);
}

extern "C" int fpc_class_compress_bits(const float* mask_logits, const float* quat, const float* scales, const float* xy,
                                       const float* z, const int64_t* cat_mask_in, int B, int C, int HW,
                                       int64_t* cat_mask, float* oq, float* os, float* oxy, float* oz, uint64_t* fg_bits,
                                       fpc_stream_t stream) {
    if ((uintptr_t)fg_bits & 7) return FPC_EINVAL;
    unsigned long long* fb = reinterpret_cast<unsigned long long*>(fg_bits);
    const size_t fstride = (size_t)((HW + 4095) / 4096) * 64;
    if (B < 0 || C < 2 || C > 32 || HW < 1) return FPC_EINVAL;
    if (B == 0) return FPC_OK;
    if (B > 65535) return FPC_EINVAL;
    if ((!mask_logits && !cat_mask_in) || !quat || !scales || !xy || !z || !cat_mask || !oq || !os || !oxy || !oz)
        return FPC_EINVAL;
    int gx = cdiv(HW, 256);
    if (gx > 4096) gx = 4096;
    dim3 grid(gx, B), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (C <= 8)
        hipLaunchKernelGGL(k_class_compress<8>, grid, block, 0, s, mask_logits, quat, scales, xy, z, cat_mask_in, C,
                           HW, cat_mask, oq, os, oxy, oz, fb, fstride);
    else
        hipLaunchKernelGGL(k_class_compress<32>, grid, block, 0, s, mask_logits, quat, scales, xy, z, cat_mask_in, C,
                           HW, cat_mask, oq, os, oxy, oz, fb, fstride);
    return check_launch();
}
