// common.hpp — shared host/device helpers of libfpc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fpc.h"
#include "../../include/fpc_rng.h"

namespace fpc {

constexpr int kWave = 64;  // CDNA wavefront

void set_hip_error(hipError_t e);
void clear_hip_error();      // fpc_last_hip_error() describes the LAST failing call of this thread only

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error(e);
        return FPC_ELAUNCH;
    }
    clear_hip_error();
    return FPC_OK;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// (double)x < 1e-6 for a float x, without leaving fp32: the nearest float to 1e-6 is
// 9.99999997e-7 < 1e-6, so the double comparison of the reference
// (RV/src/ransac_voting_kernel.cu:42-43,121) holds exactly when x <= 1e-6f
// (tests/test_host_logic.py::test_eps_threshold_equivalence).
__device__ __forceinline__ bool below_eps(float x) { return x <= 1e-6f; }

// IEEE division as ONE instruction group.  hipcc expands `a / b` into v_div_scale / v_rcp / fma / v_div_fmas /
// v_div_fixup, and when it interleaves two independent divisions the second one's scale flag travels through an SGPR
// pair and returns as `s_mov_b64 vcc, s[..]` one or two instructions before its `v_div_fmas`.  On gfx950 that
// v_div_fmas was seen reading the PREVIOUS vcc (the first division's flag) in aligned 16-lane groups when MFMA-heavy
// waves shared the CU: k_vote_plan's hypothesis x came out wrong in about one frame in 10^4 under the streaming
// runtime (tools_dev/pipe_soak.py).  Inside this block each division takes vcc straight from its own
// v_div_scale, seven VALU instructions before the v_div_fmas, and nothing can be scheduled into it.  The sequence is
// the compiler's own (f32 denormals on, the HIP default), so quotients are the correctly rounded ones the oracle gets.
// fastposecnn_amd/isa_lint.py fails the build when any kernel still has an SALU write of vcc close before a
// v_div_fmas.
__device__ __forceinline__ float div_ieee(float a, float b) {
    float d, n, r, e, q, o;
    asm("v_div_scale_f32 %0, vcc, %7, %7, %6\n\t"
        "v_rcp_f32_e32 %2, %0\n\t"
        "v_div_scale_f32 %1, vcc, %6, %7, %6\n\t"
        "v_fma_f32 %3, -%0, %2, 1.0\n\t"
        "v_fmac_f32_e32 %2, %3, %2\n\t"
        "v_mul_f32_e32 %4, %1, %2\n\t"
        "v_fma_f32 %3, -%0, %4, %1\n\t"
        "v_fmac_f32_e32 %4, %3, %2\n\t"
        "v_fma_f32 %3, -%0, %4, %1\n\t"
        "v_div_fmas_f32 %3, %3, %2, %4\n\t"
        "v_div_fixup_f32 %5, %3, %7, %6"
        : "=&v"(d), "=&v"(n), "=&v"(r), "=&v"(e), "=&v"(q), "=&v"(o)
        : "v"(a), "v"(b)
        : "vcc");
    return o;
}

__device__ __forceinline__ double div_ieee(double a, double b) {
    double d, n, r, e, q, o;
    asm("v_div_scale_f64 %0, vcc, %7, %7, %6\n\t"
        "v_rcp_f64_e32 %2, %0\n\t"
        "v_div_scale_f64 %1, vcc, %6, %7, %6\n\t"
        "v_fma_f64 %3, -%0, %2, 1.0\n\t"
        "v_fmac_f64_e32 %2, %2, %3\n\t"
        "v_fma_f64 %3, -%0, %2, 1.0\n\t"
        "v_fmac_f64_e32 %2, %2, %3\n\t"
        "v_mul_f64 %4, %1, %2\n\t"
        "v_fma_f64 %3, -%0, %4, %1\n\t"
        "v_div_fmas_f64 %3, %3, %2, %4\n\t"
        "v_div_fixup_f64 %5, %3, %7, %6"
        : "=&v"(d), "=&v"(n), "=&v"(r), "=&v"(e), "=&v"(q), "=&v"(o)
        : "v"(a), "v"(b)
        : "vcc");
    return o;
}

// One (pixel, hypothesis) vote, RV/src/ransac_voting_kernel.cu:106-125, with the
// pixel's |n| passed in (it does not depend on the hypothesis).  Compiled with
// -ffp-contract=off; `/` and sqrtf are correctly rounded in hipcc's default mode, so
// this is bit-identical to oracle/fpc_oracle.c:fpco_pair_is_inlier.
__device__ __forceinline__ bool pair_is_inlier(float cx, float cy, float nx, float ny, float norm1,
                                               float hx, float hy, float thresh) {
    float dx = hx - cx;
    float dy = hy - cy;
    float norm2 = sqrtf(dx * dx + dy * dy);
    if (below_eps(norm1) || below_eps(norm2)) return false;
    float angle_dist = (dx * nx + dy * ny) / (norm1 * norm2);
    return angle_dist > thresh;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers (HIP's float4 struct copies can land in scratch)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Split-precision operands (opt-in, ConvArgs::bf3): x = p1 + p2 + p3 exactly, each part 8 significant bits (bf16 by
// truncation), four values -> three packed 8-byte groups.  The six products p_i q_j with i + j <= 4 on
// v_mfma_f32_32x32x16_bf16 (f32 accumulation) reproduce the f32 product chain to ~2^-24 relative
// (tools_dev/split_precision_check.py) at 2.2x the f32 matrix rate (tools_dev/bf16x3_probe.hip).
__device__ __forceinline__ unsigned pack_hi16(float lo, float hi) {        // {bf16(lo), bf16(hi)}: the two high halves
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo), 0x07060302u);
}
__device__ __forceinline__ void split_bf3(f32x4 v, u32x2& p1, u32x2& p2, u32x2& p3) {
    float r[4], q[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e];                     // (bit_cast of a vector ELEMENT expression misbehaved: go through a scalar)
        const unsigned xb = __builtin_bit_cast(unsigned, x) & 0xFFFF0000u;
        r[e] = x - __builtin_bit_cast(float, xb);
        const float y = r[e];
        const unsigned yb = __builtin_bit_cast(unsigned, y) & 0xFFFF0000u;
        q[e] = y - __builtin_bit_cast(float, yb);
    }
    p1 = u32x2{pack_hi16(v[0], v[1]), pack_hi16(v[2], v[3])};
    p2 = u32x2{pack_hi16(r[0], r[1]), pack_hi16(r[2], r[3])};
    p3 = u32x2{pack_hi16(q[0], q[1]), pack_hi16(q[2], q[3])};
}

__device__ __forceinline__ int wave_reduce_add(int v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    return v;  // valid in lane 0
}

__device__ __forceinline__ double wave_reduce_add(double v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    return v;
}

// Block-wide sum broadcast to every thread. `scratch` holds >= blockDim.x/64 ints.
__device__ __forceinline__ int block_sum_bcast(int v, int* scratch) {
    int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave, nw = (blockDim.x + kWave - 1) / kWave;
    int s = wave_reduce_add(v);
    __syncthreads();
    if (lane == 0) scratch[w] = s;
    __syncthreads();
    int t = 0;
    for (int i = 0; i < nw; ++i) t += scratch[i];
    return t;
}

// gtf.batchwise_get_RT + quats_2_rotation_matrix of ONE instance (F/lib/gpu_tensor_funcs.py:204-235, 306-326): k_pose_rt's body, shared
// with k_vote_final, which appends it to an instance's voted centre when the caller passes the pose operands (ransac.hip).
// RT = inverse([[inverse(R), T],[0 0 0 1]]) is [[R, -R T],[0 0 0 1]] for orthonormal R.
__device__ __forceinline__ void pose_rt_one(size_t i, float x, float y, const float* __restrict__ q, const float* __restrict__ z,
                                            const float* __restrict__ kinv, float* __restrict__ R, float* __restrict__ T,
                                            float* __restrict__ RT) {
    float zz = div_ieee(z[i], 1000.0f);
    float px = x * zz, py = y * zz;
    float t[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) t[r] = kinv[3 * r] * px + kinv[3 * r + 1] * py + kinv[3 * r + 2] * zz;
    float q1 = q[4 * i], q2 = q[4 * i + 1], q3 = q[4 * i + 2], q4 = q[4 * i + 3];
    float nrm = sqrtf(q1 * q1 + q2 * q2 + q3 * q3 + q4 * q4);
    if (!(nrm > 0.0f)) nrm = 1.0f;
    q1 = div_ieee(q1, nrm); q2 = div_ieee(q2, nrm); q3 = div_ieee(q3, nrm); q4 = div_ieee(q4, nrm);
    float a = q1 * q1, b = q2 * q2, c = q3 * q3, d = q4 * q4;
    // M as written at gpu_tensor_funcs.py:316-324; the function returns its transpose
    float M[9] = {a - b - c + d, 2 * (q1 * q2 + q3 * q4), 2 * (q1 * q3 - q2 * q4),
                  2 * (q1 * q2 - q3 * q4), -a + b - c + d, 2 * (q2 * q3 + q1 * q4),
                  2 * (q1 * q3 + q2 * q4), 2 * (q2 * q3 - q1 * q4), -a - b + c + d};
    float Ri[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) Ri[3 * r + cc] = M[3 * cc + r];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[9 * i + k] = Ri[k];
#pragma unroll
    for (int r = 0; r < 3; ++r) T[3 * i + r] = t[r];
    float* G = RT + 16 * i;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        G[4 * r] = Ri[3 * r]; G[4 * r + 1] = Ri[3 * r + 1]; G[4 * r + 2] = Ri[3 * r + 2];
        G[4 * r + 3] = -(Ri[3 * r] * t[0] + Ri[3 * r + 1] * t[1] + Ri[3 * r + 2] * t[2]);
    }
    G[12] = 0.0f; G[13] = 0.0f; G[14] = 0.0f; G[15] = 1.0f;
}

}  // namespace fpc
