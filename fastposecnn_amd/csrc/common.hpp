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

}  // namespace fpc
