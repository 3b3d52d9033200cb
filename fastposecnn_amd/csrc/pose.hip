// pose.hip — gtf.batchwise_get_RT + quats_2_rotation_matrix
// (F/lib/gpu_tensor_funcs.py:204-235, 306-326): one lane per instance.
// The reference chains ~30 tiny torch launches including two batched LU inversions;
// RT = inverse([[inverse(R), T],[0 0 0 1]]) is [[R, -R T],[0 0 0 1]] for orthonormal R.
#include "common.hpp"

namespace fpc {

__global__ void k_pose_rt(const float* __restrict__ q, const float* __restrict__ xy, const float* __restrict__ z,
                          const float* __restrict__ kinv, int n, float* __restrict__ R, float* __restrict__ T,
                          float* __restrict__ RT) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float zz = z[i] / 1000.0f;
    float px = xy[2 * i] * zz, py = xy[2 * i + 1] * zz;
    float t[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) t[r] = kinv[3 * r] * px + kinv[3 * r + 1] * py + kinv[3 * r + 2] * zz;
    float q1 = q[4 * i], q2 = q[4 * i + 1], q3 = q[4 * i + 2], q4 = q[4 * i + 3];
    float nrm = sqrtf(q1 * q1 + q2 * q2 + q3 * q3 + q4 * q4);
    if (!(nrm > 0.0f)) nrm = 1.0f;
    q1 /= nrm; q2 /= nrm; q3 /= nrm; q4 /= nrm;
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
    for (int k = 0; k < 9; ++k) R[9 * (size_t)i + k] = Ri[k];
#pragma unroll
    for (int r = 0; r < 3; ++r) T[3 * (size_t)i + r] = t[r];
    float* G = RT + 16 * (size_t)i;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        G[4 * r] = Ri[3 * r]; G[4 * r + 1] = Ri[3 * r + 1]; G[4 * r + 2] = Ri[3 * r + 2];
        G[4 * r + 3] = -(Ri[3 * r] * t[0] + Ri[3 * r + 1] * t[1] + Ri[3 * r + 2] * t[2]);
    }
    G[12] = 0.0f; G[13] = 0.0f; G[14] = 0.0f; G[15] = 1.0f;
}

}  // namespace fpc

using namespace fpc;

extern "C" int fpc_pose_rt(const float* q, const float* xy, const float* z, const float* kinv, int n, float* R,
                           float* T, float* RT, fpc_stream_t stream) {
    if (n < 0) return FPC_EINVAL;
    if (n == 0) return FPC_OK;
    if (!q || !xy || !z || !kinv || !R || !T || !RT) return FPC_EINVAL;
    hipLaunchKernelGGL(k_pose_rt, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, q, xy, z, kinv, n, R, T, RT);
    return check_launch();
}
