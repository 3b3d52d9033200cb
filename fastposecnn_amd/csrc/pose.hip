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
    pose_rt_one((size_t)i, xy[2 * i], xy[2 * i + 1], q, z, kinv, R, T, RT);
}

// Pose records of one rank for the multi-GPU gather (fastposecnn_amd/parallel.py): out f32 [capacity + 1][40].
// Row 0 carries the instance count (int32 bits in column 0); row 1 + i =
// {sample_id + offset (i32 bits), class_id (i32 bits), quaternion 4, scales 3, xy 2, z 1, R 9, T 3, RT 16}; unused rows 0.
__global__ __launch_bounds__(64) void k_pack_pose_records(const int64_t* __restrict__ sample_ids, const int64_t* __restrict__ class_ids,
                                                          const float* __restrict__ q, const float* __restrict__ sc,
                                                          const float* __restrict__ xy, const float* __restrict__ z,
                                                          const float* __restrict__ R, const float* __restrict__ T,
                                                          const float* __restrict__ RT, int n, int sample_offset,
                                                          float* __restrict__ out) {
    const int row = blockIdx.x, c = threadIdx.x;
    if (c >= 40) return;
    float v = 0.0f;
    if (row == 0) {
        if (c == 0) v = __builtin_bit_cast(float, n);
    } else if (row - 1 < n) {
        const int i = row - 1;
        if (c == 0) v = __builtin_bit_cast(float, (int)sample_ids[i] + sample_offset);
        else if (c == 1) v = __builtin_bit_cast(float, (int)class_ids[i]);
        else if (c < 6) v = q[4 * i + c - 2];
        else if (c < 9) v = sc[3 * i + c - 6];
        else if (c < 11) v = xy[2 * i + c - 9];
        else if (c < 12) v = z[i];
        else if (c < 21) v = R[9 * i + c - 12];
        else if (c < 24) v = T[3 * i + c - 21];
        else v = RT[16 * i + c - 24];
    }
    out[(size_t)row * 40 + c] = v;
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

extern "C" int fpc_pack_pose_records(const int64_t* sample_ids, const int64_t* class_ids, const float* q, const float* scales,
                                     const float* xy, const float* z, const float* R, const float* T, const float* RT, int n,
                                     int sample_offset, int capacity, float* out, fpc_stream_t stream) {
    if (n < 0 || capacity < 0 || n > capacity || !out) return FPC_EINVAL;
    if (n > 0 && (!sample_ids || !class_ids || !q || !scales || !xy || !z || !R || !T || !RT)) return FPC_EINVAL;
    hipLaunchKernelGGL(k_pack_pose_records, dim3(capacity + 1), dim3(64), 0, (hipStream_t)stream, sample_ids, class_ids, q, scales,
                       xy, z, R, T, RT, n, sample_offset, out);
    return check_launch();
}
