// eval.hip — the evaluation maths right after the matching, one launch for all matched pairs:
//   degree error     F/lib/gpu_tensor_funcs.py:411-476  get_quat_distance / get_raw_quat_distance / get_symmetric_quat_distance
//                    (+ quat_symmetric_tf :752-799, quaternion_multiply :717-750)
//   3-D IoU          :486-547  get_3d_ious -> get_asymmetric_3d_iou (+ get_3d_bbox :328-378, transform_3d_camera_coords_to_
//                    3d_world_coords :177-202)
//   offset error     :563-565  from_Ts_get_offset_error
// The reference runs ~40 small torch kernels per metric and a Python loop over the pairs for the IoU (a 4x4 torch.inverse
// and ~25 launches per pair).  One wave per pair here: the lanes share the 360 rotations of the symmetric distance; lanes
// 0-15 transform the sixteen box corners.  Arithmetic follows the reference's dtypes: the plain distance in f32, the
// symmetric one in f64 (its rotation table is f32 values widened), the boxes in f32-rounded outputs of f64 arithmetic.
// Reference quirks kept on purpose: the "angle" is the chord length |q0 -+ q1| passed through rad2deg, and
// get_asymmetric_3d_iou reduces the [3,8] corner matrix over dim 0, i.e. per corner over x/y/z, then multiplies 8 extents.
#include "common.hpp"

namespace fpc {

__device__ __forceinline__ bool inverse4(const float* m, double* inv) {
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (double)m[i];
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    const double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    if (det == 0.0) return false;
    const double r = div_ieee(1.0, det);
#pragma unroll
    for (int i = 0; i < 16; ++i) inv[i] *= r;
    return true;
}

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, kWave));
    return v;
}
// grid (n), block 64
__global__ __launch_bounds__(64) void k_pose_errors(const float* __restrict__ q0, const float* __restrict__ q1,
                                                    const int64_t* __restrict__ sym, const float* __restrict__ rot /* [nrot,4] */,
                                                    int nrot, const float* __restrict__ RT1, const float* __restrict__ RT2,
                                                    const float* __restrict__ sc1, const float* __restrict__ sc2,
                                                    const float* __restrict__ T1, const float* __restrict__ T2,
                                                    double* __restrict__ out_deg, float* __restrict__ out_iou,
                                                    float* __restrict__ out_off) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (out_deg) {
        const float a0 = q0[i * 4], a1 = q0[i * 4 + 1], a2 = q0[i * 4 + 2], a3 = q0[i * 4 + 3];
        const float b0 = q1[i * 4], b1 = q1[i * 4 + 1], b2 = q1[i * 4 + 2], b3 = q1[i * 4 + 3];
        if (!sym || sym[i] == 0) {
            // f32: min(|q0 - q1|, |q0 + q1|) -> rad2deg
            const float m0 = a0 - b0, m1 = a1 - b1, m2 = a2 - b2, m3 = a3 - b3;
            const float p0 = a0 + b0, p1 = a1 + b1, p2 = a2 + b2, p3 = a3 + b3;
            const float dm = sqrtf(m0 * m0 + m1 * m1 + m2 * m2 + m3 * m3), dp = sqrtf(p0 * p0 + p1 * p1 + p2 * p2 + p3 * p3);
            if (lane == 0) out_deg[i] = (double)(fminf(dm, dp) * 57.295779513082320876798154814105f);
        } else {
            // q1 (x) rot_k, normalised, against q0: f64
            double best = 1e300;
            for (int k = lane; k < nrot; k += kWave) {
                const double rw = (double)rot[k * 4], rx = (double)rot[k * 4 + 1], ry = (double)rot[k * 4 + 2], rz = (double)rot[k * 4 + 3];
                const double aw = (double)b0, ax = (double)b1, ay = (double)b2, az = (double)b3;
                double ow = aw * rw - ax * rx - ay * ry - az * rz;
                double ox = aw * rx + ax * rw + ay * rz - az * ry;
                double oy = aw * ry - ax * rz + ay * rw + az * rx;
                double oz = aw * rz + ax * ry - ay * rx + az * rw;
                double nn = sqrt(ow * ow + ox * ox + oy * oy + oz * oz);
                if (nn == 0.0) nn = 1.0;
                ow = div_ieee(ow, nn); ox = div_ieee(ox, nn); oy = div_ieee(oy, nn); oz = div_ieee(oz, nn);
                const double m0 = (double)a0 - ow, m1 = (double)a1 - ox, m2 = (double)a2 - oy, m3 = (double)a3 - oz;
                const double p0 = (double)a0 + ow, p1 = (double)a1 + ox, p2 = (double)a2 + oy, p3 = (double)a3 + oz;
                const double dm = sqrt(m0 * m0 + m1 * m1 + m2 * m2 + m3 * m3), dp = sqrt(p0 * p0 + p1 * p1 + p2 * p2 + p3 * p3);
                best = fmin(best, fmin(dm, dp) * 57.295779513082320876798154814105);
            }
            best = wave_min(best);
            if (lane == 0) out_deg[i] = best;
        }
    }
    if (out_off && lane == 0) {
        const float d0 = T1[i * 3] - T2[i * 3], d1 = T1[i * 3 + 1] - T2[i * 3 + 1], d2 = T1[i * 3 + 2] - T2[i * 3 + 2];
        out_off[i] = sqrtf(d0 * d0 + d1 * d1 + d2 * d2) * 10.0f;
    }
    if (out_iou) {
        // lanes 0-7: corners of box 1, lanes 8-15: box 2; per corner the max / min over its x, y, z (the reference's dim-0 reduce)
        const int box = (lane >> 3) & 1, c = lane & 7;
        const float* RT = (box ? RT2 : RT1) + (size_t)i * 16;
        const float* sc = (box ? sc2 : sc1) + (size_t)i * 3;
        double inv[16];
        const bool ok = inverse4(RT, inv);
        const double ux = (c & 2) ? -0.5 : 0.5, uy = (c & 4) ? -0.5 : 0.5, uz = (c & 1) ? -0.5 : 0.5;    // get_3d_bbox's row order
        const double px = ux * (double)sc[0], py = uy * (double)sc[1], pz = uz * (double)sc[2];
        const double wx = inv[0] * px + inv[1] * py + inv[2] * pz + inv[3];
        const double wy = inv[4] * px + inv[5] * py + inv[6] * pz + inv[7];
        const double wz = inv[8] * px + inv[9] * py + inv[10] * pz + inv[11];
        const double ww = inv[12] * px + inv[13] * py + inv[14] * pz + inv[15];
        const double cx = div_ieee(wx, ww), cy = div_ieee(wy, ww), cz = div_ieee(wz, ww);
        const double mx = ok ? fmax(cx, fmax(cy, cz)) : nan(""), mn = ok ? fmin(cx, fmin(cy, cz)) : nan("");
        // corner j of one box meets corner j of the other (lanes j and j + 8)
        const double omx = __shfl_xor(mx, 8, kWave), omn = __shfl_xor(mn, 8, kWave);
        const double ext = fmin(mx, omx) - fmax(mn, omn);
        double e = ext, ie = ext, v = mx - mn;
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) {
            e = fmin(e, __shfl_xor(e, o, kWave));
            ie *= __shfl_xor(ie, o, kWave);
            v *= __shfl_xor(v, o, kWave);
        }
        const double vol1 = __shfl(v, 0, kWave), vol2 = __shfl(v, 8, kWave);
        const double inter = (e < 0.0) ? 0.0 : ie;
        if (lane == 0) out_iou[i] = (float)div_ieee(inter, vol1 + vol2 - inter);
    }
}

}  // namespace fpc

using namespace fpc;

extern "C" int fpc_pose_errors(const float* q0, const float* q1, const int64_t* symmetric_ids, const float* rot, int nrot,
                               const float* RT1, const float* RT2, const float* scales1, const float* scales2,
                               const float* T1, const float* T2, int n, double* out_degree, float* out_iou3d,
                               float* out_offset, fpc_stream_t stream) {
    if (n < 0 || nrot < 0) return FPC_EINVAL;
    if (n == 0) return FPC_OK;
    if (out_degree && (!q0 || !q1 || (symmetric_ids && (!rot || nrot < 1)))) return FPC_EINVAL;
    if (out_iou3d && (!RT1 || !RT2 || !scales1 || !scales2)) return FPC_EINVAL;
    if (out_offset && (!T1 || !T2)) return FPC_EINVAL;
    hipLaunchKernelGGL(k_pose_errors, dim3(n), dim3(64), 0, (hipStream_t)stream, q0, q1, symmetric_ids, rot, nrot, RT1, RT2, scales1,
                       scales2, T1, T2, out_degree, out_iou3d, out_offset);
    return check_launch();
}
