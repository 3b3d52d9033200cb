// How much f32-MFMA throughput does an interleaved VALU instruction cost on gfx950?  ./tools_dev/mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x = a + threadIdx.x * 1e-3f, y = b - threadIdx.x * 1e-3f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = x * (i + 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[u & 1], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; ++n) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(u + n) & 7]) : "v"(y));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV>
void run(int blocks) {
    float* out; (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(256), 0, 0, out, 10, 0.5f, 0.25f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double nm = (double)blocks / 256.0 * iters * 8;   // MFMAs per SIMD
    printf("waves/SIMD=%d  VALU per MFMA=%d  %8.3f ms  %6.1f TF  ns per MFMA slot %.1f\n", blocks / 256, NV, ms,
           (double)blocks * 4 * iters * 8 * 4096.0 / ms / 1e9, ms * 1e6 / nm);
    (void)hipFree(out);
}
int main() {
    run<0>(256); run<1>(256); run<2>(256); run<4>(256); run<8>(256);
    run<0>(512); run<1>(512); run<2>(512); run<4>(512); run<8>(512);
    return 0;
}
