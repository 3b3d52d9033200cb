// f32 MFMA peak / clock calibration on the GPU box:  ./tools_dev/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x = a + threadIdx.x * 1e-3f, y = b - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, const char* tag) {
    float* out; hipMalloc(&out, sizeof(float) * blocks * 256);
    int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, 0.5f, 0.25f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
    double mfma_per_simd = (double)blocks / 256.0 * iters * 8 * NACC;
    printf("%-28s blocks=%5d nacc=%d  %8.3f ms  %7.1f TF   implied clock at 64 cyc/MFMA: %.2f GHz\n", tag, blocks, NACC, ms,
           flop / ms / 1e9, mfma_per_simd * 64.0 / (ms * 1e6));
    hipFree(out);
}
int main() {
    run<1>(256, "1 wave/SIMD, 1 chain");
    run<2>(256, "1 wave/SIMD, 2 acc");
    run<4>(256, "1 wave/SIMD, 4 acc");
    run<1>(512, "2 waves/SIMD, 1 chain");
    run<2>(512, "2 waves/SIMD, 2 acc");
    run<4>(1024, "4 waves/SIMD, 4 acc");
    return 0;
}
