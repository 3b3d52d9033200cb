// Split-precision probe for the ranked next step: what does a six-term bf16 x 3 product chain on
// v_mfma_f32_32x32x16_bf16 sustain, in f32-equivalent TFLOP/s, (a) with both operands already split, (b) when the
// A operand is split from f32 registers inside the loop (truncation split: and / sub, pack by v_perm)?
// Reference: the f32 matrix path (v_mfma_f32_32x32x2_f32) in the same harness.
//   hipcc --offload-arch=gfx950 -O3 -o tools_dev/bf16x3_probe tools_dev/bf16x3_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_hi(float a, float b) {      // {bf16(a), bf16(b)} by truncation: high halves
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
// split 8 f32 into three bf16x8 parts (truncation; exact: a = p1 + p2 + p3)
__device__ __forceinline__ void split8(const float* v, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
    float r[8], r2[8];
    u32x4 q1, q2, q3;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float h = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v[i]) & 0xFFFF0000u);
        r[i] = v[i] - h;
        float h2 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r[i]) & 0xFFFF0000u);
        r2[i] = r[i] - h2;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        q1[i] = pack_hi(v[2 * i], v[2 * i + 1]);
        q2[i] = pack_hi(r[2 * i], r[2 * i + 1]);
        q3[i] = pack_hi(r2[2 * i], r2[2 * i + 1]);
    }
    p1 = __builtin_bit_cast(bf16x8, q1); p2 = __builtin_bit_cast(bf16x8, q2); p3 = __builtin_bit_cast(bf16x8, q3);
}

template <int MODE>      // 0: f32 MFMA, 1: bf16x3 six terms pre-split, 2: bf16x3 with A split in the loop
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x16 acc0 = {0}, acc1 = {0};
    float av[8], bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { av[i] = seed * (threadIdx.x + i + 1) * 1e-3f; bv[i] = seed * (i + 3) * 1e-2f; }
    bf16x8 a1, a2, a3, b1, b2, b3;
    split8(av, a1, a2, a3); split8(bv, b1, b2, b3);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {         // 16 k of f32: 8 MFMAs of k = 2, two accumulators
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bv[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[e], av[e], acc1, 0, 0, 0);
            }
        } else {
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) av[i] += 1e-7f * it;      // new A values every step
                split8(av, a1, a2, a3);
            }
            // two independent 32x32 tiles (as a 64-wide wave tile would have), six terms each
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a2, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b3, a1, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a2, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a3, acc1, 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, int wg_per_cu) {
    const int iters = 4096, grid = 256 * wg_per_cu;
    float* out; (void)hipMalloc(&out, sizeof(float) * grid * 256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    // per iteration and wave: two 32x32 tiles x 16 k = 2 * 2*32*32*16 f32-equivalent FLOP
    double flop = (double)grid * 4 * iters * 2.0 * 2 * 32 * 32 * 16;
    printf("%-46s %d waves/SIMD: %8.3f ms -> %7.1f TFLOP/s f32-equivalent\n", name, wg_per_cu, ms, flop / ms / 1e9);
    (void)hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("f32 MFMA 32x32x2", w);
        run<1>("bf16 x 3, six terms, operands pre-split", w);
        run<2>("bf16 x 3, six terms, A split in the loop", w);
    }
    return 0;
}
