// Issue cost and throughput of global -> LDS staging forms on one CU-filling launch (L2-resident source).
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, 1 KB per wave instruction)
//   mode 1: global_load_dwordx4 into registers, then ds_write_b128
//   mode 2: global_load_lds_dword (256 B per wave instruction)
// Per wave: R rounds of K loads (+ their wait).  Prints cycles per instruction for the issue phase alone and
// for issue + wait, at 1, 2, 4, 8 waves per CU.     hipcc --offload-arch=gfx950 -O3 -o tools_dev/dma_rate tools_dev/dma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 8, R = 64;
template <int MODE>
__global__ __launch_bounds__(512) void k(const float* g, long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) float lds[8 * K * 256];      // 8 KB per wave
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* src = g + ((size_t)(blockIdx.x * 8 + wv) % 512) * (K * 256) + lane * (MODE == 2 ? 1 : 4);
    float* dst = lds + wv * K * 256;
    long long t_issue = 0, t_all = 0;
    f32x4 r[K];
    for (int it = 0; it < R; ++it) {
        long long t0 = clock64();
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < K; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 256),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < K; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 64),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 64), 4, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < K; ++i) r[i] = *reinterpret_cast<const f32x4*>(src + i * 256);
        }
        long long t1 = clock64();
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < K; ++i) *reinterpret_cast<f32x4*>(dst + i * 256 + 4 * lane) = r[i];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        long long t2 = clock64();
        t_issue += t1 - t0; t_all += t2 - t0;
    }
    if (lane == 0) { out[(blockIdx.x * 8 + wv) * 2] = t_issue; out[(blockIdx.x * 8 + wv) * 2 + 1] = t_all; }
    if (sink && lane == 0) sink[blockIdx.x] = lds[wv * 7];
}
template <int MODE>
void run(const char* name, const float* g, long long* o, int waves) {
    int nblk = 256;
    long long h[256 * 8 * 2];
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(64 * waves), 0, 0, g, o, (float*)nullptr);
    hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(64 * waves), 0, 0, g, o, (float*)nullptr);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0, b = 0; int n = 0;
    for (int i = 0; i < nblk; ++i) for (int w = 0; w < waves; ++w) { a += h[(i * 8 + w) * 2]; b += h[(i * 8 + w) * 2 + 1]; ++n; }
    a /= (double)n * R * K; b /= (double)n * R * K;
    double bytes = (MODE == 2 ? 256.0 : 1024.0);
    printf("%-28s %d waves/CU: issue %7.1f clk/instr, issue+wait %7.1f clk/instr -> %6.1f B/clk/CU\n", name, waves, a, b,
           bytes * waves / b);
}
int main() {
    float* g; long long* o;
    (void)hipMalloc(&g, (size_t)512 * K * 256 * 4 + 4096); (void)hipMemset(g, 0, (size_t)512 * K * 256 * 4 + 4096);
    (void)hipMalloc(&o, 256 * 8 * 2 * 8);
    for (int w = 1; w <= 8; w *= 2) {
        run<0>("LDS-DMA dwordx4", g, o, w);
        run<1>("global_load x4 + ds_write", g, o, w);
        run<2>("LDS-DMA dword", g, o, w);
    }
    return 0;
}
