// Fills the LDS of every CU with a NaN / huge-integer pattern (two rounds of 64 KB workgroups, so both halves of the 160 KB
// are covered): a kernel launched afterwards that reads LDS it has not written shows it.  Test aid, not product code.
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void k_poison(unsigned pattern, int* sink) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 16 * 1024; i += 256) lds[i] = pattern;      // 64 KB
    __syncthreads();
    if (lds[(threadIdx.x * 61) & 16383] != pattern) *sink = 1;                  // keep the stores
}
extern "C" int poison_lds(unsigned pattern, int* sink, void* stream) {
    hipLaunchKernelGGL(k_poison, dim3(512), dim3(256), 64 * 1024, (hipStream_t)stream, pattern, sink);
    return (int)hipGetLastError();
}
