// Does the immediate offset of global_load_lds apply to the LDS address too?  ./tools_dev/glds_offset
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* g, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = -1.f;
    __syncthreads();
    const float* src = g + threadIdx.x * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds, 16, 1024, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
    float *g, *o; float h[4096], r[2048];
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    (void)hipMalloc(&g, sizeof(h)); (void)hipMalloc(&o, sizeof(r));
    (void)hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o);
    (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int first = -1; for (int i = 0; i < 2048; ++i) if (r[i] >= 0) { first = i; break; }
    printf("first written LDS float index %d holds global float %g (offset imm = 1024 bytes = 256 floats)\n", first, first >= 0 ? r[first] : -1.f);
    return 0;
}
