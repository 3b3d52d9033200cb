// Which XCDs / CUs does a CU-masked stream run on?  hipcc --offload-arch=gfx950 -O2 tools_dev/cumask_probe.hip -o tools_dev/cumask_probe
// Launches 4096 short workgroups on a stream created with hipExtStreamCreateWithCUMask for a few masks and prints the
// histogram of XCC_ID (hwreg 20) and of (SE, CU) from HW_ID.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void k_probe(unsigned* out) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
    // a little work so that the blocks spread
    float x = threadIdx.x;
    for (int i = 0; i < 2000; ++i) x = x * 1.0001f + 0.5f;
    if (x == 12345.f) out[0] = 1;
}
static void run(const char* name, const std::vector<unsigned>& mask) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
    const int N = 4096;
    unsigned* d; hipMalloc(&d, N * 8);
    hipLaunchKernelGGL(k_probe, dim3(N), dim3(256), 0, s, d);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * N);
    hipMemcpy(h.data(), d, N * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> xc; std::map<unsigned, int> cus;
    for (int i = 0; i < N; ++i) { xc[h[2 * i] & 0xf]++; cus[((h[2 * i] & 0xf) << 16) | (h[2 * i + 1] & 0xffff00)]++; }
    printf("%s: XCDs used:", name);
    for (auto& kv : xc) printf(" %u:%d", kv.first, kv.second);
    printf("  distinct (xcd, se/cu) = %zu\n", cus.size());
    hipFree(d); hipStreamDestroy(s);
}
int main() {
    std::vector<unsigned> all(8, 0xffffffffu);
    run("all 256", all);
    for (int k = 0; k < 8; ++k) {            // bits [32k, 32k+32)
        std::vector<unsigned> m(8, 0); m[k] = 0xffffffffu;
        char nm[64]; snprintf(nm, 64, "word %d", k); run(nm, m);
    }
    {   // every 8th bit
        std::vector<unsigned> m(8, 0x01010101u); run("bits = 0 mod 8", m);
        std::vector<unsigned> m2(8, 0x02020202u); run("bits = 1 mod 8", m2);
    }
    return 0;
}
