// Do kernels on differently CU-masked streams run at the same time?  hipcc --offload-arch=gfx950 -O2 tools_dev/cumask_overlap.hip -o tools_dev/cumask_overlap
// Launches a ~1 ms spin kernel (one workgroup per CU of the share) on n masked streams and times the whole; also the same on
// n unmasked streams.  Concurrent: ~1 ms; serialised: ~n ms.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_spin(long long ticks, unsigned* out) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
static double run(int n, bool masked, int wgs, int reps, long long ticks = 100000LL) {
    std::vector<hipStream_t> st(n);
    for (int k = 0; k < n; ++k) {
        if (masked) {
            unsigned w[8];
            const int per = 8 / n;
            for (int i = 0; i < 8; ++i) w[i] = (i >= k * per && i < (k + 1) * per) ? 0xFFFFFFFFu : 0u;
            if (hipExtStreamCreateWithCUMask(&st[k], 8, w) != hipSuccess) { printf("mask create failed\n"); return -1; }
        } else if (hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking) != hipSuccess) return -1;
    }
    unsigned* d; hipMalloc(&d, 64);
    for (int k = 0; k < n; ++k) hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(256), 0, st[k], 1000LL, d);      // warm
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r)
        for (int k = 0; k < n; ++k) hipLaunchKernelGGL(k_spin, dim3(wgs), dim3(256), 0, st[k], ticks, d);   // 1 ms at 100 MHz by default
    hipDeviceSynchronize();
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (auto s : st) hipStreamDestroy(s);
    hipFree(d);
    return ms / reps;
}
int main() {
    for (int n : {2, 4}) {
        printf("%d streams, %3d workgroups each: masked %.2f ms per round, unmasked %.2f ms per round (1 ms kernels)\n", n, 256 / n,
               run(n, true, 256 / n, 5), run(n, false, 256 / n, 5));
        printf("%d streams, %3d workgroups each: masked %.2f ms per round, unmasked %.2f ms per round\n", n, 4 * 256 / n,
               run(n, true, 4 * 256 / n, 5), run(n, false, 4 * 256 / n, 5));
    }
    // oversubscribed: 4096 workgroups of 50 us per launch (a share holds 8 per CU): alone a launch takes 4096 / (8 x CUs of the share) rounds
    for (int n : {2, 4}) {
        printf("%d streams x 4096 workgroups of 50 us: masked %.2f ms per round (one stream alone on its share: %.2f ms), unmasked %.2f ms per round\n", n,
               run(n, true, 4096, 5, 5000LL), run(1, false, 4096 * 1, 5, 5000LL) * n, run(n, false, 4096, 5, 5000LL));
    }
    return 0;
}
