// What does a staging instruction cost a wave whose SIMD partner issues MFMAs back to back?
// 8 waves per workgroup = 2 per SIMD: waves 0-3 run a pure v_mfma_f32_32x32x2_f32 loop, waves 4-7 issue K
// staging instructions per round and stamp the issue time.   Forms:
//   0: global_load_lds_dwordx4, 64-bit VGPR address        1: same, SGPR base + 32-bit VGPR offset (inline asm)
//   2: global_load_dwordx4 into registers (VGPR address)   3: ds_read_b128        4: v_pk_fma_f32
//   5: buffer_load_dwordx4 offen (descriptor + 32-bit VGPR offset + SGPR offset)   6: ds_write_b128
// hipcc --offload-arch=gfx950 -O3 -o tools_dev/dma_vs_mfma tools_dev/dma_vs_mfma.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 8, R = 200;
template <int MODE, bool WITH_MFMA>
__global__ __launch_bounds__(512) void k(const float* g, long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) float lds[8 * K * 256];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wv < 4) {
        if (!WITH_MFMA) return;
        f32x16 acc0 = {0}, acc1 = {0};
        float x = (float)lane, y = 1.f;
        for (int it = 0; it < R * 12; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, acc1, 0, 0, 0);
            }
        }
        if (sink) sink[threadIdx.x] = acc0[0] + acc1[3];
        return;
    }
    const float* src = g + ((size_t)(blockIdx.x * 4 + wv - 4) % 512) * (K * 256) + lane * 4;
    float* dst = lds + wv * K * 256;
    long long t_issue = 0;
    f32x4 r[K];
    f32x2 p = {1.f, 2.f}, q2 = {0.5f, 0.25f};
    const unsigned voff = lane * 16;
    const float* sbase = g + ((size_t)(blockIdx.x * 4 + wv - 4) % 512) * (K * 256);
    for (int it = 0; it < R; ++it) {
        long long t0 = clock64();
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < K; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 256),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
        } else if (MODE == 1) {
            unsigned m0v = (unsigned)(size_t)(__attribute__((address_space(3))) void*)dst;
            asm volatile("s_mov_b32 m0, %0\n s_nop 0\n"
                         "global_load_lds_dwordx4 %1, %2\n global_load_lds_dwordx4 %1, %2 offset:1024\n"
                         "global_load_lds_dwordx4 %1, %2 offset:2048\n global_load_lds_dwordx4 %1, %2 offset:3072\n"
                         :: "s"(m0v), "v"(voff), "s"(sbase) : "memory");
            asm volatile("s_mov_b32 m0, %0\n s_nop 0\n"
                         "global_load_lds_dwordx4 %1, %2\n global_load_lds_dwordx4 %1, %2 offset:1024\n"
                         "global_load_lds_dwordx4 %1, %2 offset:2048\n global_load_lds_dwordx4 %1, %2 offset:3072\n"
                         :: "s"(m0v + 4096), "v"(voff), "s"(sbase + 1024) : "memory");
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < K; ++i) r[i] = *reinterpret_cast<const f32x4*>(src + i * 256);
        } else if (MODE == 5) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)sbase, 0, K * 1024, 0x00020000);
#pragma unroll
            for (int i = 0; i < K; ++i) r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, i * 1024, 0));
        } else if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < K; ++i) *reinterpret_cast<f32x4*>(dst + i * 256 + 4 * lane) = f32x4{p[0], p[1], q2[0], q2[1]};
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < K; ++i) r[i] = *reinterpret_cast<const f32x4*>(dst + i * 256 + 4 * lane);
        } else {
#pragma unroll
            for (int i = 0; i < K; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(q2), "v"(q2));
        }
        long long t1 = clock64();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (MODE == 2 || MODE == 3 || MODE == 5) {
#pragma unroll
            for (int i = 0; i < K; ++i) asm volatile("" :: "v"(r[i]));
        }
        t_issue += t1 - t0;
    }
    if (lane == 0) out[blockIdx.x * 4 + wv - 4] = t_issue;
    if (sink && lane == 0) sink[blockIdx.x] = lds[wv * 7] + p[0];
}
__global__ void oob(const float* d, float* out) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)d, 0, 64, 0x00020000);
    if (threadIdx.x == 0) {
        out[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 0u, 0, 0))[0];
        out[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 0x80000000u, 0, 0))[0];
        out[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 0u, 128, 0))[0];
        out[3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 48u, 0, 0))[0];
    }
}
template <int MODE, bool W>
void run(const char* name, const float* g, long long* o) {
    const int nblk = 256;
    long long h[nblk * 4];
    hipLaunchKernelGGL((k<MODE, W>), dim3(nblk), dim3(512), 0, 0, g, o, (float*)nullptr);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0; for (int i = 0; i < nblk * 4; ++i) a += h[i];
    printf("%-44s partner %s: %7.1f clk per instruction (issue)\n", name, W ? "issues MFMAs" : "absent      ", a / (nblk * 4.0 * R * K));
}
int main() {
    float* g; long long* o;
    (void)hipMalloc(&g, (size_t)512 * K * 256 * 4 + 8192); (void)hipMemset(g, 0, (size_t)512 * K * 256 * 4 + 8192);
    (void)hipMalloc(&o, 256 * 4 * 8);
    run<0, false>("LDS-DMA x4, VGPR address", g, o);   run<0, true>("LDS-DMA x4, VGPR address", g, o);
    run<1, false>("LDS-DMA x4, SGPR base + VGPR offset", g, o); run<1, true>("LDS-DMA x4, SGPR base + VGPR offset", g, o);
    run<2, false>("global_load_dwordx4 to registers", g, o); run<2, true>("global_load_dwordx4 to registers", g, o);
    run<3, false>("ds_read_b128", g, o); run<3, true>("ds_read_b128", g, o);
    run<4, false>("v_pk_fma_f32", g, o); run<4, true>("v_pk_fma_f32", g, o);
    run<5, false>("buffer_load_dwordx4 offen", g, o); run<5, true>("buffer_load_dwordx4 offen", g, o);
    run<6, false>("ds_write_b128", g, o); run<6, true>("ds_write_b128", g, o);
    // out-of-range semantics of a raw buffer load
    {
        float h[64]; for (int i = 0; i < 64; ++i) h[i] = 1.f + i;
        float* d; float* r2; (void)hipMalloc(&d, 4096); (void)hipMalloc(&r2, 4096);
        (void)hipMemset(d, 0, 4096); (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(oob, dim3(1), dim3(64), 0, 0, d, r2);
        float o4[4]; (void)hipMemcpy(o4, r2, sizeof(o4), hipMemcpyDeviceToHost);
        printf("raw buffer, num_records = 64 bytes: voffset 0 -> %g; voffset 0x80000000 -> %g; voffset 0 + soffset 128 -> %g (in allocation, past num_records); voffset 48 + 16-byte access -> %g\n",
               o4[0], o4[1], o4[2], o4[3]);
    }
    return 0;
}
