// Scheduling probe for k_vote_count's tile loop (gfx950): one wave = 2 groups x 2 row tiles of 32 entries against NT hypothesis
// tiles; B fragments from LDS.  Variants of the VALU block (r = Ft - |Fs|, two bits per pair into a row):
//   0  as the kernel has it: one serial chain  row = alignbit(row, sub_i, 30)
//   1  two independent chains (even / odd registers), merged at the end
//   2  next tile's MFMAs issued before this tile's VALU block (two accumulator sets)
//   3  = 2 with two chains
//   4  one chain, 4 subs hoisted in front of their 4 alignbits
// Prints cycles per (32 x 32) tile step per SIMD at 1-4 workgroups per CU (launch_bounds(256, 4): <= 128 VGPRs).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form tools_dev/tile_loop_probe.hip -o tools_dev/tile_loop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ unsigned fu(float x) { return __float_as_uint(x); }

template <int V>
__device__ __forceinline__ unsigned classify(const f32x16& Fs, const f32x16& Ft) {
    if (V == 1 || V == 3) {
        unsigned ra = 0, rb = 0;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            ra = __builtin_amdgcn_alignbit(ra, fu(Ft[i] - fabsf(Fs[i])), 30);
            rb = __builtin_amdgcn_alignbit(rb, fu(Ft[i + 1] - fabsf(Fs[i + 1])), 30);
        }
        return ra | (rb << 16);      // fields in a different (fixed) order: fine for a count
    } else if (V == 4) {
        unsigned row = 0;
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            const float d0 = Ft[i] - fabsf(Fs[i]), d1 = Ft[i + 1] - fabsf(Fs[i + 1]), d2 = Ft[i + 2] - fabsf(Fs[i + 2]), d3 = Ft[i + 3] - fabsf(Fs[i + 3]);
            asm volatile("" ::: "memory");
            row = __builtin_amdgcn_alignbit(row, fu(d0), 30); row = __builtin_amdgcn_alignbit(row, fu(d1), 30);
            row = __builtin_amdgcn_alignbit(row, fu(d2), 30); row = __builtin_amdgcn_alignbit(row, fu(d3), 30);
        }
        return row;
    } else {
        unsigned row = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) row = __builtin_amdgcn_alignbit(row, fu(Ft[i] - fabsf(Fs[i])), 30);
        return row;
    }
}

template <int V>
__global__ __launch_bounds__(256, 4) void k_tiles(const u32x4* __restrict__ bfrag, int nt, int reps, const u32x4* __restrict__ afrag,
                                                  int* __restrict__ out, long long* __restrict__ cyc) {
    __shared__ u32x4 s_B[16 * 64];
    __shared__ int s_cnt[16 * 32];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < nt * 64; i += 256) s_B[i] = bfrag[i];
    for (int i = threadIdx.x; i < nt * 32; i += 256) s_cnt[i] = 0;
    u32x4 As[4], At[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { As[k] = afrag[((blockIdx.x * 4 + wv) * 8 + k) * 64 + lane]; At[k] = afrag[((blockIdx.x * 4 + wv) * 8 + 4 + k) * 64 + lane]; }
    __syncthreads();
    int und = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        u32x4 Bn = s_B[lane];
        if (V == 2 || V == 3) {
            f32x16 Fs[2], Ft[2];
            const f32x16 z = {0};
            bf16x8 B = __builtin_bit_cast(bf16x8, Bn);
            Fs[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, As[0]), B, z, 0, 0, 0);
            Ft[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, At[0]), B, z, 0, 0, 0);
            for (int t = 0; t < nt; ++t) {
                if (t + 1 < nt) Bn = s_B[(t + 1) * 64 + lane];
                unsigned r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int nb = (k + 1) & 1;
                    if (k < 3) {
                        Fs[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, As[k + 1]), B, z, 0, 0, 0);
                        Ft[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, At[k + 1]), B, z, 0, 0, 0);
                    } else {
                        B = __builtin_bit_cast(bf16x8, Bn);
                        Fs[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, As[0]), B, z, 0, 0, 0);
                        Ft[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, At[0]), B, z, 0, 0, 0);
                    }
                    r[k] = classify<V>(Fs[k & 1], Ft[k & 1]);
                }
                const int neg = __popc(r[0] & 0xAAAAAAAAu) + __popc(r[1] & 0xAAAAAAAAu) + __popc(r[2] & 0xAAAAAAAAu) + __popc(r[3] & 0xAAAAAAAAu);
                const unsigned w01 = ((r[0] >> 1) & ~r[0] & 0x55555555u) | (r[1] & ~(r[1] << 1) & 0xAAAAAAAAu);
                const unsigned w23 = ((r[2] >> 1) & ~r[2] & 0x55555555u) | (r[3] & ~(r[3] << 1) & 0xAAAAAAAAu);
                atomicAdd(&s_cnt[t * 32 + (lane & 31)], 64 - neg);
                if (__builtin_amdgcn_ballot_w64((w01 | w23) != 0u)) und += __popc(w01 | w23);
            }
        } else {
            for (int t = 0; t < nt; ++t) {
                const bf16x8 B = __builtin_bit_cast(bf16x8, Bn);
                if (t + 1 < nt) Bn = s_B[(t + 1) * 64 + lane];
                unsigned r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f32x16 Fs = {0}, Ft = {0};
                    Fs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, As[k]), B, Fs, 0, 0, 0);
                    Ft = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, At[k]), B, Ft, 0, 0, 0);
                    r[k] = classify<V>(Fs, Ft);
                }
                const int neg = __popc(r[0] & 0xAAAAAAAAu) + __popc(r[1] & 0xAAAAAAAAu) + __popc(r[2] & 0xAAAAAAAAu) + __popc(r[3] & 0xAAAAAAAAu);
                const unsigned w01 = ((r[0] >> 1) & ~r[0] & 0x55555555u) | (r[1] & ~(r[1] << 1) & 0xAAAAAAAAu);
                const unsigned w23 = ((r[2] >> 1) & ~r[2] & 0x55555555u) | (r[3] & ~(r[3] << 1) & 0xAAAAAAAAu);
                atomicAdd(&s_cnt[t * 32 + (lane & 31)], 64 - neg);
                if (__builtin_amdgcn_ballot_w64((w01 | w23) != 0u)) und += __popc(w01 | w23);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = und + s_cnt[threadIdx.x & 31];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nt = 16, reps = 64;
    std::vector<unsigned> hb(nt * 64 * 4), ha((size_t)1024 * 4 * 8 * 64 * 4);
    for (auto& x : hb) x = (rand() & 0x7fff7fff) | 0x30003000;
    for (auto& x : ha) x = (rand() & 0xffffffff & ~0x40004000) | 0x30003000;
    u32x4 *db, *da; int* dout; long long* dc;
    hipMalloc(&db, hb.size() * 4); hipMalloc(&da, ha.size() * 4); hipMalloc(&dout, 1024 * 256 * 4); hipMalloc(&dc, 1024 * 8);
    hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice); hipMemcpy(da, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    for (int v = 0; v < 5; ++v)
        for (int w : {1, 2, 4}) {
            const int grid = 256 * w;
            auto launch = [&]() {
                if (v == 0) k_tiles<0><<<grid, 256>>>(db, nt, reps, da, dout, dc);
                if (v == 1) k_tiles<1><<<grid, 256>>>(db, nt, reps, da, dout, dc);
                if (v == 2) k_tiles<2><<<grid, 256>>>(db, nt, reps, da, dout, dc);
                if (v == 3) k_tiles<3><<<grid, 256>>>(db, nt, reps, da, dout, dc);
                if (v == 4) k_tiles<4><<<grid, 256>>>(db, nt, reps, da, dout, dc);
            };
            launch(); launch();
            hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            std::vector<long long> hc(grid); hipMemcpy(hc.data(), dc, grid * 8, hipMemcpyDeviceToHost);
            double mc = 0; for (auto c : hc) mc += c; mc /= grid;
            const double steps = (double)nt * reps * 4;        // tile steps per wave
            printf("variant %d  WGs/CU=%d: %.3f ms; in-kernel %.1f cyc per tile step per wave = %.1f per SIMD (wall @2.4GHz: %.1f)\n", v, w, ms,
                   mc / steps, mc / steps / w * 1.0, ms * 1e6 * 2.4 / (steps * w));
        }
    return 0;
}
