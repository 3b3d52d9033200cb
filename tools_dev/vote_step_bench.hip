// Inner-loop probe for k_vote_count (gfx950): cycles per (64-pixel tile, hypothesis) step per SIMD for candidate
// ways of turning the per-pair decision into per-hypothesis counts.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
//   A  4 FMA + v_cmp -> SGPR pair + s_bcnt1 + v_writelane per step                      (shipped in round 2, first cut)
//   B  as A, four counts packed into one SGPR (s_lshl/s_or), one v_writelane per 4 steps
//   C<T> per-lane counters: T tiles per lane, v_cmp -> vcc + v_addc_co per step, one transposed wave reduction per group
//   D  4 FMA only (floor)            E  4 FMA + v_cmp -> SGPR + s_bcnt1 + s_add (no writelane)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define STEP4(M, g) M(g) M((g) + 1) M((g) + 2) M((g) + 3)
#define STEP16(M, g) STEP4(M, g) STEP4(M, (g) + 4) STEP4(M, (g) + 8) STEP4(M, (g) + 12)
#define STEP64(M) STEP16(M, 0) STEP16(M, 16) STEP16(M, 32) STEP16(M, 48)

template <int KIND, int T>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ hx, const float* __restrict__ hy, int ngroups,
                                             int reps, int* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    float a_s[T], b_s[T], c_s[T], a_t[T], b_t[T], c_t[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float ph = 0.001f * (threadIdx.x + 256 * t + 7 * blockIdx.x);
        a_s[t] = __sinf(ph); b_s[t] = -__cosf(ph); c_s[t] = 3.0f * ph; a_t[t] = 0.045f * __cosf(ph); b_t[t] = 0.045f * __sinf(ph); c_t[t] = 0.2f + ph;
    }
    int total = 0;
    for (int r = 0; r < reps; ++r)
        for (int G = 0; G < ngroups; ++G) {
            const float* HX = static_cast<const float*>(__builtin_assume_aligned(hx + G * 64, 256));
            const float* HY = static_cast<const float*>(__builtin_assume_aligned(hy + G * 64, 256));
            if (KIND == 0) {
                int cntv = 0;
#define MA(g) { const float gx = HX[(g)], gy = HY[(g)]; const float ss = __builtin_fmaf(a_s[0], gx, __builtin_fmaf(b_s[0], gy, c_s[0])); \
                const float th = __builtin_fmaf(a_t[0], gx, __builtin_fmaf(b_t[0], gy, c_t[0])); \
                const int c = __popcll(__builtin_amdgcn_ballot_w64(fabsf(ss) <= th)); asm("v_writelane_b32 %0, %1, %2" : "+v"(cntv) : "s"(c), "n"(g)); }
                STEP64(MA)
#undef MA
                total += cntv;
            } else if (KIND == 1) {
                int cntv = 0;
#define MB1(g, sh) { const float gx = HX[(g)], gy = HY[(g)]; const float ss = __builtin_fmaf(a_s[0], gx, __builtin_fmaf(b_s[0], gy, c_s[0])); \
                const float th = __builtin_fmaf(a_t[0], gx, __builtin_fmaf(b_t[0], gy, c_t[0])); \
                pk |= __popcll(__builtin_amdgcn_ballot_w64(fabsf(ss) <= th)) << (sh); }
#define MB(g4) { int pk = 0; MB1(4 * (g4), 0) MB1(4 * (g4) + 1, 8) MB1(4 * (g4) + 2, 16) MB1(4 * (g4) + 3, 24) \
                asm("v_writelane_b32 %0, %1, %2" : "+v"(cntv) : "s"(pk), "n"(g4)); }
                STEP16(MB, 0)
#undef MB
#undef MB1
                total += cntv;
            } else if (KIND == 2) {
                int acc[64];
#pragma unroll
                for (int g = 0; g < 64; ++g) acc[g] = 0;
#define MC(g) { const float gx = HX[(g)], gy = HY[(g)]; _Pragma("unroll") for (int t = 0; t < T; ++t) { \
                const float ss = __builtin_fmaf(a_s[t], gx, __builtin_fmaf(b_s[t], gy, c_s[t])); \
                const float th = __builtin_fmaf(a_t[t], gx, __builtin_fmaf(b_t[t], gy, c_t[t])); acc[(g)] += (fabsf(ss) <= th) ? 1 : 0; } }
                STEP64(MC)
#undef MC
                // transposed wave reduction: 64 counters x 64 lanes -> lane g holds the total of hypothesis g
                int n = 64;
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    n >>= 1;
                    const bool hi = (lane & m) != 0;
#pragma unroll
                    for (int i = 0; i < 32; ++i) {
                        if (i < n) {
                            const int send = hi ? acc[i] : acc[i + n], keep = hi ? acc[i + n] : acc[i];
                            acc[i] = keep + __shfl_xor(send, m, 64);
                        }
                    }
                }
                total += acc[0];
            } else if (KIND == 5) {
                int cntv = 0;
#define MF1(g, sh) { const float gx = HX[(g)], gy = HY[(g)]; int c = 0; _Pragma("unroll") for (int t = 0; t < T; ++t) { \
                const float ss = __builtin_fmaf(a_s[t], gx, __builtin_fmaf(b_s[t], gy, c_s[t])); \
                const float th = __builtin_fmaf(a_t[t], gx, __builtin_fmaf(b_t[t], gy, c_t[t])); \
                c += __popcll(__builtin_amdgcn_ballot_w64(fabsf(ss) <= th)); } pk |= c << (sh); }
#define MF(g2) { int pk = 0; MF1(2 * (g2), 0) MF1(2 * (g2) + 1, 16) \
                asm("v_writelane_b32 %0, %1, %2" : "+v"(cntv) : "s"(pk), "n"(g2)); }
                STEP16(MF, 0) STEP16(MF, 16)
#undef MF
#undef MF1
                total += cntv;
            } else if (KIND == 6) {
                // sign-bit rows: per (tile, hypothesis) 4 FMA + bound FMA + 2 sub + 2 alignbit, operands via v_readlane
                const float hxv = HX[lane], hyv = HY[lane];
                unsigned ro[T], rs[T];
#pragma unroll
                for (int t = 0; t < T; ++t) { ro[t] = 0; rs[t] = 0; }
#define MG(g) { const float gx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hxv), (g))); \
                const float gy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hyv), (g))); \
                _Pragma("unroll") for (int t = 0; t < T; ++t) { \
                const float ss = fabsf(__builtin_fmaf(a_s[t], gx, __builtin_fmaf(b_s[t], gy, c_s[t]))); \
                const float u1 = __builtin_fmaf(a_t[t], gx, __builtin_fmaf(b_t[t], gy, c_t[t])); \
                const float d = u1 - ss, d2 = __builtin_fmaf(0.97f, u1, -0.001f) - ss; \
                ro[t] = __builtin_amdgcn_alignbit(ro[t], __builtin_bit_cast(unsigned, d), 31); \
                rs[t] = __builtin_amdgcn_alignbit(rs[t], __builtin_bit_cast(unsigned, d2), 31); } }
                STEP64(MG)
#undef MG
#pragma unroll
                for (int t = 0; t < T; ++t) total += __popc(ro[t]) + __popc(rs[t]);
            } else if (KIND == 7) {
                // the same with the operands from scalar loads
                unsigned ro[T], rs[T];
#pragma unroll
                for (int t = 0; t < T; ++t) { ro[t] = 0; rs[t] = 0; }
#define MH(g) { const float gx = HX[(g)], gy = HY[(g)]; \
                _Pragma("unroll") for (int t = 0; t < T; ++t) { \
                const float ss = fabsf(__builtin_fmaf(a_s[t], gx, __builtin_fmaf(b_s[t], gy, c_s[t]))); \
                const float u1 = __builtin_fmaf(a_t[t], gx, __builtin_fmaf(b_t[t], gy, c_t[t])); \
                const float d = u1 - ss, d2 = __builtin_fmaf(0.97f, u1, -0.001f) - ss; \
                ro[t] = __builtin_amdgcn_alignbit(ro[t], __builtin_bit_cast(unsigned, d), 31); \
                rs[t] = __builtin_amdgcn_alignbit(rs[t], __builtin_bit_cast(unsigned, d2), 31); } }
                STEP64(MH)
#undef MH
#pragma unroll
                for (int t = 0; t < T; ++t) total += __popc(ro[t]) + __popc(rs[t]);
            } else if (KIND == 3) {
                float f = 0.f;
#define MD(g) { const float gx = HX[(g)], gy = HY[(g)]; const float ss = __builtin_fmaf(a_s[0], gx, __builtin_fmaf(b_s[0], gy, c_s[0])); \
                const float th = __builtin_fmaf(a_t[0], gx, __builtin_fmaf(b_t[0], gy, c_t[0])); asm volatile("" :: "v"(ss), "v"(th)); }
                STEP64(MD)
#undef MD
                total += (int)f;
            } else {
                int sc = 0;
#define ME(g) { const float gx = HX[(g)], gy = HY[(g)]; const float ss = __builtin_fmaf(a_s[0], gx, __builtin_fmaf(b_s[0], gy, c_s[0])); \
                const float th = __builtin_fmaf(a_t[0], gx, __builtin_fmaf(b_t[0], gy, c_t[0])); sc += __popcll(__builtin_amdgcn_ballot_w64(fabsf(ss) <= th)); }
                STEP64(ME)
#undef ME
                total += sc;
            }
        }
    out[blockIdx.x * 256 + threadIdx.x] = total;
}

template <int KIND, int T> void run(const char* name, int wg_per_cu, const float* hx, const float* hy, int* out) {
    const int ngroups = 16, reps = 8, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) probe<KIND, T><<<grid, 256>>>(hx, hy, ngroups, reps, out);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) probe<KIND, T><<<grid, 256>>>(hx, hy, ngroups, reps, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double steps_per_simd = (double)wg_per_cu * reps * ngroups * 64 * T;     // one wave of each WG per SIMD
    printf("%-44s waves/SIMD=%d  %.3f ms  %.2f cycles per (tile,hypothesis) step per SIMD\n", name, wg_per_cu, ms,
           ms * 1e-3 * 2.4e9 / steps_per_simd);
}

int main() {
    float *hx, *hy; int* out;
    hipMalloc(&hx, 4096 * 4); hipMalloc(&hy, 4096 * 4); hipMalloc(&out, 256 * 8 * 256 * 4);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 100.f + (i * 37 % 400);
    hipMemcpy(hx, h, sizeof(h), hipMemcpyHostToDevice); hipMemcpy(hy, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w : {1, 2, 4, 5, 8}) {
        run<6, 2>("G2 sign-bit rows, 2 tiles, readlane operands (20 VALU / 2-tile step)", w, hx, hy, out);
        run<7, 2>("H2 sign-bit rows, 2 tiles, s_load operands (18 VALU / 2-tile step)", w, hx, hy, out);
        run<7, 4>("H4 sign-bit rows, 4 tiles, s_load operands", w, hx, hy, out);
        run<3, 1>("D  4 FMA only", w, hx, hy, out);
        run<4, 1>("E  4 FMA + cmp->sgpr + bcnt + s_add", w, hx, hy, out);
        run<0, 1>("A  ... + v_writelane per step", w, hx, hy, out);
        run<1, 1>("B  packed x4, v_writelane per 4 steps", w, hx, hy, out);
        run<5, 2>("F2 2 tiles/lane, cmp->sgpr, packed x2 writelane", w, hx, hy, out);
        run<5, 4>("F4 4 tiles/lane, cmp->sgpr, packed x2 writelane", w, hx, hy, out);
        if (w <= 5) { run<2, 2>("C2 per-lane counters, 2 tiles + wave transpose", w, hx, hy, out);
                      run<2, 4>("C4 per-lane counters, 4 tiles + wave transpose", w, hx, hy, out); }
    }
    return 0;
}
