// VALU issue-rate probe for gfx950: cycles per wave-instruction for a few op kinds.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float s0, float s1) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q0 = p0 + 1.f, q1 = p1 + 1.f, q2 = p2 + 1.f, q3 = p3 + 1.f;
    float2v ss = {s0, s0}, tt = {s1, s1};
    int cnt = 0;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {          // 8 independent v_fma_f32 (VGPR operands)
            a0 = __builtin_fmaf(a0, a1, a2); a1 = __builtin_fmaf(a1, a2, a3); a2 = __builtin_fmaf(a2, a3, a4); a3 = __builtin_fmaf(a3, a4, a5);
            a4 = __builtin_fmaf(a4, a5, a6); a5 = __builtin_fmaf(a5, a6, a7); a6 = __builtin_fmaf(a6, a7, a0); a7 = __builtin_fmaf(a7, a0, a1);
        } else if (KIND == 1) {   // 8 v_fma_f32 with one SGPR operand
            a0 = __builtin_fmaf(a0, s0, a2); a1 = __builtin_fmaf(a1, s1, a3); a2 = __builtin_fmaf(a2, s0, a4); a3 = __builtin_fmaf(a3, s1, a5);
            a4 = __builtin_fmaf(a4, s0, a6); a5 = __builtin_fmaf(a5, s1, a7); a6 = __builtin_fmaf(a6, s0, a0); a7 = __builtin_fmaf(a7, s1, a1);
        } else if (KIND == 2) {   // 8 v_pk_fma_f32
            p0 = __builtin_elementwise_fma(p0, q0, p1); p1 = __builtin_elementwise_fma(p1, q1, p2); p2 = __builtin_elementwise_fma(p2, q2, p3); p3 = __builtin_elementwise_fma(p3, q3, p0);
            q0 = __builtin_elementwise_fma(q0, p0, q1); q1 = __builtin_elementwise_fma(q1, p1, q2); q2 = __builtin_elementwise_fma(q2, p2, q3); q3 = __builtin_elementwise_fma(q3, p3, q0);
        } else if (KIND == 3) {   // 8 v_pk_fma_f32 with splat SGPR operand
            p0 = __builtin_elementwise_fma(p0, ss, p1); p1 = __builtin_elementwise_fma(p1, tt, p2); p2 = __builtin_elementwise_fma(p2, ss, p3); p3 = __builtin_elementwise_fma(p3, tt, p0);
            q0 = __builtin_elementwise_fma(q0, ss, q1); q1 = __builtin_elementwise_fma(q1, tt, q2); q2 = __builtin_elementwise_fma(q2, ss, q3); q3 = __builtin_elementwise_fma(q3, tt, q0);
        } else if (KIND == 4) {   // 4 fma + 4 (v_cmp -> sgpr, s_bcnt1)
            a0 = __builtin_fmaf(a0, s0, a2); a1 = __builtin_fmaf(a1, s1, a3); a2 = __builtin_fmaf(a2, s0, a4); a3 = __builtin_fmaf(a3, s1, a5);
            cnt += __popcll(__builtin_amdgcn_ballot_w64(a0 >= s0)); cnt += __popcll(__builtin_amdgcn_ballot_w64(a1 >= s0));
            cnt += __popcll(__builtin_amdgcn_ballot_w64(a2 >= s1)); cnt += __popcll(__builtin_amdgcn_ballot_w64(a3 >= s1));
        } else if (KIND == 6) {   // 4 fma + 4 v_cmp -> sgpr pairs, OR-combined (no s_bcnt1 / s_add chain)
            a0 = __builtin_fmaf(a0, s0, a2); a1 = __builtin_fmaf(a1, s1, a3); a2 = __builtin_fmaf(a2, s0, a4); a3 = __builtin_fmaf(a3, s1, a5);
            unsigned long long m = __builtin_amdgcn_ballot_w64(a0 >= s0) | __builtin_amdgcn_ballot_w64(a1 >= s0) |
                                   __builtin_amdgcn_ballot_w64(a2 >= s1) | __builtin_amdgcn_ballot_w64(a3 >= s1);
            cnt += (int)(m & 1);
        } else if (KIND == 7) {   // 4 fma + 4 (v_cmp -> vcc, v_addc per-lane counter)
            a0 = __builtin_fmaf(a0, s0, a2); a1 = __builtin_fmaf(a1, s1, a3); a2 = __builtin_fmaf(a2, s0, a4); a3 = __builtin_fmaf(a3, s1, a5);
            cnt += (a0 >= s0); cnt += (a1 >= s0); cnt += (a2 >= s1); cnt += (a3 >= s1);
        } else if (KIND == 8) {   // 4 fma + 4 x (v_sub, v_min |.|) : the band-distance accumulation
            a0 = __builtin_fmaf(a0, s0, a2); a1 = __builtin_fmaf(a1, s1, a3); a2 = __builtin_fmaf(a2, s0, a4); a3 = __builtin_fmaf(a3, s1, a5);
            a4 = __builtin_fminf(a4, __builtin_fabsf(a0 - s0)); a4 = __builtin_fminf(a4, __builtin_fabsf(a1 - s0));
            a5 = __builtin_fminf(a5, __builtin_fabsf(a2 - s1)); a5 = __builtin_fminf(a5, __builtin_fabsf(a3 - s1));
        } else if (KIND == 5) {   // 8 v_readlane
            int l = i & 63;
            cnt += __builtin_amdgcn_readlane(__builtin_bit_cast(int, a0), l) + __builtin_amdgcn_readlane(__builtin_bit_cast(int, a1), l)
                 + __builtin_amdgcn_readlane(__builtin_bit_cast(int, a2), l) + __builtin_amdgcn_readlane(__builtin_bit_cast(int, a3), l)
                 + __builtin_amdgcn_readlane(__builtin_bit_cast(int, a4), l) + __builtin_amdgcn_readlane(__builtin_bit_cast(int, a5), l)
                 + __builtin_amdgcn_readlane(__builtin_bit_cast(int, a6), l) + __builtin_amdgcn_readlane(__builtin_bit_cast(int, a7), l);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + q0.x + q1.y + q2.x + q3.y + cnt;
}
template <int KIND> void run(const char* name, int wgs_per_cu) {
    int iters = 4096, grid = 256 * wgs_per_cu;
    float* out; hipMalloc(&out, sizeof(float) * grid * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) probe<KIND><<<grid, 256>>>(out, iters, 1.0001f, 0.9999f);
    hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) probe<KIND><<<grid, 256>>>(out, iters, 1.0001f, 0.9999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    // wave-instructions per SIMD = wgs_per_cu (one wave of each WG per SIMD) * iters * 8
    double instr_per_simd = (double)wgs_per_cu * iters * 8;
    printf("%-34s wg/cu=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (x2.4GHz = %.2f cyc)\n", name, wgs_per_cu, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    hipFree(out);
}
int main() {
    for (int w : {4, 8}) {
        run<0>("v_fma_f32 vgpr", w); run<1>("v_fma_f32 sgpr operand", w); run<2>("v_pk_fma_f32", w);
        run<3>("v_pk_fma_f32 splat sgpr", w); run<4>("4 fma + 4 cmp->sgpr+bcnt (per 8)", w); run<5>("v_readlane x8", w);
        run<6>("4 fma + 4 cmp->sgpr, OR (per 8)", w); run<7>("4 fma + 4 cmp->vcc + addc (per 8)", w); run<8>("4 fma + 4 (sub + min|.|) (per 8: 12 instr)", w);
    }
    return 0;
}
