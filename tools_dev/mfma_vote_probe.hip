// Probe for the round-3 k_vote_count design (gfx950): the two affine forms of the cone filter on bf16x3 MFMA, the VALU only
// packing two bits per pair.  Measures (1) plain VALU issue rate vs waves per SIMD, (2) cycles per (64 entries x 32 hypotheses)
// step of the proposed loop, (3) the numerical error of the split-precision forms against fp64.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form tools_dev/mfma_vote_probe.hip -o tools_dev/mfma_vote_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned fu(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float uf(unsigned x) { return __uint_as_float(x); }

// f32 -> three bf16 pieces (as f32 values with 16 low zero bits), v == p1 + p2 + p3 exactly
__device__ __forceinline__ void split3(float v, float& p1, float& p2, float& p3) {
    p1 = uf(fu(v) & 0xffff0000u);
    const float r = v - p1;
    p2 = uf(fu(r) & 0xffff0000u);
    p3 = r - p2;
}
__device__ __forceinline__ unsigned pk(float lo, float hi) { return (fu(lo) >> 16) | (fu(hi) & 0xffff0000u); }

// A fragment halves of one form F = a X + b Y + c S + sg ES for one entry: lowK (slots 0-7), highK (slots 8-15)
__device__ __forceinline__ void a_frag(float a, float b, float c, float sg, u32x4& lo, u32x4& hi) {
    float a1, a2, a3, b1, b2, b3, c1, c2, c3;
    split3(a, a1, a2, a3); split3(b, b1, b2, b3); split3(c, c1, c2, c3);
    lo = u32x4{pk(a1, a1), pk(a2, a2), pk(a1, a3), pk(c1, c2)};
    hi = u32x4{pk(b1, b1), pk(b2, b2), pk(b1, b3), pk(c3, sg)};
}
// B fragment halves of one hypothesis: X, Y (already scaled), S = scale (a bf16 value), ES (a bf16 value)
__device__ __forceinline__ void b_frag(float X, float Y, float S, float ES, u32x4& lo, u32x4& hi) {
    float x1, x2, x3, y1, y2, y3;
    split3(X, x1, x2, x3); split3(Y, y1, y2, y3);
    lo = u32x4{pk(x1, x2), pk(x1, x2), pk(x3, x1), pk(S, S)};
    hi = u32x4{pk(y1, y2), pk(y1, y2), pk(y3, y1), pk(S, ES)};
}

// ---- (1) VALU rate ----------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void k_valu(float* out, int iters, float s0) {
    float a[8];
    unsigned row[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) a[i] = __builtin_fmaf(a[i], s0, a[(i + 3) & 7]);
            if (KIND == 1) { const float d = a[i] - fabsf(a[(i + 1) & 7]); row[i & 3] = __builtin_amdgcn_alignbit(row[i & 3], fu(d), 30); a[i] = d; }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(row[0] ^ row[1] ^ row[2] ^ row[3]);
}

// ---- (2) the proposed loop ---------------------------------------------------------------------------------------
// wave: 64 entries (2 MFMA row tiles) x NT hypothesis tiles of 32; B fragments streamed from global
template <bool WITH_VALU, bool WITH_MFMA>
__global__ __launch_bounds__(256) void k_loop(const u32x4* __restrict__ bfrag, int nt, int reps, const float4* __restrict__ ent,
                                              int* __restrict__ out, long long* __restrict__ cyc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float4 q = ent[(blockIdx.x * 4 + wv) * 64 + lane];
    // forms of this lane's entry: s = ey X - ex Y + cs ; u2 = k2 (ex X + ey Y + ct) - E
    const float inv = __builtin_amdgcn_rsqf(q.z * q.z + q.w * q.w);
    const float ex = q.z * inv, ey = q.w * inv;
    const float cs = -(q.x * ey - q.y * ex), ct = -(q.x * ex + q.y * ey);
    const float k2 = 0.0447f;
    u32x4 s_lo, s_hi, t_lo, t_hi;
    a_frag(ey, -ex, cs, 0.0f, s_lo, s_hi);
    a_frag(k2 * ex, k2 * ey, k2 * ct, -1.0f, t_lo, t_hi);
    // tile 0 = entries 0-31 (lowK from lanes 0-31, highK from lanes 0-31 moved up), tile 1 = entries 32-63
    u32x4 As[2], At[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        auto p = __builtin_amdgcn_permlane32_swap(s_lo[r], s_hi[r], false, false);
        As[0][r] = p[0]; As[1][r] = p[1];
        auto p2 = __builtin_amdgcn_permlane32_swap(t_lo[r], t_hi[r], false, false);
        At[0][r] = p2[0]; At[1][r] = p2[1];
    }
    int neg = 0, bandhits = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        u32x4 B = bfrag[lane];
        for (int t = 0; t < nt; ++t) {
            const u32x4 Bn = bfrag[((t + 1 < nt ? t + 1 : 0)) * 64 + lane];
            const bf16x8 b8 = __builtin_bit_cast(bf16x8, B);
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                f32x16 F1 = {0}, F3 = {0};
                if (WITH_MFMA) {
                    F1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, As[tl]), b8, F1, 0, 0, 0);
                    F3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, At[tl]), b8, F3, 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) { F1[i] = uf(As[tl][i & 3] ^ B[i & 3]) ; F3[i] = uf(At[tl][i & 3] + B[(i + 1) & 3]); }
                }
                if (WITH_VALU) {
                    unsigned row = 0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) row = __builtin_amdgcn_alignbit(row, fu(F3[i] - fabsf(F1[i])), 30);
                    neg += __popc(row & 0xAAAAAAAAu);
                    const unsigned bm = (row >> 1) & ~row & 0x55555555u;
                    if (__builtin_amdgcn_ballot_w64(bm != 0)) bandhits += __popc(bm);
                } else {
                    neg += (int)fu(F1[0] + F3[5] + F1[9] + F3[15]);
                }
            }
            B = Bn;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = neg + (bandhits << 20);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// ---- (3) numerics ------------------------------------------------------------------------------------------------
// one wave: 32 entries x 32 hypotheses, F = a X + b Y + c S + sg ES by MFMA; the host compares with fp64
__global__ void k_num(const float4* __restrict__ coef /* a, b, c, sg per entry */, const float4* __restrict__ hyp /* X, Y, S, ES */,
                      float* __restrict__ out /* [32 entries][32 hyps] */) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float4 c = coef[r], g = hyp[r];
    u32x4 alo, ahi, blo, bhi;
    a_frag(c.x, c.y, c.z, c.w, alo, ahi);
    b_frag(g.x, g.y, g.z, g.w, blo, bhi);
    const u32x4 A = h ? ahi : alo, B = h ? bhi : blo;
    f32x16 F = {0};
    F = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), F, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        out[row * 32 + r] = F[i];
    }
}

static float bf16_trunc(float x) { unsigned u; memcpy(&u, &x, 4); u &= 0xffff0000u; memcpy(&x, &u, 4); return x; }

int main() {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float* out; hipMalloc(&out, sizeof(float) * 256 * 8 * 256);
    // (1)
    for (int kind = 0; kind < 2; ++kind)
        for (int w : {1, 2, 4, 8}) {
            const int iters = 8192, grid = 256 * w;
            auto launch = [&]() { if (kind == 0) k_valu<0><<<grid, 256>>>(out, iters, 1.0001f); else k_valu<1><<<grid, 256>>>(out, iters, 1.0001f); };
            launch(); launch();
            hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            const double per_simd = (double)w * iters * (kind == 0 ? 8 : 16);
            printf("valu kind=%d (%s) waves/SIMD=%d: %.3f ms -> %.2f cyc per wave-instruction per SIMD @2.4GHz\n", kind,
                   kind == 0 ? "v_fma" : "v_sub+v_alignbit", w, ms, ms * 1e6 * 2.4 / per_simd);
        }
    // (2)
    {
        const int nt = 32, reps = 64;
        std::vector<unsigned> hb(nt * 64 * 4);
        for (auto& x : hb) x = (rand() & 0x7fff7fff) | 0x30003000;      // finite bf16 pairs
        std::vector<float> he(256 * 8 * 4 * 64 * 4);
        for (size_t i = 0; i < he.size(); i += 4) { he[i] = rand() % 640; he[i + 1] = rand() % 480; he[i + 2] = cosf(i); he[i + 3] = sinf(i); }
        u32x4* db; float4* de; int* dout; long long* dc;
        hipMalloc(&db, hb.size() * 4); hipMalloc(&de, he.size() * 4); hipMalloc(&dout, 256 * 8 * 256 * 4); hipMalloc(&dc, 256 * 8 * 8);
        hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice); hipMemcpy(de, he.data(), he.size() * 4, hipMemcpyHostToDevice);
        for (int variant = 0; variant < 3; ++variant)
            for (int w : {1, 2, 3, 4}) {
                const int grid = 256 * w;
                auto launch = [&]() {
                    if (variant == 0) k_loop<true, true><<<grid, 256>>>(db, nt, reps, de, dout, dc);
                    if (variant == 1) k_loop<false, true><<<grid, 256>>>(db, nt, reps, de, dout, dc);
                    if (variant == 2) k_loop<true, false><<<grid, 256>>>(db, nt, reps, de, dout, dc);
                };
                launch(); launch();
                hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
                std::vector<long long> hc(grid); hipMemcpy(hc.data(), dc, grid * 8, hipMemcpyDeviceToHost);
                double mc = 0; for (auto c : hc) mc += c; mc /= grid;
                const double steps_per_simd = (double)w * nt * reps;       // (64 entries x 32 hyps) steps per SIMD
                printf("loop %-10s waves/SIMD=%d: %.3f ms, %.0f cyc per step per SIMD @2.4GHz wall; in-kernel %.0f cyc per step per wave -> %.4f cyc/pair/SIMD\n",
                       variant == 0 ? "mfma+valu" : variant == 1 ? "mfma only" : "valu only", w, ms, ms * 1e6 * 2.4 / steps_per_simd,
                       mc / (nt * reps), ms * 1e6 * 2.4 / steps_per_simd / 2048);
            }
    }
    // (3)
    {
        double worst = 0, worst_rel = 0;
        float4 *dc4, *dh4; float* dres;
        hipMalloc(&dc4, 32 * 16); hipMalloc(&dh4, 32 * 16); hipMalloc(&dres, 32 * 32 * 4);
        srand(1);
        for (int trial = 0; trial < 2000; ++trial) {
            std::vector<float> c(128), g(128), res(1024);
            const float scale = bf16_trunc(powf(2.0f, (rand() % 20) - 10) * (1.0f + (rand() % 128) / 128.0f));
            for (int r = 0; r < 32; ++r) {
                const float ang = rand() * 1e-3f, px = (rand() % 400) - 200 + 0.0f, py = (rand() % 400) - 200 + 0.0f;
                const float ex = cosf(ang), ey = sinf(ang);
                const bool tform = trial & 1;
                const float k = tform ? 0.0447f : 1.0f;
                c[4 * r] = tform ? k * ex : ey; c[4 * r + 1] = tform ? k * ey : -ex;
                c[4 * r + 2] = tform ? -k * (px * ex + py * ey) : -(px * ey - py * ex); c[4 * r + 3] = tform ? -1.0f : 0.0f;
                const float gx = ((rand() % 200000) - 100000) * 0.004f, gy = ((rand() % 200000) - 100000) * 0.004f;
                g[4 * r] = scale * gx; g[4 * r + 1] = scale * gy; g[4 * r + 2] = scale; g[4 * r + 3] = bf16_trunc(scale * 3e-4f);
            }
            hipMemcpy(dc4, c.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dh4, g.data(), 512, hipMemcpyHostToDevice);
            k_num<<<1, 64>>>(dc4, dh4, dres);
            hipMemcpy(res.data(), dres, 4096, hipMemcpyDeviceToHost);
            for (int r = 0; r < 32; ++r)
                for (int h = 0; h < 32; ++h) {
                    const double a = c[4 * r], b = c[4 * r + 1], cc = c[4 * r + 2], sg = c[4 * r + 3];
                    const double X = g[4 * h], Y = g[4 * h + 1], S = g[4 * h + 2], ES = g[4 * h + 3];
                    const double exact = a * X + b * Y + cc * S + sg * ES;
                    const double mag = fabs(a * X) + fabs(b * Y) + fabs(cc * S) + fabs(sg * ES);
                    const double err = fabs((double)res[r * 32 + h] - exact);
                    if (mag > 0 && err / mag > worst_rel) worst_rel = err / mag;
                    if (err > worst) worst = err;
                }
        }
        printf("numerics: worst |MFMA - fp64| / sum|terms| = %.3e (f32 eps = 5.96e-8), worst abs %.3e\n", worst_rel, worst);
    }
    return 0;
}
