// stem.hip — k_stem7x7: the encoder's 7x7 / stride-2 stem convolution (3 -> 64 channels, folded BatchNorm + ReLU; smp ResNetEncoder
// conv1 / bn1 / relu, F/lib/pose_regressor.py:709-743) as a weight-resident product for large batches.
//
// The implicit-GEMM kernel runs the stem as a 7x1 convolution over 8-pixel groups of the NHWC4 image (K = 7 rows x 8 taps x 4
// channels = 224, net.hip) on 128 x 64 tiles: every workgroup stages its own copy of the weight rows per K-step and splits its
// activation registers on the way into LDS — 650-810 us for a 32-frame batch whose matrix work is 170 us and whose output
// (630 MB) takes 115 us to write.  Here the WEIGHTS stay: a persistent workgroup of 8 waves copies the three bf16 planes
// k_pack_weight_bf3 already wrote ([plane][Npad][Kpad], k = (kh * 8 + tap) * 4 + channel) into LDS once, fragment-major
// ([k-group 14][plane 3][column tile 2][lane 64][16 bytes]: lane-linear, conflict-free ds_read_b128), and every wave walks
// 64-pixel segments of output rows: per kernel row and 4-tap group a lane fetches the two pixels of ITS output pixel's
// fragment straight from the image (2 x 16 bytes, neighbours overlap in L1), splits them into the three planes, and issues six
// v_mfma_f32_32x32x16_bf16 per 32-pixel half and 32-channel tile (the products p_i q_j with i + j <= 4: common.hpp) against B
// fragments read once per 64 pixels; the image loads run four k-groups ahead of their use.  Taps left / right of the image
// read as zero (a row outside the image likewise: predicated, not branched).  Epilogue: y = relu(acc * scale + shift), accumulator registers stored as they stand (one channel per lane, two
// full 128-byte lines per store instruction).  Bound: the matrix pipe (14 x 24 MFMAs per 64 pixels) beside ~2 300 vector
// instructions of loading and splitting per wave and tile.
#include <algorithm>

#include "net_kernels.hpp"

namespace fpc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kStemKG = 14;                       // 16-deep k-groups: 7 kernel rows x (taps 0-3 | taps 4-7)
constexpr int kStemLds = kStemKG * 3 * 2 * 1024;  // bytes

__global__ __launch_bounds__(512, 1) void k_stem7x7(const StemArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char s_w[kStemLds];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, h = lane >> 5;
    // ---- weight planes -> LDS, fragment-major: chunk (kg, plane, nt, lane) = column nt * 32 + (lane & 31), k = 16 kg + 8 (lane >> 5) .. + 7
    for (int c = tid; c < kStemKG * 3 * 2 * 64; c += 512) {
        const int l = c & 63, nt = (c >> 6) & 1, pk = c >> 7, plane = pk % 3, kg = pk / 3;
        const unsigned short* src = a.wpl + ((size_t)plane * a.Npad + nt * 32 + (l & 31)) * a.Kpad + 16 * kg + 8 * (l >> 5);
        *reinterpret_cast<u32x4*>(&s_w[(size_t)c * 16]) = *reinterpret_cast<const u32x4*>(src);
    }
    float sc[2], sh[2];                             // this lane's two output channels (column tiles 0 / 1)
    sc[0] = a.scale ? a.scale[col] : 1.f; sc[1] = a.scale ? a.scale[32 + col] : 1.f;
    sh[0] = a.shift ? a.shift[col] : 0.f; sh[1] = a.shift ? a.shift[32 + col] : 0.f;
    __syncthreads();

    const int segs = a.Wo >> 6;                     // 64-pixel segments per output row
    const int ntile = a.B * a.Ho * segs;            // < 2^31: launch_stem7x7
    // XCD-banded walk: workgroups go to the 8 XCDs round-robin; with the grid a multiple of 8, XCD x = blockIdx.x % 8 walks the
    // x-th contiguous eighth of the tiles (row segments in raster order), so the 3.5 output rows that share an input row are
    // served by one L2
    int t_first = blockIdx.x * 8 + wv, t_end = ntile, t_step = gridDim.x * 8;
    if ((gridDim.x & 7) == 0) {
        const int per = (ntile + 7) >> 3, lo = (blockIdx.x & 7) * per;
        t_first = lo + (blockIdx.x >> 3) * 8 + wv; t_end = min(lo + per, ntile); t_step = (gridDim.x >> 3) * 8;
    }
    for (int tile = t_first; tile < t_end; tile += t_step) {
        const int t2 = tile / segs, seg = tile - t2 * segs;
        const int b = t2 / a.Ho, oy = t2 - b * a.Ho;
        const int ox0 = seg * 64;
        const float* img = a.in + (size_t)b * a.Hi * a.Wi * 4;         // NHWC4 image (uniform)
        f32x16 acc[2][2];                                               // [32-pixel half][column tile]
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[s][nt][i] = 0.f;
        // input x of this lane's first tap of k-group half g2 and pixel half s: 2 (ox0 + 32 s + col) - 3 + 4 g2 + 2 h
        const int ixb = 2 * (ox0 + col) - 3 + 2 * h;
        // The 14 k-groups (kernel row j / 2, taps 4 (j % 2) ..) run fully unrolled with the image loads kPre groups ahead of their use
        // (a group's loads take 1-2 us to land, its 24 MFMAs 0.3 us, and two waves share a SIMD): a ring of kPre x 4 sixteen-byte
        // registers, slot j % kPre refilled right after group j's split.  A kernel row above / below the image is predicated off
        // (its loads return zero and its MFMAs add nothing: 1.5 % of the rows of a 240-row output) instead of branched around, so the
        // whole tile is one basic block; the scheduling barriers keep the compiler from hoisting all 56 loads to the top.
        constexpr int kPre = 4;
        f32x4 ring[kPre][4];
        unsigned keep[kPre][4];                             // all ones / zero: the tap lies inside / outside the image
        // (loads are unconditional from a clamped address and masked at their use: a load under a lane predicate becomes a branch,
        // and with branches between them the compiler's wait counts fall back to vmcnt(0))
        auto issue = [&](int j, f32x4 (&r)[4], unsigned (&m)[4]) {
            const int kh = j >> 1, g2 = j & 1;
            const int iy = 2 * oy - 3 + kh;
            const bool rowok = iy >= 0 && iy < a.Hi;
            const float* rowp = img + (size_t)min(max(iy, 0), a.Hi - 1) * a.Wi * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ix = ixb + 64 * (q >> 1) + 4 * g2 + (q & 1);
                m[q] = (rowok && ix >= 0 && ix < a.Wi) ? 0xFFFFFFFFu : 0u;
                r[q] = *reinterpret_cast<const f32x4*>(rowp + (size_t)min(max(ix, 0), a.Wi - 1) * 4);
            }
        };
        auto masked = [](f32x4 v, unsigned m) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float x = v[e]; o[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & m); }
            return o;
        };
#pragma unroll
        for (int j = 0; j < kPre; ++j) issue(j, ring[j], keep[j]);
#pragma unroll
        for (int j = 0; j < kStemKG; ++j) {
            u32x4 A1[2], A2[2], A3[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                u32x2 p1, p2, p3, q1, q2, q3;
                split_bf3(masked(ring[j % kPre][2 * s], keep[j % kPre][2 * s]), p1, p2, p3);
                split_bf3(masked(ring[j % kPre][2 * s + 1], keep[j % kPre][2 * s + 1]), q1, q2, q3);
                A1[s] = u32x4{p1[0], p1[1], q1[0], q1[1]};
                A2[s] = u32x4{p2[0], p2[1], q2[0], q2[1]};
                A3[s] = u32x4{p3[0], p3[1], q3[0], q3[1]};
            }
            if (j + kPre < kStemKG) issue(j + kPre, ring[j % kPre], keep[j % kPre]);
            const unsigned char* wb = &s_w[(size_t)(j * 3 * 2) * 1024 + lane * 16];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wb + (0 * 2 + nt) * 1024));
                const bf16x8 b2 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wb + (1 * 2 + nt) * 1024));
                const bf16x8 b3 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wb + (2 * 2 + nt) * 1024));
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f32x16 c = acc[s][nt];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A3[s]), b1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A1[s]), b3, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A2[s]), b2, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A2[s]), b1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A1[s]), b2, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A1[s]), b1, c, 0, 0, 0);
                    acc[s][nt] = c;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: accumulator register i of half s = pixel ox0 + 32 s + 8 (i / 4) + (i % 4) + 4 h, channel nt * 32 + col
        float* orow = a.out + (((size_t)b * a.Ho + oy) * a.Wo + ox0 + 4 * h) * a.Cout + col;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = acc[s][nt][i] * sc[nt] + sh[nt];
                    if (a.relu) v = fmaxf(v, 0.f);
                    orow[(size_t)(32 * s + 8 * (i >> 2) + (i & 3)) * a.Cout + 32 * nt] = v;
                }
    }
}

int launch_stem7x7(const StemArgs& a, hipStream_t s) {
    if (!a.in || !a.wpl || !a.out || a.B < 1 || a.Cout != 64 || a.Npad < 64 || a.Kpad != 224 || (a.Wo & 63) != 0 || a.Ho < 1 ||
        a.Ho != (a.Hi + 6 - 7) / 2 + 1 || a.Wo != (a.Wi + 6 - 7) / 2 + 1 || (long long)a.Hi * a.Wi * 4 >= (1LL << 31))
        return FPC_EINVAL;
    const long long waves = (long long)a.B * a.Ho * (a.Wo >> 6);
    if (waves >= (1LL << 31) - 8 * 4096) return FPC_EINVAL;
    const int grid = (int)std::min<long long>(a.grid > 0 ? a.grid : 256, (waves + 7) / 8);
    hipLaunchKernelGGL(k_stem7x7, dim3(grid), dim3(512), 0, s, a);
    return check_launch();
}

}  // namespace fpc
