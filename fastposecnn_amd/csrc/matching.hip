// matching.hip — 2D IoU of every (mask1, mask2) pair: gtf.batchwise_get_2d_iou
// (F/lib/gpu_tensor_funcs.py:386-409), the hot spot of mg.batchwise_find_matches (F/lib/matching.py:264-267).
//
// The reference expands both mask stacks to [n1,n2,H,W], takes logical_and / logical_or and sums: at 640x480
// that is 2 * n1*n2 * 307 200 bool elements written and re-read per call (38 MB for 8 x 8 masks), and it is
// called once per class.  Here every mask is read ONCE: k_mask_pack turns it into a bitset (one v_cmp per
// pixel, ballot -> 64 pixels per u64 word; 38 KB per 640x480 mask, L2-resident); k_mask_iou takes
// popcount(a & b) and popcount(a | b) over the words of a pair.  No atomics (a per-mask pixel counter updated by
// every wave cost 30-250 us in same-address atomics — measured — against 25 us for the whole read).
// Algorithmic bytes: (n1 + n2) * H*W * elem_size read + 4*n1*n2 written; HBM-bound.
// The bit order inside a word is a fixed permutation of the pixel order (the float4 path interleaves four
// pixels per lane): it is the same for both stacks, which is all AND / popcount need.
#include "common.hpp"

namespace fpc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// grid (blocks, n): mask blockIdx.y of the concatenated stack; words = cdiv(hw, 64) u64 per mask.
// VEC: hw % 256 == 0 and 16-byte aligned rows: a wave covers 256 pixels per step (float4 per lane, four words).
template <int ELEM, bool VEC>
__global__ __launch_bounds__(256) void k_mask_pack(const void* __restrict__ m1, int n1, const void* __restrict__ m2,
                                                   long long hw, long long words, unsigned long long* __restrict__ bits) {
    const int mi = blockIdx.y;
    const char* base = (const char*)(mi < n1 ? m1 : m2) + (size_t)(mi < n1 ? mi : mi - n1) * hw * ELEM;
    unsigned long long* out = bits + (size_t)mi * words;
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (long long)gridDim.x * 4;
    if (VEC) {
        for (long long g = wave; g * 256 < hw; g += nwave) {            // group g = pixels [256 g, 256 g + 256)
            f32x4 v = reinterpret_cast<const f32x4*>(base)[g * 64 + lane];
            unsigned long long w0 = __builtin_amdgcn_ballot_w64(v[0] != 0.0f), w1 = __builtin_amdgcn_ballot_w64(v[1] != 0.0f);
            unsigned long long w2 = __builtin_amdgcn_ballot_w64(v[2] != 0.0f), w3 = __builtin_amdgcn_ballot_w64(v[3] != 0.0f);
            if (lane < 4) out[g * 4 + lane] = lane == 0 ? w0 : lane == 1 ? w1 : lane == 2 ? w2 : w3;
        }
    } else {
        for (long long g = wave; g < words; g += nwave) {               // word g = pixels [64 g, 64 g + 64)
            long long p = g * 64 + lane;
            bool set = false;
            if (p < hw) set = ELEM == 4 ? reinterpret_cast<const float*>(base)[p] != 0.0f      // NaN is set, -0.0 is not
                                        : reinterpret_cast<const unsigned char*>(base)[p] != 0;
            unsigned long long w = __builtin_amdgcn_ballot_w64(set);
            if (lane == 0) out[g] = w;
        }
    }
}

// grid (n2, n1): one workgroup per pair.
__global__ __launch_bounds__(256) void k_mask_iou(const unsigned long long* __restrict__ bits,
                                                  int n1, int n2, long long words, float* __restrict__ iou,
                                                  int* __restrict__ inter_out, int* __restrict__ uni_out) {
    __shared__ int s_part[8];
    const int j = blockIdx.x, i = blockIdx.y;
    const unsigned long long* a = bits + (size_t)i * words;
    const unsigned long long* b = bits + (size_t)(n1 + j) * words;
    int c = 0, u = 0;
    for (long long w = threadIdx.x; w < words; w += 256) {
        unsigned long long x = a[w], y = b[w];
        c += __popcll(x & y);
        u += __popcll(x | y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c += __shfl_down(c, o, 64); u += __shfl_down(u, o, 64); }
    if ((threadIdx.x & 63) == 0) { s_part[threadIdx.x >> 6] = c; s_part[4 + (threadIdx.x >> 6)] = u; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int in = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        int un = s_part[4] + s_part[5] + s_part[6] + s_part[7];
        // torch: int64 / int64 -> both to float32, one IEEE division (0 / 0 = NaN for two empty masks)
        iou[(size_t)i * n2 + j] = (float)in / (float)un;
        if (inter_out) inter_out[(size_t)i * n2 + j] = in;
        if (uni_out) uni_out[(size_t)i * n2 + j] = un;
    }
}

static size_t iou_words(int64_t hw) { return (size_t)((hw + 63) / 64); }

}  // namespace fpc

using namespace fpc;

extern "C" size_t fpc_mask_iou_workspace_bytes(int n1, int n2, int64_t hw) {
    if (n1 < 0 || n2 < 0 || hw < 0) return 0;
    size_t n = (size_t)n1 + (size_t)n2;
    return align_up(n * iou_words(hw) * sizeof(unsigned long long), 256) + 256;
}

extern "C" int fpc_mask_iou(const void* masks1, int n1, const void* masks2, int n2, int64_t hw, int elem_size, float* iou,
                            int32_t* inter, int32_t* uni, void* ws, size_t ws_bytes, fpc_stream_t stream) {
    if (n1 < 0 || n2 < 0 || hw < 0 || (elem_size != 4 && elem_size != 1)) return FPC_EINVAL;
    if (hw >= ((int64_t)1 << 31)) return FPC_EINVAL;             // per-mask pixel counts are int32
    if (n1 == 0 || n2 == 0) return FPC_OK;                       // empty [n1, n2] result
    if (!masks1 || !masks2 || !iou || !ws) return FPC_EINVAL;
    if (((uintptr_t)ws & 255) != 0 || ws_bytes < fpc_mask_iou_workspace_bytes(n1, n2, hw)) return FPC_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int n = n1 + n2;
    const long long words = (long long)iou_words(hw);
    unsigned long long* bits = (unsigned long long*)ws;
    if (hw > 0) {
        const bool vec = elem_size == 4 && hw % 256 == 0 && (((uintptr_t)masks1 | (uintptr_t)masks2) & 15) == 0;
        long long groups = vec ? hw / 256 : words;
        dim3 grid((unsigned)std::min<long long>((groups + 3) / 4, 512), n);
        if (vec) hipLaunchKernelGGL((k_mask_pack<4, true>), grid, dim3(256), 0, s, masks1, n1, masks2, (long long)hw, words, bits);
        else if (elem_size == 4) hipLaunchKernelGGL((k_mask_pack<4, false>), grid, dim3(256), 0, s, masks1, n1, masks2, (long long)hw, words, bits);
        else hipLaunchKernelGGL((k_mask_pack<1, false>), grid, dim3(256), 0, s, masks1, n1, masks2, (long long)hw, words, bits);
        int rc = check_launch();
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_mask_iou, dim3(n2, n1), dim3(256), 0, s, bits, n1, n2, words, iou, inter, uni);
    return check_launch();
}
