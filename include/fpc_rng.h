/* fpc_rng.h — counter-based sampling used when the caller does not inject
 * RANSAC pair indices / a thinning selection.
 *
 * The reference draws both from torch's global device RNG
 * (RV/ransac_voting_gpu.py:552 `random_(0, tn)`, :543 `uniform_(0,1)`), a
 * stream that is not reproducible across back ends.  This header is the
 * specification of OUR stream; it is shared verbatim by the HIP library
 * (fastposecnn_amd/csrc) and by the CPU oracle (oracle/fpc_oracle.c) so that
 * "same seed -> same samples" holds across the two.  Plain C, no dependencies.
 */
#ifndef FPC_RNG_H_
#define FPC_RNG_H_

#include <stdint.h>

#ifdef __HIPCC__
#define FPC_HD __host__ __device__ static inline
#else
#define FPC_HD static inline
#endif

/* 3-word avalanche hash (murmur3 fmix32 rounds over a running state). */
FPC_HD uint32_t fpc_mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu;
    h ^= h >> 13; h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

FPC_HD uint32_t fpc_hash3(uint64_t seed, uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = fpc_mix32((uint32_t)seed ^ 0x9e3779b9u);
    h = fpc_mix32(h ^ (uint32_t)(seed >> 32));
    h = fpc_mix32(h ^ (a * 0x9e3779b1u + 0x7f4a7c15u));
    h = fpc_mix32(h ^ (b * 0x85ebca77u + 0x165667b1u));
    h = fpc_mix32(h ^ (c * 0xc2b2ae3du + 0x27d4eb2fu));
    return h;
}

/* Uniform integer in [0, n) (Lemire multiply-shift; n >= 1). */
FPC_HD int32_t fpc_rand_index(uint64_t seed, uint32_t inst, uint32_t hyp, uint32_t which, uint32_t n) {
    uint32_t r = fpc_hash3(seed, inst, hyp, which);
    return (int32_t)(((uint64_t)r * (uint64_t)n) >> 32);
}

/* Pair sampling of a THINNED instance (fg > max_num, RV/ransac_voting_gpu.py:541-552).  The reference compacts the kept
 * pixels and draws ranks among them; an equivalent draw that needs no kept-rank table: draw a rank among ALL fg foreground
 * pixels (raster order) and reject it while that pixel is thinned out, attempt a = 0, 1, ... using the stream word
 * `which + 2 a` (which = 0 / 1: first / second point of the pair); after FPC_SAMPLE_MAX_TRIES rejected attempts the last
 * draw is used as it is (probability (1 - max_num / fg)^256).  An instance that is not thinned takes attempt 0:
 * fpc_rand_index(seed, inst, hyp, which, fg), the same draw as before this rule existed. */
#define FPC_SAMPLE_MAX_TRIES 256

/* Bernoulli keep decision for the > max_num thinning: keep iff u < max_num/fg,
 * evaluated in integers as  r * fg < max_num * 2^32  (r uniform 32-bit). */
FPC_HD int fpc_rand_keep(uint64_t seed, uint32_t inst, uint32_t pixel, uint32_t fg, uint32_t max_num) {
    uint32_t r = fpc_hash3(seed, inst, pixel, 0xfeedu);
    return ((uint64_t)r * (uint64_t)fg) < ((uint64_t)max_num << 32);
}

#endif /* FPC_RNG_H_ */
