// vote_count.hip — k_vote_count, the exact inlier counts of the fused hough vote (pipeline and filter: csrc/ransac.hip).
#include "vote.hpp"

namespace fpc {

// ---- k_vote_count ------------------------------------------------------------------
// EXACT inlier count of every hypothesis.  Task -> (count unit = 512 consecutive foreground ranks of an instance, slice of
// hypothesis tiles); the slicing is chosen on the device from the unit count so that the tasks fill one resident round of
// workgroups.  256 threads; wave w owns the unit's 64-entry groups w and w + 4.  Per group the wave builds the A
// fragments of the two forms (prologue, once per task); per hypothesis tile it loads ONE B fragment (16 bytes per lane)
// and issues two MFMAs per 32 entries; the result registers hold, per lane, ONE hypothesis (column lane & 31) against 16
// entries (rows), so counts stay lane-local:
//     r = F_t - |F_s| ;  row = (row << 2) | (r >> 30) ;  after 16: neg += popc(row & 0xAAAAAAAA), undecided = odd bit set & even clear.
// The unit record of the task after next and the entries of the NEXT task are requested while the current one computes
// (a task is ~1 us of arithmetic behind ~3 us of dependent loads otherwise); undecided pairs wait in the wave's queue
// ACROSS tasks, one 16-byte record per (lane, hypothesis tile) that has any.
// Error budget of r (units of the unscaled margin, M = |gx - ox| + |gy - oy| + radius, |e| = 1):
//     unit vote by v_rsq_f32 (1 ulp) and two products ............................ 4e-7 M   (s and t forms alike)
//     gx - ox, sigma (gx - ox): two roundings; c_s / c_t: three at <= radius ........ 3e-7 M
//     kappa2 e, kappa2 c_t: one rounding each ....................................... 1.2e-7 kappa M
//     dropped piece products ......................................................... 1.2e-7 M
//     f32 accumulation of 16 exact products inside the MFMA (any order, any rounding mode) ... <= 16 x 1.2e-7 M
//  => |r_computed - r_exact| / sigma <= 2.8e-6 (1 + kappa2) M  <  E = efac M  with efac = 3.2e-6 (1 + kappa1);
//     measured worst over 2 M random pairs: 4.2e-7 M (tools_dev/mfma_vote_probe.hip).
// dynamic LDS: [gps * 32] counts of the slice.

struct UnitRef { int inst, u, c_lo, nvalid, slot0, fic, fg; bool thin; float fox, foy; };
__device__ __forceinline__ UnitRef decode_unit(int4 a, int4 b) {
    UnitRef r;
    r.inst = a.x & 0xffff; r.thin = (a.x >> 16) & 1; r.nvalid = ((a.x >> 17) & 0x1ff) + 1;
    r.u = a.y; r.c_lo = a.z; r.fox = (float)(a.w & 0xffff); r.foy = (float)((unsigned)a.w >> 16);
    r.slot0 = b.x; r.fic = b.y; r.fg = b.z;
    return r;
}

// entry `ent` (0..511) of a unit: its first `fic` ranks sit in the unit's first chunk at consecutive slots
__device__ __forceinline__ float4 unit_entry(const float4* __restrict__ list, size_t ls, const int32_t* __restrict__ chunk_pre,
                                             int nch, const UnitRef& u, int ent) {
    int slot = u.slot0 + ent;
    if (ent >= u.fic) {
        int c = u.c_lo + 1;
        slot = rank_slot_from(chunk_pre + (size_t)u.inst * (nch + 1), c, u.u * kUnitEntries + ent);
    }
    return list[(size_t)u.inst * ls + slot];
}

__device__ __forceinline__ float lane_fetch(float v, int src_lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}

// The pairs the filter could not decide, one record per (lane, hypothesis tile) of THIS wave and THIS task:
// {bits of row tiles 0 | 1 << 1 of group w, the same of group w + 4, tile, lane}.  A lane takes a record and walks its bits
// with the reference's own arithmetic; the entry comes from the lane that holds it (ds_bpermute), the point from memory;
// the inliers go to the slice's LDS counts.
template <bool KEEP>
__device__ __forceinline__ void band_flush(int nq, const int4* __restrict__ queue, float4 qa, float4 qb, const UnitRef& u, int wv,
                                           int T0, int* __restrict__ s_cnt, const VoteParams& p,
                                           const int32_t* __restrict__ map = nullptr /* slot -> hypothesis (progressive count) */) {
    const int lane = threadIdx.x & (kWave - 1), hn = p.hn;
    for (int base = 0; base < nq; base += kWave) {                           // uniform
        const bool on = base + lane < nq;
        const int4 e = on ? queue[base + lane] : make_int4(0, 0, 0, 0);
        const int T = e.z, src_lane = e.w;
        int h = T * kHypTile + (src_lane & 31);
        if (map) h = on ? map[h] : 0;                                        // -1: padding of the last alive tile
        float gx = 0.f, gy = 0.f;
        if (on && h >= 0 && h < hn) { gx = p.hyp[((size_t)u.inst * hn + h) * 2]; gy = p.hyp[((size_t)u.inst * hn + h) * 2 + 1]; }
        unsigned w0 = (on && h >= 0 && h < hn) ? (unsigned)e.x : 0u, w1 = (on && h >= 0 && h < hn) ? (unsigned)e.y : 0u;
        int add = 0;
        while (__builtin_amdgcn_ballot_w64((w0 | w1) != 0u)) {              // uniform: every lane takes part in the fetches
            const bool has = (w0 | w1) != 0u;
            const int half = w0 ? 0 : 1;
            const unsigned w = w0 ? w0 : w1;
            const int b = has ? __ffs((int)w) - 1 : 0;
            if (w0) w0 &= w0 - 1u; else w1 &= w1 - 1u;
            const int i = 15 - (b >> 1);                                   // register index: the last one shifted in sits lowest
            const int eg = (b & 1) * 32 + (i & 3) + 8 * (i >> 2) + 4 * (src_lane >> 5);   // entry inside its 64-entry group
            const int src = has ? eg : lane;
            const float ax = lane_fetch(qa.x, src), ay = lane_fetch(qa.y, src), az = lane_fetch(qa.z, src), aw = lane_fetch(qa.w, src);
            const float bx = lane_fetch(qb.x, src), by = lane_fetch(qb.y, src), bz = lane_fetch(qb.z, src), bw = lane_fetch(qb.w, src);
            const float qx = half ? bx : ax, qy = half ? by : ay, qz = half ? bz : az, qw = half ? bw : aw;
            bool valid = has && (wv + 4 * half) * kWave + eg < u.nvalid;
            if (u.thin && valid) valid = pixel_kept<KEEP>(qx, qy, p.W, p.HW, u.inst, u.fg, p.max_num, p.seed, p.keep);
            if (valid && pair_is_inlier(qx, qy, qz, qw, sqrtf(qz * qz + qw * qw), gx, gy, p.thresh)) ++add;
        }
        if (add) atomicAdd(&s_cnt[(T - T0) * kHypTile + (src_lane & 31)], add);
    }
}

struct GroupFrags { u32x4 s[2], t[2]; };          // A fragments of one 64-entry group: forms s / t, row tiles 0 / 1

// the A fragments of this lane's entry, exchanged so that tile 0 = entries 0-31 and tile 1 = entries 32-63 of the group
__device__ __forceinline__ void build_group(GroupFrags& g, bool valid, float4 q, float fox, float foy, float kappa2) {
    float a_s = 0.f, b_s = 0.f, c_s = kNeverS, a_t = 0.f, b_t = 0.f, c_t = 0.f;
    const float n2 = q.z * q.z + q.w * q.w;
    // .cu:121 skips a vote with |d| < 1e-6 (compared in double): n1 <= 1e-6f in f32 (common.hpp); near that bound the
    // correctly rounded sqrt decides.  A non-finite or overflowing |d|^2 never votes either (NaN / 0 cosine).
    bool votes = valid && n2 <= 3.0e38f && n2 >= 4.0e-12f;
    if (__builtin_amdgcn_ballot_w64(valid && n2 < 4.0e-12f)) votes = votes || (valid && n2 < 4.0e-12f && !below_eps(sqrtf(n2)));
    if (votes) {
        const float inv = __builtin_amdgcn_rsqf(n2);
        const float ex = q.z * inv, ey = q.w * inv;
        const float xs = q.x - fox, ys = q.y - foy;
        a_s = ey; b_s = -ex; c_s = -(xs * ey - ys * ex);
        a_t = kappa2 * ex; b_t = kappa2 * ey; c_t = kappa2 * -(xs * ex + ys * ey);
    }
    u32x4 slo, shi, tlo, thi;
    a_fragment(a_s, b_s, c_s, 0.0f, slo, shi);
    a_fragment(a_t, b_t, c_t, -1.0f, tlo, thi);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // lanes 32-63 of `lo` <-> lanes 0-31 of `hi`: lo' = slots 0-7 | 8-15 of entries 0-31, hi' = the same of entries 32-63
        const auto a = __builtin_amdgcn_permlane32_swap(slo[r], shi[r], false, false);
        g.s[0][r] = a[0]; g.s[1][r] = a[1];
        const auto b = __builtin_amdgcn_permlane32_swap(tlo[r], thi[r], false, false);
        g.t[0][r] = b[0]; g.t[1][r] = b[1];
    }
}

// one row tile (32 entries) against one hypothesis tile: two MFMAs, 16 x (v_sub, v_alignbit); returns the 2-bit rows
__device__ __forceinline__ unsigned tile_rows(const u32x4& As, const u32x4& At, const bf16x8 B) {
    f32x16 Fs = {0}, Ft = {0};
    Fs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, As), B, Fs, 0, 0, 0);
    Ft = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, At), B, Ft, 0, 0, 0);
    unsigned row = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) row = __builtin_amdgcn_alignbit(row, __float_as_uint(Ft[i] - fabsf(Fs[i])), 30);
    return row;
}

template <int WAVES /* waves per SIMD the register allocation aims at */, bool KEEP>
__global__ __launch_bounds__(256, WAVES) void k_vote_count(const VoteParams p) {
    extern __shared__ __attribute__((aligned(16))) int s_cnt[];      // [gps * 32]
    __shared__ int4 s_bandq[4][kBandQ];
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    FPC_STAMP(2, 0);
    const int nu = p.ctrl[0];
    const int ntiles = p.ntiles;
    // units x S slices ~ the task count the launch was sized for: one round of equal tasks over the chip
    const int S0 = max(1, min(ntiles, p.task_target / max(nu, 1)));
    const int gps = min(kMaxSliceTiles, (ntiles + S0 - 1) / S0);
    const int S = (ntiles + gps - 1) / gps;
    // task t = (unit t / S, slice t % S), t = blockIdx.x + k gridDim.x: unit and slice advance without a division per task
    const int G = gridDim.x, Gu = G / S, Gs = G - Gu * S;
    int4* bq = s_bandq[wv];
    if ((long long)blockIdx.x >= (long long)nu * S) return;                  // uniform
    auto advance = [&](int& uu, int& ss) { uu += Gu; ss += Gs; if (ss >= S) { ss -= S; ++uu; } };
    // software pipeline over this workgroup's tasks: unit records two tasks ahead, entries one task ahead
    auto load_entries = [&](const UnitRef& u, float4& qa, float4& qb) {
        const int ea = wv * kWave + lane, eb = (wv + 4) * kWave + lane;
        qa = ea < u.nvalid ? unit_entry(p.list, p.ls, p.chunk_pre, p.nch, u, ea) : make_float4(0.f, 0.f, 0.f, 0.f);
        qb = eb < u.nvalid ? unit_entry(p.list, p.ls, p.chunk_pre, p.nch, u, eb) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    const int4 zero4 = make_int4(0, 0, 0, 0);
    int uidx = blockIdx.x / S, s = blockIdx.x - uidx * S;                    // this task
    int u1 = uidx, s1 = s; advance(u1, s1);                                   // the next
    int u2 = u1, s2 = s1; advance(u2, s2);                                    // the one after
    int4 ua = p.units[2 * uidx], ub = p.units[2 * uidx + 1];
    int4 ua1 = zero4, ub1 = zero4;
    if (u1 < nu) { ua1 = p.units[2 * u1]; ub1 = p.units[2 * u1 + 1]; }
    float4 qa, qb;
    load_entries(decode_unit(ua, ub), qa, qb);
    FPC_STAMP(2, 1);
    for (;;) {
        const bool more = u1 < nu;
        int4 ua2 = zero4, ub2 = zero4;
        if (u2 < nu) { ua2 = p.units[2 * u2]; ub2 = p.units[2 * u2 + 1]; }
        float4 qan = make_float4(0.f, 0.f, 0.f, 0.f), qbn = qan;
        if (more) load_entries(decode_unit(ua1, ub1), qan, qbn);
        const UnitRef u = decode_unit(ua, ub);
        const int inst = u.inst;
        const int T0 = s * gps, T1 = min(ntiles, T0 + gps);
        // this wave's groups: w and w + 4 of the unit's eight
        const int ng = (wv * kWave < u.nvalid ? 1 : 0) + ((wv + 4) * kWave < u.nvalid ? 1 : 0);
        GroupFrags Gf[2];
        {
            bool va = wv * kWave + lane < u.nvalid, vb = (wv + 4) * kWave + lane < u.nvalid;
            if (u.thin) {
                va = va && pixel_kept<KEEP>(qa.x, qa.y, p.W, p.HW, inst, u.fg, p.max_num, p.seed, p.keep);
                vb = vb && pixel_kept<KEEP>(qb.x, qb.y, p.W, p.HW, inst, u.fg, p.max_num, p.seed, p.keep);
            }
            build_group(Gf[0], va, qa, u.fox, u.foy, p.kappa2);
            build_group(Gf[1], vb, qb, u.fox, u.foy, p.kappa2);
        }
        for (int i = threadIdx.x; i < (T1 - T0) * kHypTile; i += blockDim.x) s_cnt[i] = 0;
        __syncthreads();
        FPC_STAMP(2, 2);
        if (ng > 0) {                                                       // uniform per wave
            const u32x4* Bp = p.hypB + ((size_t)inst * ntiles + T0) * kWave + lane;
            u32x4 Bn = *Bp;
            int qn = 0;
            for (int T = T0; T < T1; ++T) {
                const bf16x8 B = __builtin_bit_cast(bf16x8, Bn);
                if (T + 1 < T1) Bn = Bp[(size_t)(T + 1 - T0) * kWave];
                unsigned w01 = 0u, w23 = 0u;
                int neg = 0;
                {
                    const unsigned r0 = tile_rows(Gf[0].s[0], Gf[0].t[0], B), r1 = tile_rows(Gf[0].s[1], Gf[0].t[1], B);
                    neg = __popc(r0 & 0xAAAAAAAAu) + __popc(r1 & 0xAAAAAAAAu);
                    w01 = ((r0 >> 1) & ~r0 & 0x55555555u) | (r1 & ~(r1 << 1) & 0xAAAAAAAAu);
                }
                if (ng > 1) {                                               // uniform per wave
                    const unsigned r2 = tile_rows(Gf[1].s[0], Gf[1].t[0], B), r3 = tile_rows(Gf[1].s[1], Gf[1].t[1], B);
                    neg += __popc(r2 & 0xAAAAAAAAu) + __popc(r3 & 0xAAAAAAAAu);
                    w23 = ((r2 >> 1) & ~r2 & 0x55555555u) | (r3 & ~(r3 << 1) & 0xAAAAAAAAu);
                }
                // lane-local: column (lane & 31) of tile T against 16 rows x 2 tiles x ng groups
                atomicAdd(&s_cnt[(T - T0) * kHypTile + (lane & 31)], ng * 32 - neg);
                // undecided pairs -> one record per lane that has any (a handful per step)
                const bool has = (w01 | w23) != 0u;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(has);
                if (mk) {                                                   // uniform
                    const int add = __popcll(mk);
                    if (qn + add > kBandQ) {
                        band_flush<KEEP>(qn, bq, qa, qb, u, wv, T0, s_cnt, p);
                        qn = 0;
                    }
                    if (has)
                        bq[qn + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0))] =
                            make_int4((int)w01, (int)w23, T, lane);
                    qn += add;
                }
            }
            if (qn) band_flush<KEEP>(qn, bq, qa, qb, u, wv, T0, s_cnt, p);
        }
        FPC_STAMP(2, 3);
        __syncthreads();
        FPC_STAMP(2, 4);
        // one integer atomic per (unit, hypothesis) with any count: order-independent result
        for (int i = threadIdx.x; i < (T1 - T0) * kHypTile; i += blockDim.x) {
            const int h = T0 * kHypTile + i, cv = s_cnt[i];
            if (cv && h < p.hn) atomicAdd(&p.counts[(size_t)inst * p.hnp + h], cv);
        }
        __syncthreads();
        if (!more) break;
        uidx = u1; s = s1; u1 = u2; s1 = s2; advance(u2, s2);
        ua = ua1; ub = ub1; ua1 = ua2; ub1 = ub2; qa = qan; qb = qbn;
    }
    FPC_STAMP(2, 5);
}

// ---- k_vote_count_prog ----------------------------------------------------------------
// One PASS of the progressive count (csrc/ransac.hip: k_vote_lead): the unit positions [pass_begin(pass), pass_begin(pass + 1))
// of every instance against the instance's ALIVE hypothesis tiles (pass 0: all of them, p.hypB; later: p.hypC and the slot ->
// hypothesis map k_vote_lead wrote).  The work items (unit, tile) of the pass are numbered instance by instance, unit by
// unit, and every workgroup takes ONE contiguous range of equal length (a unit that straddles two ranges is staged by both
// workgroups): no tail round.  Per segment = (unit, tile range) the arithmetic is k_vote_count's, to the instruction.
// dynamic LDS: [kMaxSliceTiles * 32] counts, then per instance: items before it [n + 1], first unit record, units, tiles.
struct Seg { int inst, urec, ta, tb; };

template <bool KEEP>
__global__ __launch_bounds__(256, 4) void k_vote_count_prog(const VoteParams p, const int pass) {
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];
    __shared__ int4 s_bandq[4][kBandQ];
    __shared__ int s_w[8];
    int* s_cnt = s_dyn;
    int* s_pre = s_dyn + kMaxSliceTiles * kHypTile;
    int* s_ub = s_pre + (p.n + 1);
    int* s_U = s_ub + p.n;
    int* s_T = s_U + p.n;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    FPC_STAMP(2, 8 * pass);
#ifdef FPC_STAMP_VOTE
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x < 1024) { p.dbg[(pass * 1024 + blockIdx.x) * 4] = dbg_t0; p.dbg[(pass * 1024 + blockIdx.x) * 4 + 1] = dbg_t0; }
#endif
    const int n_act = active_instances(p.n, p.n_dev);
    for (int inst = threadIdx.x; inst < n_act; inst += blockDim.x) {
        const int4 pi = *reinterpret_cast<const int4*>(p.pinfo + (size_t)inst * kPInfoI);      // unit base, units, alive, tiles
        const int b0 = pass_begin(p.pcum, p.npass, pass, pi.y), b1 = pass_begin(p.pcum, p.npass, pass + 1, pi.y);
        s_ub[inst] = pi.x + b0; s_U[inst] = b1 - b0; s_T[inst] = pi.w;
    }
    __syncthreads();
    const int total = __builtin_amdgcn_readfirstlane(block_scan([&](int i) { return s_U[i] * s_T[i]; }, s_pre, n_act, s_w));
    const int Q = max(4, (total + (int)gridDim.x - 1) / (int)gridDim.x);
    const long long begin_ll = (long long)blockIdx.x * Q;
    if (begin_ll >= total) return;                                           // uniform
    const int begin = (int)begin_ll, end = min(total, begin + Q);
    auto sgpr = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    int hint;
    {   // the instance of the first item
        int lo = 0, hi = n_act;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sgpr(s_pre[mid]) <= begin) lo = mid; else hi = mid; }
        hint = lo;
    }
    // (every value here is uniform over the workgroup: readfirstlane keeps the descriptors in scalar registers)
    auto seg_at = [&](int i) -> Seg {
        while (sgpr(s_pre[hint + 1]) <= i) ++hint;                           // items only grow
        const int T = sgpr(s_T[hint]), local = i - sgpr(s_pre[hint]);
        const int k = local / T, ta = local - k * T;
        const int tb = min(min(T, ta + (end - i)), ta + kMaxSliceTiles);
        return Seg{hint, sgpr(s_ub[hint]) + k, ta, tb};
    };
    const int32_t* map_all = pass == 0 ? nullptr : p.hmap + (size_t)(pass & 1) * p.n * p.hnp;
    const u32x4* B_all = pass == 0 ? p.hypB : p.hypC;
    int4* bq = s_bandq[wv];
    auto load_entries = [&](const UnitRef& u, float4& qa, float4& qb) {
        const int ea = wv * kWave + lane, eb = (wv + 4) * kWave + lane;
        qa = ea < u.nvalid ? unit_entry(p.list, p.ls, p.chunk_pre, p.nch, u, ea) : make_float4(0.f, 0.f, 0.f, 0.f);
        qb = eb < u.nvalid ? unit_entry(p.list, p.ls, p.chunk_pre, p.nch, u, eb) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    const int4 zero4 = make_int4(0, 0, 0, 0);
    // software pipeline over this workgroup's segments: unit records two ahead, entries one ahead
    Seg c0 = seg_at(begin), c1 = c0, c2 = c0;
    int i1 = begin + (c0.tb - c0.ta), i2 = end;
    if (i1 < end) { c1 = seg_at(i1); i2 = i1 + (c1.tb - c1.ta); }
    int4 ua = p.units[2 * c0.urec], ub = p.units[2 * c0.urec + 1];
    int4 ua1 = zero4, ub1 = zero4;
    if (i1 < end) { ua1 = p.units[2 * c1.urec]; ub1 = p.units[2 * c1.urec + 1]; }
    float4 qa, qb;
    load_entries(decode_unit(ua, ub), qa, qb);
    FPC_STAMP(2, 8 * pass + 1);
    int nseg = 0;
    for (;;) {
        const bool more = i1 < end;
        int4 ua2 = zero4, ub2 = zero4;
        int i3 = end;
        if (i2 < end) { c2 = seg_at(i2); i3 = i2 + (c2.tb - c2.ta); ua2 = p.units[2 * c2.urec]; ub2 = p.units[2 * c2.urec + 1]; }
        float4 qan = make_float4(0.f, 0.f, 0.f, 0.f), qbn = qan;
        if (more) load_entries(decode_unit(ua1, ub1), qan, qbn);
        const UnitRef u = decode_unit(ua, ub);
        const int inst = u.inst;
        const int T0 = c0.ta, T1 = c0.tb;
        const int32_t* map = map_all ? map_all + (size_t)inst * p.hnp : nullptr;
        const int ng = (wv * kWave < u.nvalid ? 1 : 0) + ((wv + 4) * kWave < u.nvalid ? 1 : 0);
        GroupFrags Gf[2];
        {
            bool va = wv * kWave + lane < u.nvalid, vb = (wv + 4) * kWave + lane < u.nvalid;
            if (u.thin) {
                va = va && pixel_kept<KEEP>(qa.x, qa.y, p.W, p.HW, inst, u.fg, p.max_num, p.seed, p.keep);
                vb = vb && pixel_kept<KEEP>(qb.x, qb.y, p.W, p.HW, inst, u.fg, p.max_num, p.seed, p.keep);
            }
            build_group(Gf[0], va, qa, u.fox, u.foy, p.kappa2);
            build_group(Gf[1], vb, qb, u.fox, u.foy, p.kappa2);
        }
        for (int i = threadIdx.x; i < (T1 - T0) * kHypTile; i += blockDim.x) s_cnt[i] = 0;
        __syncthreads();
        if (nseg == 0) FPC_STAMP(2, 8 * pass + 2);
        if (ng > 0) {                                                       // uniform per wave
            const u32x4* Bp = B_all + ((size_t)inst * p.ntiles + T0) * kWave + lane;
            u32x4 Bn = *Bp;
            int qn = 0;
            for (int T = T0; T < T1; ++T) {
                const bf16x8 B = __builtin_bit_cast(bf16x8, Bn);
                if (T + 1 < T1) Bn = Bp[(size_t)(T + 1 - T0) * kWave];
                unsigned w01 = 0u, w23 = 0u;
                int neg = 0;
                {
                    const unsigned r0 = tile_rows(Gf[0].s[0], Gf[0].t[0], B), r1 = tile_rows(Gf[0].s[1], Gf[0].t[1], B);
                    neg = __popc(r0 & 0xAAAAAAAAu) + __popc(r1 & 0xAAAAAAAAu);
                    w01 = ((r0 >> 1) & ~r0 & 0x55555555u) | (r1 & ~(r1 << 1) & 0xAAAAAAAAu);
                }
                if (ng > 1) {                                               // uniform per wave
                    const unsigned r2 = tile_rows(Gf[1].s[0], Gf[1].t[0], B), r3 = tile_rows(Gf[1].s[1], Gf[1].t[1], B);
                    neg += __popc(r2 & 0xAAAAAAAAu) + __popc(r3 & 0xAAAAAAAAu);
                    w23 = ((r2 >> 1) & ~r2 & 0x55555555u) | (r3 & ~(r3 << 1) & 0xAAAAAAAAu);
                }
                atomicAdd(&s_cnt[(T - T0) * kHypTile + (lane & 31)], ng * 32 - neg);
                const bool has = (w01 | w23) != 0u;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(has);
                if (mk) {                                                   // uniform
                    const int add = __popcll(mk);
                    if (qn + add > kBandQ) {
                        band_flush<KEEP>(qn, bq, qa, qb, u, wv, T0, s_cnt, p, map);
                        qn = 0;
                    }
                    if (has)
                        bq[qn + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0))] =
                            make_int4((int)w01, (int)w23, T, lane);
                    qn += add;
                }
            }
            if (qn) band_flush<KEEP>(qn, bq, qa, qb, u, wv, T0, s_cnt, p, map);
        }
        __syncthreads();
        if (nseg == 0) FPC_STAMP(2, 8 * pass + 3);
        for (int i = threadIdx.x; i < (T1 - T0) * kHypTile; i += blockDim.x) {
            const int cv = s_cnt[i];
            if (cv) {
                const int slot = T0 * kHypTile + i, h = map ? map[slot] : slot;
                if (h >= 0 && h < p.hn) atomicAdd(&p.counts[(size_t)inst * p.hnp + h], cv);
            }
        }
        __syncthreads();
        if (nseg == 0) FPC_STAMP(2, 8 * pass + 4);
        ++nseg;
        if (!more) break;
        c0 = c1; c1 = c2; i1 = i2; i2 = i3;
        ua = ua1; ub = ub1; ua1 = ua2; ub1 = ub2; qa = qan; qb = qbn;
    }
    FPC_STAMP(2, 8 * pass + 5);
#ifdef FPC_STAMP_VOTE
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        p.dbg[(pass * 1024 + blockIdx.x) * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        p.dbg[(pass * 1024 + blockIdx.x) * 4 + 2] = (unsigned long long)nseg | ((unsigned long long)(end - begin) << 16);
        p.dbg[(pass * 1024 + blockIdx.x) * 4 + 3] = (unsigned long long)(xcc & 15) | ((unsigned long long)hwid << 8);
    }
#endif
#ifdef FPC_STAMP_VOTE
    if (blockIdx.x == 0 && threadIdx.x == 0) { p.stamps[2 * 32 + 8 * pass + 6] = (unsigned long long)nseg; p.stamps[2 * 32 + 8 * pass + 7] = (unsigned long long)(end - begin); }
#endif
}


void launch_vote_count(const VoteParams& p, int grid, size_t lds_bytes, hipStream_t s) {
    if (p.keep) hipLaunchKernelGGL((k_vote_count<4, true>), dim3(grid), dim3(256), lds_bytes, s, p);
    else hipLaunchKernelGGL((k_vote_count<4, false>), dim3(grid), dim3(256), lds_bytes, s, p);
}

void launch_vote_count_prog(const VoteParams& p, int pass, int grid, hipStream_t s) {
    const size_t lds = sizeof(int) * ((size_t)kMaxSliceTiles * kHypTile + 4 * (size_t)p.n + 1);
    if (p.keep) hipLaunchKernelGGL((k_vote_count_prog<true>), dim3(grid), dim3(256), lds, s, p, pass);
    else hipLaunchKernelGGL((k_vote_count_prog<false>), dim3(grid), dim3(256), lds, s, p, pass);
}

}  // namespace fpc
