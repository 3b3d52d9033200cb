/* fpc_oracle.c — CPU restatement of the FastPoseCNN post-network hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity ORACLE: it may be called
 * from tests/, from __graft_entry__.smoke() and from bench.py's cpu_baseline
 * leg, and from nowhere else.  The product (fastposecnn_amd/) never links,
 * imports or falls back to it.
 *
 * Pinning: every function below is checked against golden vectors produced by
 * importing the reference's own Python in the build container
 * (oracle/gen_golden.py -> tests/golden/ *.npz, tests/test_oracle_golden.py).
 * The reference's two live CUDA kernels cannot be compiled here (no NVIDIA
 * toolchain; removed ATen APIs), and the reference ships no test for them, so
 * fpco_generate_hypothesis / fpco_voting_for_hypothesis are a line-by-line
 * restatement of the kernel bodies, pinned by known-answer vectors only.
 *
 * Citation shorthand: RV/ = source_code/FastPoseCNN/lib/ransac_voting_gpu_layer/,
 * F/ = source_code/FastPoseCNN/ (both under /root/reference).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 * Contraction is OFF on purpose: every +,-,*,/ and sqrt is a separately
 * rounded IEEE binary32 operation, which is what the HIP kernels are compiled
 * to as well (nvcc's own fmad choices for the original are not observable
 * from here).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/fpc_rng.h"

#define FPCO_OK 0
#define FPCO_EINVAL (-1)
#define FPCO_ENOMEM (-2)

/* ------------------------------------------------------------------------ */
/* RV/src/ransac_voting_kernel.cu:11-49  generate_hypothesis_kernel          */
/* direct [tn,vn,2], coords [tn,2] (x=col, y=row), idxs [hn,vn,2]            */
/* -> hyp [hn,vn,2], zero-initialised as the launcher does (:75).            */
#ifdef _OPENMP
#include <omp.h>
#endif
/* Threads used by the parallel loops (bench.py's cpu_baseline): n <= 0 = the cores this process may use, capped at
 * FPCO_MAX_THREADS (default 16: a GPU box gives one GPU's job 16 of its 256 logical CPUs, and 256 threads on that share
 * ran the vote 3.7x SLOWER than one).  Returns the count in effect.  The count is THIS library's own: it is handed to
 * every parallel loop through a num_threads clause and the process-wide OpenMP setting (which torch's CPU kernels
 * share) is never touched — round 5's omp_set_num_threads(1) here left torch on one thread for the rest of the run. */
static int fpco_threads = 0;          /* 0 = not chosen yet: the capped core count at first use */
static int fpco_default_threads(void) {
#ifdef _OPENMP
    const char* cap = getenv("FPCO_MAX_THREADS");
    int lim = cap ? atoi(cap) : 16;
    int n = omp_get_num_procs();
    if (lim > 0 && n > lim) n = lim;
    return n < 1 ? 1 : n;
#else
    return 1;
#endif
}
int fpco_set_threads(int n) {
    fpco_threads = n <= 0 ? fpco_default_threads() : n;
#ifndef _OPENMP
    fpco_threads = 1;
#endif
    return fpco_threads;
}
int fpco_get_threads(void) {
    if (fpco_threads <= 0) fpco_threads = fpco_default_threads();
    return fpco_threads;
}

int fpco_generate_hypothesis(const float* direct, const float* coords, const int32_t* idxs,
                             float* hyp, int tn, int vn, int hn) {
    if (tn < 0 || vn < 1 || hn < 0) return FPCO_EINVAL;
    memset(hyp, 0, sizeof(float) * (size_t)hn * vn * 2);
    for (int hi = 0; hi < hn; ++hi) {
        for (int vi = 0; vi < vn; ++vi) {
            int t0 = idxs[hi * vn * 2 + vi * 2];
            int t1 = idxs[hi * vn * 2 + vi * 2 + 1];
            if (t0 < 0 || t0 >= tn || t1 < 0 || t1 >= tn) return FPCO_EINVAL;

            float nx0 = direct[t0 * vn * 2 + vi * 2 + 1];
            float ny0 = -direct[t0 * vn * 2 + vi * 2];
            float cx0 = coords[t0 * 2];
            float cy0 = coords[t0 * 2 + 1];

            float nx1 = direct[t1 * vn * 2 + vi * 2 + 1];
            float ny1 = -direct[t1 * vn * 2 + vi * 2];
            float cx1 = coords[t1 * 2];
            float cy1 = coords[t1 * 2 + 1];

            /* .cu:42-43 — float |det| promoted to double against the double literal */
            float det_y = nx1 * ny0 - nx0 * ny1;
            float det_x = ny1 * nx0 - ny0 * nx1;
            if ((double)fabsf(det_y) < 1e-6) continue;
            if ((double)fabsf(det_x) < 1e-6) continue;
            /* .cu:44-45 */
            float y = (nx1 * (nx0 * cx0 + ny0 * cy0) - nx0 * (nx1 * cx1 + ny1 * cy1)) / det_y;
            float x = (ny1 * (nx0 * cx0 + ny0 * cy0) - ny0 * (nx1 * cx1 + ny1 * cy1)) / det_x;

            hyp[hi * vn * 2 + vi * 2] = x;
            hyp[hi * vn * 2 + vi * 2 + 1] = y;
        }
    }
    return FPCO_OK;
}

/* One pair test, RV/src/ransac_voting_kernel.cu:106-125. Returns 0/1. */
static inline int fpco_pair_is_inlier(float cx, float cy, float nx, float ny,
                                      float hx, float hy, float thresh) {
    float dx = hx - cx;
    float dy = hy - cy;
    float norm1 = sqrtf(nx * nx + ny * ny);
    float norm2 = sqrtf(dx * dx + dy * dy);
    if ((double)norm1 < 1e-6 || (double)norm2 < 1e-6) return 0;
    float angle_dist = (dx * nx + dy * ny) / (norm1 * norm2);
    return angle_dist > thresh;
}

/* RV/src/ransac_voting_kernel.cu:88-126  voting_for_hypothesis_kernel.       */
/* inliers [hn,vn,tn] is caller-owned, only ever written with 1 (caller       */
/* pre-zeroes, RV/ransac_voting_gpu.py:562).                                  */
int fpco_voting_for_hypothesis(const float* direct, const float* coords, const float* hyp,
                               uint8_t* inliers, int tn, int vn, int hn, float thresh) {
    if (tn < 0 || vn < 1 || hn < 0) return FPCO_EINVAL;
    for (int hi = 0; hi < hn; ++hi)
        for (int vi = 0; vi < vn; ++vi) {
            float hx = hyp[hi * vn * 2 + vi * 2];
            float hy = hyp[hi * vn * 2 + vi * 2 + 1];
            uint8_t* row = inliers + ((size_t)hi * vn + vi) * tn;
            for (int ti = 0; ti < tn; ++ti) {
                if (fpco_pair_is_inlier(coords[ti * 2], coords[ti * 2 + 1],
                                        direct[ti * vn * 2 + vi * 2], direct[ti * vn * 2 + vi * 2 + 1],
                                        hx, hy, thresh))
                    row[ti] = 1;
            }
        }
    return FPCO_OK;
}

/* ------------------------------------------------------------------------ */
/* Symmetric 2x2 solve with b_inv semantics (RV/ransac_voting_gpu.py:503-516):*/
/* exact inverse when regular, Moore-Penrose pseudo-inverse when singular.    */
/* a = [[a00,a01],[a01,a11]] (sum of n n^T, PSD).                             */
static void fpco_solve2_sym(double a00, double a01, double a11, double b0, double b1,
                            double* x0, double* x1) {
    double tr = a00 + a11;
    double det = a00 * a11 - a01 * a01;
    if (!(tr > 0.0)) { *x0 = 0.0; *x1 = 0.0; return; }          /* rank 0: pinv(0) = 0 */
    if (det <= 1e-12 * tr * tr) {                               /* rank 1: A+ = A / tr^2 */
        double s = 1.0 / (tr * tr);
        *x0 = (a00 * b0 + a01 * b1) * s;
        *x1 = (a01 * b0 + a11 * b1) * s;
        return;
    }
    double inv = 1.0 / det;
    *x0 = (a11 * b0 - a01 * b1) * inv;
    *x1 = (-a01 * b0 + a00 * b1) * inv;
}

/* ransac_voting_layer_v3, RV/ransac_voting_gpu.py:518-607, for ONE keypoint   */
/* channel (vn = 1 per call; FastPoseCNN always uses vn = 1,                   */
/* F/lib/hough_voting.py:51).                                                  */
/*                                                                             */
/* mask   : f32 [n,H,W] contiguous; a pixel is foreground iff mask != 0 (:532) */
/* vertex : base pointer + element strides for the [n,H,W,(vn),2] view (:521)  */
/* idxs   : i32 [n,hn,2] injected pair indices, or NULL -> fpc_rng.h stream    */
/* keep   : u8 [n,H,W] injected thinning selection (applied only where the     */
/*          foreground exceeds max_num, :542-545), or NULL -> fpc_rng.h stream */
/* out_xy : f32 [n,2]; diagnostics (nullable): out_tn, out_win_idx,            */
/*          out_win_count, out_inl_count (i32 [n]), out_hyp (f32 [n,hn,2]),    */
/*          out_counts (i32 [n,hn]).                                           */
/*                                                                             */
/* The reference's while-loop (:557-581) re-evaluates the SAME idxs every      */
/* round (idxs is drawn once, :552) and only replaces the best on a strictly   */
/* larger ratio (:572), so its result equals one round's arg-max; this         */
/* restatement runs one round.                                                 */
int fpco_ransac_voting_v3(const float* mask, const float* vertex,
                          int64_t vs_n, int64_t vs_h, int64_t vs_w, int64_t vs_c,
                          int n, int H, int W, int hn,
                          const int32_t* idxs, const uint8_t* keep, uint64_t seed,
                          float thresh, int min_num, int max_num,
                          float* out_xy, int32_t* out_tn, int32_t* out_win_idx,
                          int32_t* out_win_count, int32_t* out_inl_count,
                          float* out_hyp, int32_t* out_counts) {
    if (n < 0 || H < 1 || W < 1 || hn < 1) return FPCO_EINVAL;
    size_t HW = (size_t)H * W;
    float* coords = (float*)malloc(sizeof(float) * 2 * HW);
    float* direct = (float*)malloc(sizeof(float) * 2 * HW);
    float* hyp = (float*)malloc(sizeof(float) * 2 * (size_t)hn);
    int32_t* pair = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)hn);
    if (!coords || !direct || !hyp || !pair) { free(coords); free(direct); free(hyp); free(pair); return FPCO_ENOMEM; }
    int rc = FPCO_OK;

    for (int bi = 0; bi < n && rc == FPCO_OK; ++bi) {
        const float* m = mask + (size_t)bi * HW;
        const float* v = vertex + (int64_t)bi * vs_n;
        int fg = 0;
        for (size_t p = 0; p < HW; ++p) fg += (m[p] != 0.0f);

        float* oxy = out_xy + 2 * (size_t)bi;
        if (out_tn) out_tn[bi] = 0;
        if (out_win_idx) out_win_idx[bi] = -1;
        if (out_win_count) out_win_count[bi] = 0;
        if (out_inl_count) out_inl_count[bi] = 0;
        if (out_hyp) memset(out_hyp + 2 * (size_t)bi * hn, 0, sizeof(float) * 2 * (size_t)hn);
        if (out_counts) memset(out_counts + (size_t)bi * hn, 0, sizeof(int32_t) * (size_t)hn);

        /* :536-539 too few points -> zeros */
        if (fg < min_num) { oxy[0] = 0.0f; oxy[1] = 0.0f; continue; }

        /* :541-550 compaction in raster order, x = column, y = row */
        int thin = fg > max_num;
        int tn = 0;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                size_t p = (size_t)y * W + x;
                if (m[p] == 0.0f) continue;
                if (thin) {
                    int k = keep ? (keep[(size_t)bi * HW + p] != 0)
                                 : fpc_rand_keep(seed, (uint32_t)bi, (uint32_t)p, (uint32_t)fg, (uint32_t)max_num);
                    if (!k) continue;
                }
                coords[2 * tn] = (float)x;
                coords[2 * tn + 1] = (float)y;
                direct[2 * tn] = v[(int64_t)y * vs_h + (int64_t)x * vs_w];
                direct[2 * tn + 1] = v[(int64_t)y * vs_h + (int64_t)x * vs_w + vs_c];
                ++tn;
            }
        if (out_tn) out_tn[bi] = tn;
        if (tn == 0) { oxy[0] = 0.0f; oxy[1] = 0.0f; continue; }

        /* :552 pair indices.  Injected, or our own stream (include/fpc_rng.h).  For a thinned instance the stream draws a
         * rank among ALL foreground pixels and rejects thinned-out ones (FPC_SAMPLE_MAX_TRIES, then the last draw as it
         * is): the same distribution as the reference's draw among the kept pixels, without a kept-rank table. */
        if (idxs || !thin) {
            for (int hi = 0; hi < hn; ++hi)
                for (int k = 0; k < 2; ++k)
                    pair[2 * hi + k] = idxs ? idxs[((size_t)bi * hn + hi) * 2 + k]
                                            : fpc_rand_index(seed, (uint32_t)bi, (uint32_t)hi, (uint32_t)k, (uint32_t)tn);
            /* :559 */
            rc = fpco_generate_hypothesis(direct, coords, pair, hyp, tn, 1, hn);
            if (rc != FPCO_OK) break;
        } else {
            int32_t* fgpix = (int32_t*)malloc(sizeof(int32_t) * (size_t)fg);
            if (!fgpix) { rc = FPCO_ENOMEM; break; }
            int r = 0;
            for (size_t p = 0; p < HW; ++p) if (m[p] != 0.0f) fgpix[r++] = (int32_t)p;
            for (int hi = 0; hi < hn && rc == FPCO_OK; ++hi) {
                float c2[4], d2[4];
                const int32_t two[2] = {0, 1};
                for (int k = 0; k < 2; ++k) {
                    int32_t px = 0;
                    for (int a = 0; a < FPC_SAMPLE_MAX_TRIES; ++a) {
                        px = fgpix[fpc_rand_index(seed, (uint32_t)bi, (uint32_t)hi, (uint32_t)(k + 2 * a), (uint32_t)fg)];
                        int kept = keep ? (keep[(size_t)bi * HW + px] != 0)
                                        : fpc_rand_keep(seed, (uint32_t)bi, (uint32_t)px, (uint32_t)fg, (uint32_t)max_num);
                        if (kept) break;
                    }
                    int y = px / W, x = px - y * W;
                    c2[2 * k] = (float)x; c2[2 * k + 1] = (float)y;
                    d2[2 * k] = v[(int64_t)y * vs_h + (int64_t)x * vs_w];
                    d2[2 * k + 1] = v[(int64_t)y * vs_h + (int64_t)x * vs_w + vs_c];
                }
                rc = fpco_generate_hypothesis(d2, c2, two, hyp + 2 * hi, 2, 1, 1);        /* :559 */
            }
            free(fgpix);
            if (rc != FPCO_OK) break;
        }
        if (out_hyp) memcpy(out_hyp + 2 * (size_t)bi * hn, hyp, sizeof(float) * 2 * (size_t)hn);

        /* :562-567 counts and arg-max; torch.max returns the first maximal index */
        int best = 0, best_cnt = -1;
        /* the hn x tn decisions are independent: OpenMP over the hypotheses when the library is built with it
         * (bench.py's all-cores baseline; fpco_set_threads(1) = the scalar port); the arg-max stays sequential */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 8) num_threads(fpco_get_threads())
#endif
        for (int hi = 0; hi < hn; ++hi) {
            int cnt = 0;
            float hx = hyp[2 * hi], hy = hyp[2 * hi + 1];
            for (int ti = 0; ti < tn; ++ti)
                cnt += fpco_pair_is_inlier(coords[2 * ti], coords[2 * ti + 1],
                                           direct[2 * ti], direct[2 * ti + 1], hx, hy, thresh);
            pair[2 * hi] = cnt;                              /* pair[] is free after generate_hypothesis */
        }
        for (int hi = 0; hi < hn; ++hi) {
            int cnt = pair[2 * hi];
            if (out_counts) out_counts[(size_t)bi * hn + hi] = cnt;
            if (cnt > best_cnt) { best_cnt = cnt; best = hi; }
        }
        /* :571-574 all_win_* start at 0 and update only on a strictly larger ratio */
        float wx = 0.0f, wy = 0.0f;
        if (best_cnt > 0) { wx = hyp[2 * best]; wy = hyp[2 * best + 1]; }
        if (out_win_idx) out_win_idx[bi] = best_cnt > 0 ? best : -1;
        if (out_win_count) out_win_count[bi] = best_cnt;

        /* :583-599 vote again with the winner, least squares over its inliers.
         * The reference sums in fp32 with torch's reduction order; the oracle sums
         * in fp64 (order-independent to ~1e-16), compared under a tolerance. */
        double a00 = 0, a01 = 0, a11 = 0, b0 = 0, b1 = 0;
        int inl = 0;
        for (int ti = 0; ti < tn; ++ti) {
            if (!fpco_pair_is_inlier(coords[2 * ti], coords[2 * ti + 1],
                                     direct[2 * ti], direct[2 * ti + 1], wx, wy, thresh))
                continue;
            ++inl;
            double nx = (double)direct[2 * ti + 1];      /* normal = (dy, -dx) :584-586 */
            double ny = -(double)direct[2 * ti];
            double bb = nx * (double)coords[2 * ti] + ny * (double)coords[2 * ti + 1];
            a00 += nx * nx; a01 += nx * ny; a11 += ny * ny;
            b0 += nx * bb; b1 += ny * bb;
        }
        if (out_inl_count) out_inl_count[bi] = inl;
        double x0, x1;
        fpco_solve2_sym(a00, a01, a11, b0, b1, &x0, &x1);
        oxy[0] = (float)x0;
        oxy[1] = (float)x1;
    }
    free(coords); free(direct); free(hyp); free(pair);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* Model.class_compression + gtf.class_compress                              */
/* F/lib/pose_regressor.py:445-457, F/lib/gpu_tensor_funcs.py:37-99.          */
/* mask_logits [B,C,HW]; quat [B,4(C-1),HW]; scales [B,3(C-1),HW];            */
/* xy [B,2(C-1),HW]; z [B,(C-1),HW].  cat_mask_in (nullable) overrides the    */
/* arg-max (gtf.class_compress takes cat_mask as an argument).                */
/* cat_mask = argmax_c( (x_c - max) - log(sum exp(x - max)) ), first maximal  */
/* index on ties (torch.argmax over LogSoftmax, pose_regressor.py:449).        */
int fpco_class_compress(const float* mask_logits, const float* quat, const float* scales,
                        const float* xy, const float* z, const int64_t* cat_mask_in,
                        int B, int C, int HW,
                        int64_t* cat_mask, float* oq, float* os, float* oxy, float* oz) {
    if (B < 0 || C < 2 || HW < 1) return FPCO_EINVAL;
    for (int b = 0; b < B; ++b)
        for (int p = 0; p < HW; ++p) {
            int64_t cls;
            if (cat_mask_in) {
                cls = cat_mask_in[(size_t)b * HW + p];
                if (cls < 0 || cls >= C) return FPCO_EINVAL;
            } else {
                const float* ml = mask_logits + (size_t)b * C * HW + p;
                float mx = ml[0];
                for (int c = 1; c < C; ++c) mx = fmaxf(mx, ml[(size_t)c * HW]);
                float s = 0.0f;
                for (int c = 0; c < C; ++c) s += expf(ml[(size_t)c * HW] - mx);
                float lse = logf(s);
                float bestv = (ml[0] - mx) - lse;
                cls = 0;
                for (int c = 1; c < C; ++c) {
                    float val = (ml[(size_t)c * HW] - mx) - lse;
                    if (val > bestv) { bestv = val; cls = c; }
                }
            }
            cat_mask[(size_t)b * HW + p] = cls;
            /* class k (1..C-1) selects channel group k-1 (torch.chunk, :67; [:,1:], :65);
             * background -> all zeros (where(mask, v, 0) summed over classes, :78-85). */
            float q[4] = {0, 0, 0, 0}, sc[3] = {0, 0, 0}, v[2] = {0, 0}, zz = 0.0f;
            if (cls > 0) {
                int g = (int)cls - 1;
                for (int a = 0; a < 4; ++a) q[a] = quat[((size_t)b * 4 * (C - 1) + 4 * g + a) * HW + p];
                for (int a = 0; a < 3; ++a) sc[a] = scales[((size_t)b * 3 * (C - 1) + 3 * g + a) * HW + p];
                for (int a = 0; a < 2; ++a) v[a] = xy[((size_t)b * 2 * (C - 1) + 2 * g + a) * HW + p];
                zz = z[((size_t)b * (C - 1) + g) * HW + p];
            }
            /* gtf.normalize (:37-50): x / where(norm != 0, norm, 1) */
            float nq = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
            if (nq == 0.0f) nq = 1.0f;
            float nv = sqrtf(v[0] * v[0] + v[1] * v[1]);
            if (nv == 0.0f) nv = 1.0f;
            for (int a = 0; a < 4; ++a) oq[((size_t)b * 4 + a) * HW + p] = q[a] / nq;
            for (int a = 0; a < 3; ++a) os[((size_t)b * 3 + a) * HW + p] = sc[a];
            for (int a = 0; a < 2; ++a) oxy[((size_t)b * 2 + a) * HW + p] = v[a] / nv;
            oz[(size_t)b * HW + p] = zz;
        }
    return FPCO_OK;
}

/* ------------------------------------------------------------------------ */
/* AggregationLayer.batchwise_break_segmentation_mask                         */
/* F/lib/aggregation_layer.py:36-59,160-183: scipy.ndimage.label with a       */
/* structuring element that is a plus-shape in the middle slice only, i.e.     */
/* 4-connectivity inside an image and nothing across the batch axis.  Labels   */
/* are 1..N in raster order of each component's first pixel, numbered          */
/* continuously over the batch.  fg: u8 [B,H,W] (cat_mask != 0).               */
static int32_t uf_find(int32_t* parent, int32_t x) {
    while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
    return x;
}

int fpco_cc_label(const uint8_t* fg, int B, int H, int W, int32_t* labels, int32_t* n_out) {
    if (B < 0 || H < 1 || W < 1) return FPCO_EINVAL;
    size_t HW = (size_t)H * W;
    int32_t* parent = (int32_t*)malloc(sizeof(int32_t) * HW);
    if (!parent) return FPCO_ENOMEM;
    int32_t next = 0;
    for (int b = 0; b < B; ++b) {
        const uint8_t* f = fg + (size_t)b * HW;
        int32_t* L = labels + (size_t)b * HW;
        /* union-find on linear indices; the root is always the minimum index */
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int32_t p = y * W + x;
                if (!f[p]) { parent[p] = -1; continue; }
                parent[p] = p;
                if (x > 0 && f[p - 1]) {
                    parent[p] = uf_find(parent, p - 1);
                }
                if (y > 0 && f[p - W]) {
                    int32_t r1 = uf_find(parent, p - W), r2 = uf_find(parent, p);
                    if (r1 < r2) parent[r2] = r1; else if (r2 < r1) parent[r1] = r2;
                }
            }
        /* roots in raster order get consecutive labels */
        for (size_t p = 0; p < HW; ++p) {
            if (!f[p]) { L[p] = 0; continue; }
            if (parent[p] == (int32_t)p) L[p] = ++next;
        }
        for (size_t p = 0; p < HW; ++p)
            if (f[p] && parent[p] != (int32_t)p) L[p] = L[uf_find(parent, (int32_t)p)];
    }
    free(parent);
    *n_out = next;
    return FPCO_OK;
}

/* ------------------------------------------------------------------------ */
/* AggregationLayer.forward, F/lib/aggregation_layer.py:61-158.               */
/* labels i32 [B,H,W] (1..N global), cat_mask i64 [B,H,W], categorical planes  */
/* quat [B,4,HW], scales [B,3,HW], xy [B,2,HW], z [B,HW].                      */
/* Outputs, instances in label order (= sample order then raster order):       */
/* class_ids i64[N] (smallest non-zero class inside the instance, :111-113),   */
/* sample_ids i64[N], inst_masks f32[N,HW], oq [N,4] (mean then re-normalised, */
/* :137-147), os [N,3] (mean), oz [N,1] (exp(mean), :142-143),                 */
/* oxy f32 [N,2,HW] (masked, un-averaged field, :150-151).                     */
/* Sums are fp64 here (reference: fp32 torch.sum), compared under tolerance.   */
int fpco_aggregate(const int32_t* labels, const int64_t* cat_mask,
                   const float* quat, const float* scales, const float* xy, const float* z,
                   int B, int H, int W, int N,
                   int64_t* class_ids, int64_t* sample_ids, float* inst_masks,
                   float* oq, float* os, float* oz, float* oxy) {
    if (B < 0 || N < 0) return FPCO_EINVAL;
    size_t HW = (size_t)H * W;
    double* sums = (double*)calloc((size_t)N * 8 + 1, sizeof(double));
    int64_t* cnt = (int64_t*)calloc((size_t)N + 1, sizeof(int64_t));
    if (!sums || !cnt) { free(sums); free(cnt); return FPCO_ENOMEM; }
    for (int i = 0; i < N; ++i) { class_ids[i] = 0; sample_ids[i] = -1; }
    if (inst_masks) memset(inst_masks, 0, sizeof(float) * (size_t)N * HW);
    if (oxy) memset(oxy, 0, sizeof(float) * (size_t)N * 2 * HW);
    for (int b = 0; b < B; ++b)
        for (size_t p = 0; p < HW; ++p) {
            int32_t l = labels[(size_t)b * HW + p];
            if (l <= 0) continue;
            if (l > N) { free(sums); free(cnt); return FPCO_EINVAL; }
            int i = l - 1;
            sample_ids[i] = b;
            int64_t c = cat_mask[(size_t)b * HW + p];
            if (c != 0 && (class_ids[i] == 0 || c < class_ids[i])) class_ids[i] = c;
            cnt[i] += 1;
            for (int a = 0; a < 4; ++a) sums[(size_t)i * 8 + a] += (double)quat[((size_t)b * 4 + a) * HW + p];
            for (int a = 0; a < 3; ++a) sums[(size_t)i * 8 + 4 + a] += (double)scales[((size_t)b * 3 + a) * HW + p];
            sums[(size_t)i * 8 + 7] += (double)z[(size_t)b * HW + p];
            if (inst_masks) inst_masks[(size_t)i * HW + p] = 1.0f;
            if (oxy) {
                oxy[((size_t)i * 2 + 0) * HW + p] = xy[((size_t)b * 2 + 0) * HW + p];
                oxy[((size_t)i * 2 + 1) * HW + p] = xy[((size_t)b * 2 + 1) * HW + p];
            }
        }
    for (int i = 0; i < N; ++i) {
        double c = (double)cnt[i];
        float q[4];
        for (int a = 0; a < 4; ++a) q[a] = (float)(sums[(size_t)i * 8 + a] / c);
        float nq = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        if (nq == 0.0f) nq = 1.0f;
        for (int a = 0; a < 4; ++a) oq[(size_t)i * 4 + a] = q[a] / nq;
        for (int a = 0; a < 3; ++a) os[(size_t)i * 3 + a] = (float)(sums[(size_t)i * 8 + 4 + a] / c);
        oz[i] = expf((float)(sums[(size_t)i * 8 + 7] / c));
    }
    free(sums); free(cnt);
    return FPCO_OK;
}

/* ------------------------------------------------------------------------ */
/* gtf.batchwise_get_RT + quats_2_rotation_matrix                             */
/* F/lib/gpu_tensor_funcs.py:204-235, 306-326.                                */
/* q [n,4] scalar-LAST, xy [n,2], z [n] (already exp'ed), kinv [3,3] row-major */
/* -> R [n,3,3], T [n,3], RT [n,4,4].                                          */
/* The reference forms RT = inverse([[inverse(R), T],[0 0 0 1]]) with two LU    */
/* inversions in fp32; for an orthonormal R that is [[R, -R T],[0 0 0 1]] up to */
/* rounding, which is the closed form used here (compared under tolerance).    */
int fpco_pose_rt(const float* q, const float* xy, const float* z, const float* kinv, int n,
                 float* R, float* T, float* RT) {
    if (n < 0) return FPCO_EINVAL;
    for (int i = 0; i < n; ++i) {
        float zz = z[i] / 1000.0f;
        float px = xy[2 * i] * zz, py = xy[2 * i + 1] * zz;
        float t[3];
        for (int r = 0; r < 3; ++r)
            t[r] = kinv[3 * r] * px + kinv[3 * r + 1] * py + kinv[3 * r + 2] * zz;
        float q1 = q[4 * i], q2 = q[4 * i + 1], q3 = q[4 * i + 2], q4 = q[4 * i + 3];
        float nrm = sqrtf(q1 * q1 + q2 * q2 + q3 * q3 + q4 * q4);
        if (!(nrm > 0.0f)) nrm = 1.0f;
        q1 /= nrm; q2 /= nrm; q3 /= nrm; q4 /= nrm;
        float a = q1 * q1, b = q2 * q2, c = q3 * q3, d = q4 * q4;
        /* M as written at :316-324, the function returns its transpose (:326) */
        float M[9] = { a - b - c + d, 2 * (q1 * q2 + q3 * q4), 2 * (q1 * q3 - q2 * q4),
                       2 * (q1 * q2 - q3 * q4), -a + b - c + d, 2 * (q2 * q3 + q1 * q4),
                       2 * (q1 * q3 + q2 * q4), 2 * (q2 * q3 - q1 * q4), -a - b + c + d };
        float* Ri = R + 9 * (size_t)i;
        for (int r = 0; r < 3; ++r)
            for (int cidx = 0; cidx < 3; ++cidx) Ri[3 * r + cidx] = M[3 * cidx + r];
        for (int r = 0; r < 3; ++r) T[3 * (size_t)i + r] = t[r];
        float* G = RT + 16 * (size_t)i;
        for (int r = 0; r < 3; ++r) {
            for (int cidx = 0; cidx < 3; ++cidx) G[4 * r + cidx] = Ri[3 * r + cidx];
            G[4 * r + 3] = -(Ri[3 * r] * t[0] + Ri[3 * r + 1] * t[1] + Ri[3 * r + 2] * t[2]);
        }
        G[12] = 0; G[13] = 0; G[14] = 0; G[15] = 1;
    }
    return FPCO_OK;
}

/* ---- matching: 2D IoU of every (mask1, mask2) pair ------------------------------------------
 * F/lib/gpu_tensor_funcs.py:386-409 (batchwise_get_2d_iou), the [n1,n2,H,W] expansion that
 * F/lib/matching.py:264-267 (batchwise_find_matches) calls per class.  logical_and / logical_or
 * treat every non-zero element (NaN included, -0.0 not) as true; the sums are int64 and
 * `intersection / union` is torch's true division of two int64 tensors: both converted to
 * float32, one IEEE division (0/0 = NaN for two empty masks).  inter / uni may be NULL. */
int fpco_mask_iou(const float* m1, int n1, const float* m2, int n2, int64_t hw, float* iou, int64_t* inter,
                  int64_t* uni) {
    if (n1 < 0 || n2 < 0 || hw < 0) return FPCO_EINVAL;
    for (int i = 0; i < n1; ++i)
        for (int j = 0; j < n2; ++j) {
            const float* a = m1 + (size_t)i * hw;
            const float* b = m2 + (size_t)j * hw;
            int64_t in = 0, un = 0;
            for (int64_t p = 0; p < hw; ++p) {
                int ta = a[p] != 0.0f, tb = b[p] != 0.0f;   /* NaN != 0 is true */
                in += ta & tb;
                un += ta | tb;
            }
            if (inter) inter[(size_t)i * n2 + j] = in;
            if (uni) uni[(size_t)i * n2 + j] = un;
            iou[(size_t)i * n2 + j] = (float)in / (float)un;
        }
    return FPCO_OK;
}
