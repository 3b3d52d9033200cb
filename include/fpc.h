/* fpc.h — C ABI of libfpc_hip.so: the MI355X (gfx950) implementation of
 * FastPoseCNN's per-frame post-network hot path.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only; no torch types.
 * Every pointer marked "device" must be device-accessible memory of the GPU
 * that is current on the calling thread; every function
 *   - allocates nothing and frees nothing (caller-owned buffers + workspace),
 *   - enqueues its kernels on `stream` (a hipStream_t, passed as void*; NULL is
 *     the default stream) and returns without synchronising,
 *   - returns FPC_OK or a negative FPC_E* code — it never exits the process
 *     (the reference's gpuAssert calls exit(), RV/src/cuda_common.h:17-26).
 *
 * Reference interfaces replaced (paths under /root/reference/source_code/FastPoseCNN,
 * RV = lib/ransac_voting_gpu_layer):
 *   fpc_generate_hypothesis      RV/src/ransac_voting.cpp:20-31  (+ kernel RV/src/ransac_voting_kernel.cu:11-86)
 *   fpc_voting_for_hypothesis    RV/src/ransac_voting.cpp:41-55  (+ kernel .cu:88-167)
 *   fpc_ransac_voting_v3         RV/ransac_voting_gpu.py:518-607 (ransac_voting_layer_v3 + b_inv :503-516)
 *   fpc_class_compress           lib/pose_regressor.py:445-457, lib/gpu_tensor_funcs.py:37-99
 *   fpc_cc_label                 lib/aggregation_layer.py:160-183 (cupyx / scipy ndimage.label)
 *   fpc_aggregate                lib/aggregation_layer.py:61-158
 *   fpc_pose_rt                  lib/gpu_tensor_funcs.py:204-253, 306-326
 *   fpc_pack_pose_records        (none: multi-GPU gather record)
 *   fpc_mask_iou                 lib/gpu_tensor_funcs.py:386-409 (batchwise_get_2d_iou), called by lib/matching.py:264-267
 *   fpc_net_*                    lib/pose_regressor.py:709-743 (+ segmentation_models_pytorch encoder/decoder/head)
 *   fpc_preprocess_u8            tools/dataset.py:249-262 (preprocessing_fn, transpose, / max|.|, img_as_float32)
 *   fpc_png_info / _decode / _decode_batch   tools/dataset.py:158-176 (skimage.io.imread / cv2.imread of *_color / *_mask / *_depth.png)
 *   fpc_pose_errors              lib/gpu_tensor_funcs.py:411-476, 486-547, 563-565 (degree error, 3-D IoU, offset error)
 *   fpc_post_network_backward    torch autograd over lib/aggregation_layer.py:119-156 + RV/ransac_voting_gpu.py:583-599
 *   fpc_vote_refine_backward     torch autograd over RV/ransac_voting_gpu.py:583-599
 *   fpc_class_compress_backward  torch autograd over lib/gpu_tensor_funcs.py:37-99
 *   fpc_mask_losses              lib/loss.py:26-98 (CE, CCE, Focal: forward sums and the combined logit gradient)
 *   fpc_lookahead_radam_step     lib/pose_regressor.py:417-423 (catalyst Lookahead(RAdam)), F/train.py gradient_clip_val,
 *                                lib/pose_regressor.py:341-415 (inf / NaN guard)
 * The Python-side bindings a maintainer would add are shown in INTEGRATION.md.
 */
#ifndef FPC_H_
#define FPC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPC_ABI_VERSION 11

#define FPC_OK 0
#define FPC_EINVAL (-1)      /* bad argument (shape, null pointer, ...) */
#define FPC_EWORKSPACE (-2)  /* workspace too small / misaligned */
#define FPC_ELAUNCH (-3)     /* HIP reported a launch error (hipGetLastError) */
#define FPC_EDEVICE (-4)     /* not running on a gfx950 device / no device */
#define FPC_EFORMAT (-5)     /* fpc_png_*: not a PNG this decoder reads (bad signature / CRC / zlib stream, Adam7, 1-4 bit samples) */

typedef void* fpc_stream_t;  /* hipStream_t */

int fpc_abi_version(void);
const char* fpc_error_string(int code);
/* Text of the last HIP error seen by this thread's failing call ("" if none). */
const char* fpc_last_hip_error(void);

/* ---- B1: the reference extension's two live entry points ------------------
 * direct f32 [tn,vn,2], coords f32 [tn,2] (x = column, y = row), idxs i32 [hn,vn,2]
 * -> hyp f32 [hn,vn,2].  hyp is fully written (zero where the pair is degenerate,
 * as the reference's zero-initialised result tensor).  All device, contiguous. */
int fpc_generate_hypothesis(const float* direct, const float* coords, const int32_t* idxs,
                            float* hyp, int tn, int vn, int hn, fpc_stream_t stream);

/* inliers u8 [hn,vn,tn] is caller-owned and only ever written with 1
 * (the caller pre-zeroes it, RV/ransac_voting_gpu.py:562). */
int fpc_voting_for_hypothesis(const float* direct, const float* coords, const float* hyp,
                              uint8_t* inliers, int tn, int vn, int hn, float inlier_thresh,
                              fpc_stream_t stream);

/* ---- B2: fused ransac_voting_layer_v3 for one keypoint channel ------------
 * mask    f32 [n,H,W] contiguous, foreground iff != 0
 * vertex  base pointer of the [n,H,W,(vn),2] view for the chosen keypoint, with
 *         ELEMENT strides vs_n, vs_h, vs_w, vs_c (the reference passes a permuted
 *         view of two planes, lib/hough_voting.py:51 — no copy is needed)
 * idxs    i32 [n,hn,2] injected pair indices, or NULL -> include/fpc_rng.h stream(seed)
 * keep    u8 [n,H,W] injected thinning selection (used only where fg > max_num), or NULL
 * out_xy  f32 [n,2]
 * optional diagnostics (NULL to skip): out_tn, out_win_idx, out_win_count, out_inl_count i32 [n];
 *         out_hyp f32 [n,hn,2]; out_counts i32 [n,hn] (the exact inlier count of every hypothesis — the path
 *         computes all of them; the diagnostics are copies).
 * n_dev   NULL, or DEVICE i32[1]: only the first min(n, *n_dev) instances are processed (output rows past it are
 *         left untouched).  Lets a caller size buffers by a capacity `n` and enqueue the vote behind
 *         fpc_cc_label without reading the instance count back to the host.
 * hn      1 .. 65536.
 * ws      device workspace of at least fpc_ransac_workspace_bytes(n,H,W,hn) bytes, 256-byte aligned.  Contents
 *         are scratch: nothing has to survive between calls and nothing has to be cleared before one (four stateless
 *         launches on `stream` — eight with the progressive count below —, no memset, no second stream: mask scan +
 *         compacted pixel list -> per-instance plan (hypotheses, work records) -> exact counts on the matrix cores ->
 *         winner + refinement; csrc/ransac.hip, csrc/vote_count.hip).
 * Results are those of the reference's exhaustive vote bit for bit (counts, winner, inlier set); the refinement solves
 * the 2x2 normal equations in fp64 like b_inv (RV/ransac_voting_gpu.py:503-516: inverse, pseudo-inverse when singular),
 * and also takes the pseudo-inverse when det <= 1e-12 trace^2 (conditioning beyond fp64's reach for f32 votes; torch
 * raises only on an exactly singular LU).  An instance above max_num without an injected `keep` / `idxs` draws its
 * pairs by rejection over the kept pixels (include/fpc_rng.h, FPC_SAMPLE_MAX_TRIES). */
size_t fpc_ransac_workspace_bytes(int n, int H, int W, int hn);
int fpc_ransac_voting_v3(const float* mask, const float* vertex,
                         int64_t vs_n, int64_t vs_h, int64_t vs_w, int64_t vs_c,
                         int n, const int32_t* n_dev, int H, int W, int hn,
                         const int32_t* idxs, const uint8_t* keep, uint64_t seed,
                         float inlier_thresh, int min_num, int max_num,
                         float* out_xy, int32_t* out_tn, int32_t* out_win_idx,
                         int32_t* out_win_count, int32_t* out_inl_count,
                         float* out_hyp, int32_t* out_counts, double* out_refine,
                         void* ws, size_t ws_bytes, fpc_stream_t stream);
/* The same with the foreground ALSO given as bit words (fpc_aggregate_bits' inst_bits: u64 [n][fpc_mask_bits_words(H,W)],
 * 8-byte aligned): when mask_bits is not NULL the f32 `mask` planes are not read (mask may then be NULL) — two thirds of
 * the mask scan's bytes at a typical foreground share.  The caller guarantees bit == (mask != 0); results are identical. */
int fpc_ransac_voting_v3_bits(const float* mask, const uint64_t* mask_bits, const float* vertex,
                              int64_t vs_n, int64_t vs_h, int64_t vs_w, int64_t vs_c,
                              int n, const int32_t* n_dev, int H, int W, int hn,
                              const int32_t* idxs, const uint8_t* keep, uint64_t seed,
                              float inlier_thresh, int min_num, int max_num,
                              float* out_xy, int32_t* out_tn, int32_t* out_win_idx,
                              int32_t* out_win_count, int32_t* out_inl_count,
                              float* out_hyp, int32_t* out_counts, double* out_refine,
                              void* ws, size_t ws_bytes, fpc_stream_t stream);

/* The same with the RT assembly of fpc_pose_rt appended (round 6: one launch less per frame — the post-network path at batch 1 is a
 * chain of launch latencies): when the six pose pointers are given (all or none), the workgroup of k_vote_final that writes an
 * instance's out_xy also writes R [n][9], T [n][3], RT [n][16] from (out_xy, pose_q [n][4] scalar-last, pose_z [n], pose_kinv [9]) —
 * the same function, the same bits as fpc_pose_rt on the same operands (F/lib/gpu_tensor_funcs.py:204-235 after
 * F/lib/hough_voting.py:33-63).  Rows at or beyond *n_dev are not written. */
int fpc_ransac_voting_v3_pose(const float* mask, const uint64_t* mask_bits, const float* vertex,
                              int64_t vs_n, int64_t vs_h, int64_t vs_w, int64_t vs_c,
                              int n, const int32_t* n_dev, int H, int W, int hn,
                              const int32_t* idxs, const uint8_t* keep, uint64_t seed,
                              float inlier_thresh, int min_num, int max_num,
                              float* out_xy, int32_t* out_tn, int32_t* out_win_idx,
                              int32_t* out_win_count, int32_t* out_inl_count,
                              float* out_hyp, int32_t* out_counts, double* out_refine,
                              const float* pose_q, const float* pose_z, const float* pose_kinv,
                              float* pose_R, float* pose_T, float* pose_RT,
                              void* ws, size_t ws_bytes, fpc_stream_t stream);

/* The progressive count (OFF by default).  Unless the caller asks for out_counts, only the WINNER of the vote is an output
 * (RV/ransac_voting_gpu.py:566-574: arg-max of the inlier counts, first maximum).  With fpc_vote_set_prune(1, ...) the count
 * runs in passes over disjoint sets of each instance's pixels; between two passes every hypothesis h with
 *     count_so_far(h) + valid pixels not yet counted  <  L      (or == L with an index above the leader's)
 * is dropped, L being the EXACT final count of the current leader (csrc/ransac.hip: k_vote_lead).  The winner, its count,
 * its inlier set and the refined centre are those of the exhaustive vote, bit for bit; out_win_idx / out_win_count /
 * out_inl_count stay available.  With out_counts the exhaustive count runs.  Measured on the 32-frame batch at hn = 1000:
 * 0.55 of the pairs, but 326 us against the exhaustive 283 us (three count launches and two k_vote_lead launches cost more
 * than the pairs they save: profiles/r05_vote_prune.md) - hence off unless asked for.
 * fpc_vote_set_prune: mode 0 = never (default), 1 = whenever out_counts is NULL and the sizes fit (n <= 1024 instances,
 * <= 8.4 M pixels).  npass = 0 keeps the schedule; else 2..4 passes, cum16[npass - 1] = increasing 16ths of an
 * instance's count units finished after each pass but the last (default 3 passes: 5, 10).  Process-wide.
 * fpc_vote_prune_info: copy of the last call's per-instance record out of its workspace -> out i32 [n,8] (device):
 * {-, count units, alive hypotheses entering the last pass, their 32-wide tiles, last leader, its final count L,
 * valid pixels that were still uncounted at the last decision, -}. */
int fpc_vote_set_prune(int mode, int npass, const int32_t* cum16);
int fpc_vote_prune_info(const void* ws, size_t ws_bytes, int n, int H, int W, int hn, int32_t* out, fpc_stream_t stream);

/* ---- class compression ------------------------------------------------------
 * mask_logits f32 [B,C,HW]; quat [B,4(C-1),HW]; scales [B,3(C-1),HW]; xy [B,2(C-1),HW];
 * z [B,(C-1),HW]; cat_mask_in i64 [B,HW] or NULL (NULL: arg-max of log-softmax, first
 * maximal index on ties; mask_logits may be NULL when cat_mask_in is given).
 * -> cat_mask i64 [B,HW], oq [B,4,HW], os [B,3,HW], oxy [B,2,HW], oz [B,HW]. */
int fpc_class_compress(const float* mask_logits, const float* quat, const float* scales,
                       const float* xy, const float* z, const int64_t* cat_mask_in,
                       int B, int C, int HW,
                       int64_t* cat_mask, float* oq, float* os, float* oxy, float* oz,
                       fpc_stream_t stream);
/* The same, also writing the foreground (cat_mask != 0) as bit words for fpc_cc_label_bits: fg_bits u64
 * [B][fpc_mask_bits_words-style stride = ceil(HW / 4096) * 64], bit j of word w = pixel 64 w + j; words past ceil(HW / 64)
 * are not written.  NULL fg_bits = fpc_class_compress. */
int fpc_class_compress_bits(const float* mask_logits, const float* quat, const float* scales,
                            const float* xy, const float* z, const int64_t* cat_mask_in,
                            int B, int C, int HW,
                            int64_t* cat_mask, float* oq, float* os, float* oxy, float* oz,
                            uint64_t* fg_bits, fpc_stream_t stream);


/* ---- connected components ---------------------------------------------------
 * cat_mask i64 [B,H,W]; foreground iff != 0; 4-connectivity inside an image, none
 * across images.  labels i32 [B,H,W]: 0 background, 1..N numbered in raster order of
 * each component's first pixel, continuing across the batch (scipy.ndimage.label order).
 * n_out: DEVICE i32[1] receiving N.  root_pix: DEVICE i32 [cap] (nullable) receiving the
 * linear index (over B*H*W) of each component's first pixel for labels 1..min(N,cap). */
size_t fpc_cc_workspace_bytes(int B, int H, int W);
int fpc_cc_label(const int64_t* cat_mask, int B, int H, int W,
                 int32_t* labels, int32_t* n_out, int32_t* root_pix, int cap,
                 void* ws, size_t ws_bytes, fpc_stream_t stream);
/* The same labelling from the foreground as BIT WORDS: fg_bits u64 [B][fpc_mask_bits_words(H, W)], bit j of word w = pixel
 * 64 w + j of the image, zero past H W (8-byte aligned).  No reference counterpart: the reference hands cupy an f32 copy of
 * the mask (F/lib/aggregation_layer.py:165); here the class compression (fpc_class_compress_bits, the network engine) writes
 * the words beside the i64 mask and the labelling never reads the mask itself — 1/64 of its bytes, two launches.  Available
 * when fpc_cc_bits_supported(B, H, W) (rows on word boundaries: W % 64 == 0, and an image whose words fit one workgroup's LDS:
 * up to ~630 000 pixels); fpc_cc_label takes the same path by itself on such shapes after converting the mask (fpc_fg_bits). */
int fpc_cc_bits_supported(int B, int H, int W);
int fpc_fg_bits(const int64_t* cat_mask, int B, int H, int W, uint64_t* fg_bits, fpc_stream_t stream);
int fpc_cc_label_bits(const uint64_t* fg_bits, int B, int H, int W, int32_t* labels, int32_t* n_out,
                      int32_t* root_pix, int cap, void* ws, size_t ws_bytes, fpc_stream_t stream);


/* ---- aggregation ------------------------------------------------------------
 * labels i32 [B,H,W] from fpc_cc_label, N instances (host value: the count read back by the caller,
 * or a capacity when n_dev — DEVICE i32[1], fpc_cc_label's n_out — is given: then only the first
 * min(N, *n_dev) instances are produced and the remaining rows are left untouched), categorical planes quat [B,4,HW], scales [B,3,HW], xy [B,2,HW], z [B,HW].
 * -> class_ids i64 [N], sample_ids i64 [N], inst_masks f32 [N,HW] (nullable),
 *    oq [N,4], os [N,3], oz [N], oxy f32 [N,2,HW] (nullable), out_stats f32 [N,2] (nullable: pixel count and the norm
 *    of the mean quaternion before normalisation — what fpc_post_network_backward's table is built from). */
size_t fpc_aggregate_workspace_bytes(int N);
int fpc_aggregate(const int32_t* labels, const int64_t* cat_mask,
                  const float* quat, const float* scales, const float* xy, const float* z,
                  int B, int H, int W, int N, const int32_t* n_dev,
                  int64_t* class_ids, int64_t* sample_ids, float* inst_masks,
                  float* oq, float* os, float* oz, float* oxy, float* out_stats,
                  void* ws, size_t ws_bytes, fpc_stream_t stream);
/* The same, and — when inst_bits is not NULL — every instance's foreground also as bit words u64
 * [N][fpc_mask_bits_words(H, W)] (bit j of word w = pixel 64 w + j, zero past H W; whole 4096-pixel chunks): 1/32 of
 * the f32 mask plane, which fpc_ransac_voting_v3_bits reads INSTEAD of it.  No reference counterpart: the reference's
 * voting re-reads the f32 masks its aggregation has just written (lib/hough_voting.py:41-63).
 * root_pix (nullable): fpc_cc_label's root_pix with at least N valid entries (its cap >= N).  With it the image of an
 * instance is known without the accumulation, and accumulation and planes run as ONE launch. */
size_t fpc_mask_bits_words(int H, int W);
int fpc_aggregate_bits(const int32_t* labels, const int64_t* cat_mask,
                       const float* quat, const float* scales, const float* xy, const float* z,
                       int B, int H, int W, int N, const int32_t* n_dev,
                       int64_t* class_ids, int64_t* sample_ids, float* inst_masks,
                       float* oq, float* os, float* oz, float* oxy, float* out_stats, uint64_t* inst_bits,
                       const int32_t* root_pix, void* ws, size_t ws_bytes, fpc_stream_t stream);

/* ---- pose assembly ----------------------------------------------------------
 * q f32 [n,4] scalar-last, xy [n,2], z [n], kinv f32 [9] row-major (device)
 * -> R [n,9], T [n,3], RT [n,16]. */
int fpc_pose_rt(const float* q, const float* xy, const float* z, const float* kinv, int n,
                float* R, float* T, float* RT, fpc_stream_t stream);

/* ---- multi-GPU pose records ---------------------------------------------------
 * No reference counterpart (its evaluate / inference scripts are single-GPU, evaluate.py:90,127): one rank's
 * per-instance results as fixed-width records for ONE all-gather (fastposecnn_amd/parallel.py).
 * out f32 [capacity+1][40]: row 0 = {n as int32 bits, 0...}; row 1+i = {sample_id + sample_offset, class_id (int32
 * bits), quaternion[4], scales[3], xy[2], z[1], R[9], T[3], RT[16]}; rows past n are zero.  n <= capacity. */
int fpc_pack_pose_records(const int64_t* sample_ids, const int64_t* class_ids, const float* q, const float* scales,
                          const float* xy, const float* z, const float* R, const float* T, const float* RT, int n,
                          int sample_offset, int capacity, float* out, fpc_stream_t stream);

/* ---- matching: 2D IoU of every (mask1, mask2) pair ----------------------------
 * lib/gpu_tensor_funcs.py:386-409 batchwise_get_2d_iou (the [n1,n2,H,W] logical_and / logical_or expansion
 * of lib/matching.py:264-267).  masks1 [n1,hw], masks2 [n2,hw]: contiguous, elem_size 4 (f32: non-zero
 * incl. NaN = set, -0.0 = clear) or 1 (u8 / bool).  iou f32 [n1,n2] = (float)|a & b| / (float)|a | b|
 * (NaN for two empty masks, as torch's 0/0); inter / uni i32 [n1,n2] optional (NULL).  n1 == 0 or n2 == 0: no-op. */
size_t fpc_mask_iou_workspace_bytes(int n1, int n2, int64_t hw);
int fpc_mask_iou(const void* masks1, int n1, const void* masks2, int n2, int64_t hw, int elem_size,
                 float* iou, int32_t* inter, int32_t* uni, void* ws, size_t ws_bytes, fpc_stream_t stream);

/* ---- input side: colour frame -> network tensor --------------------------------
 * tools/dataset.py:249-262 on the device.  img_hwc u8 [B,H,W,3] (device, 16-byte aligned; RGB as skimage.io.imread
 * returns it), mean3 / std3 HOST doubles (smp's preprocessing parameters of the encoder: imagenet 0.485 0.456 0.406 /
 * 0.229 0.224 0.225), input_range_01 != 0: smp's rule "x / 255 when x.max() > 1" applies.
 * -> out f32 [B,3,H,W] = float32( ((x [/ 255]) - mean) / std / max|.| ), all in IEEE double with one final rounding:
 * bit-identical to the reference's numpy chain.  For B > 1: H*W % 4 == 0 and (3*H*W) % 16 == 0. */
size_t fpc_preprocess_workspace_bytes(int B);
int fpc_preprocess_u8(const uint8_t* img_hwc, int B, int H, int W, const double* mean3, const double* std3,
                      int input_range_01, float* out_nchw, void* ws, size_t ws_bytes, fpc_stream_t stream);

/* ---- frame decoding, HOST side (tools/dataset.py:158-176: skimage.io.imread / cv2.imread of the NOCS *_color.png,
 * *_mask.png, *_depth.png) -------------------------------------------------------------------------------------------
 * PNG (ISO/IEC 15948) through zlib: colour types 0 / 2 / 3 / 4 / 6, 8 or 16 bits per sample, non-interlaced; anything
 * else (and any damaged file: signature, chunk CRC, zlib stream, filter type) is FPC_EFORMAT.  Host pointers, no stream.
 * fpc_png_info: out5 = {width, height, bit depth, colour type, channels of the mode-0 output}.
 * fpc_png_decode mode 0: the samples as stored, [H, W, C] u8 — or u16 in host byte order for 16-bit files; a palette is
 *   expanded to RGB8 — exactly what imread returns; mode 3: RGB8 whatever the file holds (grey replicated, alpha
 *   dropped, 16-bit samples by their high byte).  out_bytes is checked.
 * fpc_png_decode_batch: n files of ONE size H x W -> out u8 [n, H, W, 3] (mode 3), decoded by `threads` host threads
 *   (the calling thread included); out is typically FrameUploader's pinned staging slot. */
int fpc_png_info(const uint8_t* data, size_t nbytes, int32_t* out5);
int fpc_png_decode(const uint8_t* data, size_t nbytes, void* out, size_t out_bytes, int mode);
int fpc_png_decode_batch(const uint8_t* const* datas, const size_t* sizes, int n, uint8_t* out, int H, int W, int threads);

/* ---- evaluation maths on matched pairs (lib/gpu_tensor_funcs.py:411-476, 486-547, 563-565) ----------------------
 * One launch for n (ground truth, prediction) pairs; every output is optional (NULL skips it and its inputs).
 * out_degree f64 [n]: get_raw_quat_distance (computed in f32) where symmetric_ids[i] == 0 or symmetric_ids is NULL,
 *   else get_symmetric_quat_distance over the nrot rotations rot f32 [nrot,4] (quat_symmetric_tf's table), in f64.
 *   Pair order is the input order (the Python wrapper applies the reference's non-symmetric-first concatenation).
 * out_iou3d f32 [n]: get_asymmetric_3d_iou(RT1, RT2, scales1, scales2), RT f32 [n,4,4], scales f32 [n,3].
 * out_offset f32 [n]: |T1 - T2| * 10, T f32 [n,3]. */
int fpc_pose_errors(const float* q0, const float* q1, const int64_t* symmetric_ids, const float* rot, int nrot,
                    const float* RT1, const float* RT2, const float* scales1, const float* scales2,
                    const float* T1, const float* T2, int n, double* out_degree, float* out_iou3d, float* out_offset,
                    fpc_stream_t stream);

/* ---- training: backward of the post-network path, optimiser step ----------------
 * fpc_post_network_backward: labels i32 [B,H,W] (fpc_cc_label), cat_xy f32 [B,2,H,W] (the categorical vote planes the
 * forward aggregated), table f64 [N,16] per instance: [0..3] dL/d(pixel quaternion), [4..6] dL/d(pixel scales),
 * [7] dL/d(pixel z) (the instance gradients chained through the mean / normalise / exp and divided by the pixel count),
 * [8..9] lam = (sum n n^T)^-1 dL/dxy, [10..11] the refined xy, [12..13] the winning hypothesis, [14] foreground count,
 * [15] != 0: the instance voted.  inlier_thresh / max_num / seed / keep: as in the forward fpc_ransac_voting_v3 call.
 * -> g_q [B,4,H,W], g_s [B,3,H,W], g_xy [B,2,H,W], g_z [B,H,W]: every element written.
 * fpc_vote_refine_backward: the vote term alone on the forward's mask [n,H,W] / vertex planes; table f64 [n,8]:
 * lam(2), refined xy(2), winner(2), foreground count, active.  -> g_vertex f32 [n,2,H,W].
 * fpc_ransac_voting_v3's out_refine f64 [n,8] = winner(2), a00, a01, a11, b0, b1, inliers supplies the normal equations.
 * fpc_class_compress_backward: go_* gradients of the four categorical outputs (NULL = zero) -> dense gradients of the four
 * logit tensors (layouts of fpc_class_compress). */
int fpc_post_network_backward(const int32_t* labels, const float* cat_xy, int B, int H, int W, int N,
                              const int32_t* n_dev, const double* table, float inlier_thresh, int max_num,
                              uint64_t seed, const uint8_t* keep, float* g_q, float* g_s, float* g_xy, float* g_z,
                              fpc_stream_t stream);
int fpc_vote_refine_backward(const float* mask, const float* vertex, int64_t vs_n, int64_t vs_h, int64_t vs_w,
                             int64_t vs_c, int n, int H, int W, const double* table, float inlier_thresh, int max_num,
                             uint64_t seed, const uint8_t* keep, float* g_vertex, fpc_stream_t stream);
int fpc_class_compress_backward(const int64_t* cat_mask, const float* quat, const float* xy, const float* go_q,
                                const float* go_s, const float* go_xy, const float* go_z, int B, int C, int HW,
                                float* g_q, float* g_s, float* g_xy, float* g_z, fpc_stream_t stream);
/* The three mask losses of lib/loss.py:26-98 (CE, CCE, Focal on log-softmax outputs) on logits f32 [B,C,HW] and target
 * i64 [B,HW], one pass.  grad == NULL: forward, sums6 f64 [6] (caller zeroes) += CE sum, CE count, CCE sum, CCE count,
 * Focal sum, Focal count (each loss = sum / count).  grad != NULL: backward, grad f32 [B,C,HW] = d/dlogits of
 * w3[0] * (CE sum) + w3[1] * (CCE sum) + w3[2] * (Focal sum), w3 DEVICE f32 [3] (upstream gradient / count). */
int fpc_mask_losses(const float* logits, const int64_t* target, int B, int C, int HW, int64_t ignore_ce,
                    int64_t ignore_cce, float alpha, float gamma, double* sums6, const float* w3, float* grad,
                    fpc_stream_t stream);
/* out2 f64 [2] (caller zeroes): [0] += sum g^2, [1] += 1 when a non-finite element was seen.  g 16-byte aligned. */
int fpc_grad_sumsq(const float* g, size_t n, double* out2, fpc_stream_t stream);
/* One Lookahead(RAdam) step on a flat f32 shard (p, g, m, v, slow: n elements each; step counts from 1).
 * ctl (device f32[2] or NULL): [0] multiplies every gradient (clip coefficient / world size); [1] != 0 (the inf / NaN guard):
 * the step runs with a ZERO gradient, as the reference's zero_grad() + optimizer.step() does. */
int fpc_lookahead_radam_step(float* p, const float* g, float* m, float* v, float* slow, size_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int64_t step, int la_k, float la_alpha,
                             const float* ctl, fpc_stream_t stream);

/* ---- backbone engine ----------------------------------------------------------
 * PoseRegressor.pure_model_forward + Model.class_compression for inference
 * (lib/pose_regressor.py:709-743, 445-457; encoder / FPN decoder / head graph of
 * segmentation_models_pytorch as built at :608-666): ResNet-18/34 encoder, four FPN decoders,
 * four 1x1 heads, x4 bilinear upsample, xyz -> xy/z split, class compression.  f32 throughout
 * (f32 matrix-core FMAs).  BatchNorm uses running statistics and Dropout2d is the identity
 * (eval mode).
 *
 * fpc_net_create     host-side plan for a fixed (encoder, classes, B, H, W); H, W multiples of 32.
 * fpc_net_param_*    the parameter tensors the plan needs, by state-dict name (smp naming,
 *                    e.g. "encoder.layer1.0.conv1.weight", "mask_decoder.p4.skip_conv.bias",
 *                    "rotation_decoder.seg_blocks.0.block.1.block.1.weight", "scales_head.0.weight"),
 *                    each a contiguous f32 tensor in torch's own layout (OIHW weights).
 * fpc_net_load_params  repacks the weights into the workspace (OHWI, padded) and folds BatchNorm.
 *                    `params[i]` (device, 16-byte aligned) must stay valid and unchanged for as
 *                    long as the plan is used: GroupNorm / bias / head parameters are read in place.
 *                    ws: device, 256-byte aligned, >= fpc_net_workspace_bytes(); owned by the caller,
 *                    must outlive the plan's use; holds packed weights AND activations.
 * fpc_net_forward    x f32 [B,3,H,W] (NCHW) -> full-resolution logits (NCHW planes; pass all five
 *                    or all NULL to skip materialising them) and the categorical outputs:
 *                    logits_mask [B,C,H,W], logits_quat [B,4(C-1),H,W], logits_scales [B,3(C-1),H,W],
 *                    logits_xy [B,2(C-1),H,W], logits_z [B,(C-1),H,W];
 *                    cat_mask i64 [B,H,W], cq [B,4,H,W], cs [B,3,H,W], cxy [B,2,H,W], cz [B,H,W]. */
typedef struct fpc_net fpc_net_t;
int fpc_net_create(const char* encoder, int classes, int B, int H, int W, fpc_net_t** out);
void fpc_net_destroy(fpc_net_t* net);
int fpc_net_param_count(const fpc_net_t* net);
const char* fpc_net_param_name(const fpc_net_t* net, int i);
int64_t fpc_net_param_numel(const fpc_net_t* net, int i);
size_t fpc_net_workspace_bytes(const fpc_net_t* net);
int fpc_net_load_params(fpc_net_t* net, const float* const* params, int count, void* ws, size_t ws_bytes,
                        fpc_stream_t stream);
int fpc_net_forward(fpc_net_t* net, const float* x, float* logits_mask, float* logits_quat,
                    float* logits_scales, float* logits_xy, float* logits_z, int64_t* cat_mask,
                    float* cq, float* cs, float* cxy, float* cz, fpc_stream_t stream);
/* The same, also writing the foreground of cat_mask as bit words (see fpc_class_compress_bits; the plan's W must be a
 * multiple of 64 when fg_bits is not NULL). */
int fpc_net_forward_bits(fpc_net_t* net, const float* x, float* logits_mask, float* logits_quat,
                         float* logits_scales, float* logits_xy, float* logits_z, int64_t* cat_mask,
                         float* cq, float* cs, float* cxy, float* cz, uint64_t* fg_bits, fpc_stream_t stream);

/* Autotuning: after fpc_net_autotune_next the NEXT fpc_net_forward times every candidate tiling
 * (block tile, split-K factor) of every convolution on the device, keeps the fastest, and is itself
 * a valid forward; it synchronises the stream, so it must not be captured into a graph.
 * fpc_net_conv_plan reports the tiling in use for convolution i: out5 = bm, bn, nsplit, Cout, K. */
int fpc_net_autotune_next(fpc_net_t* net, int mode /* 0: minimise each conv's latency; 1: latency x sqrt(share of
                                                      the chip its grid occupies) — for several frames in flight
                                                      (what FrameStreamer(tune_mode=1) and bench.py's stream use);
                                                      2: latency x share (the launch's CU-time) */);
/* Split-precision matrix products (library default 0; the Python front end turns it on unless
 * HPARAM.ENGINE_SPLIT_PRECISION is False).  1: autotuning may replace the f32 matrix instructions of a direct OR a
 * Winograd convolution by the exact three-way bf16 split of both operands (x = x1 + x2 + x3, each piece 8 significant
 * bits) and the six partial products with i + j <= 4 on v_mfma_f32_32x32x16_bf16, accumulated in f32: every kept product
 * is exact, the dropped ones are < 2^-23 of the term, so the result has f32-level accuracy (2.4e-7 of max|ref| against
 * float64, like another summation order) without being bit-identical to the f32 product chain.  Set before
 * fpc_net_autotune_next.  fpc_net_conv_plan reports -5 / -6 / -7 for a split-precision Winograd site (8 waves / 128 channels per
 * workgroup / four waves of 512 registers).
 * 2 (round 6): additionally the fp16 x 2 Winograd form (csrc/wino_h2.hip, reported as -8): two fp16 pieces per operand, all four
 * piece products in two matrix instructions per 8 channels.  Weights are scaled by a power of two on the device; a transformed ACTIVATION v is
 * represented to 2^-22 |v| while |v| >= 2^-3, to 2^-25 absolute below, and saturates beyond 1.3e5 — f32-level accuracy for
 * activations of ordinary scale (the tests hold it to the bars of every other form), NOT for tensors of tiny or huge values.  The
 * 3: additionally (and, where Cin is a multiple of 16, INSTEAD of -8) its three-product form (csrc/wino_h3.hip, reported as -9):
 * h1 g1 + h2 g1 + h1 g2 in three matrix instructions per 16 channels; the dropped h2 g2 is <= 2^-22 of the term — the size of the
 * two terms every two-piece form drops — and the same tests hold it to the same bars.
 * The Python front end uses 3 unless HPARAM.ENGINE_SPLIT_F16_3P (then 2) or HPARAM.ENGINE_SPLIT_F16 (then 1) is False. */
int fpc_net_set_split_precision(fpc_net_t* net, int on);
/* HIP graph replay (default 0).  1: after autotuning, the frame-invariant launches of fpc_net_forward (everything
 * between the image conversion and the final upsample / class compression, ~57 kernels on the plan's workspace) are
 * captured once on the caller's stream and replayed with one hipGraphLaunch per frame. */
int fpc_net_set_graph(fpc_net_t* net, int on);
int fpc_net_conv_count(const fpc_net_t* net);
int fpc_net_conv_plan(const fpc_net_t* net, int i, int* out5);
/* Copies the convolution plans (tilings, split-K factors, kernel forms) of `src` into `dst`: same encoder, classes and frame
 * size, any batch sizes — a small batch then runs on the kernels a larger one was autotuned to (tests/test_gpu_net.py: the
 * headline configuration's plan set against float64 on two frames).  Plans whose split-K partials do not fit dst keep dst's own. */
int fpc_net_copy_plans(fpc_net_t* dst, const fpc_net_t* src);
/* Puts every 3x3 / stride-1 site on Winograd form `form` (1..8: fpc_conv2d's -form; 6 only where Cout % 128 == 0; other sites keep
 * their plan) — tests hold the whole network on ONE form (8: every eligible product on fp16 x 2 pieces) to the float64 bars.
 * Returns the number of sites changed or a negative code. */
int fpc_net_force_winograd(fpc_net_t* net, int form);
/* FLOP of one forward over the batch under the current plans: out3 = {2 x MACs of the direct convolutions (what the
 * reference's cuDNN path executes), multiply-add FLOP the plans execute (Winograd sites: / 2.25), Winograd share}. */
int fpc_net_flops(const fpc_net_t* net, double* out3);
/* Intermediate activations (NHWC f32 inside the workspace) for tests: "stem", "pool", "c2".."c5",
 * "d<k>.p5".."d<k>.p2", "d<k>.seg<i>" (pre-GroupNorm conv outputs), "d<k>.low" (low-res logits). */
int fpc_net_tensor(const fpc_net_t* net, const char* name, const float** ptr, int* H, int* W, int* C);

/* Stand-alone convolution on the engine's implicit-GEMM kernel (tests / micro-benchmarks).
 * in: any element strides (sb, sh, sw, sc); w_oihw torch layout; out NHWC [B,Ho,Wo,Cout];
 * optional per-channel scale / shift, residual (as out), nearest-x2 `up` [B,Ho/2,Wo/2,Cout], ReLU,
 * GroupNorm partials gn_part [B][P32][Cout][2]; bm/bn/nsplit = 0 -> chosen by the planner.
 * `nsplit` also selects the engine's other kernels for tests: -1..-5 Winograd forms (-5 split precision), 100 + k split-K
 * summed by a second launch, 1000 + k split-precision (bf16 x 3) products, 2000 + parts the pixel-resident FPN lateral
 * product (1x1, Cin 64 / 128, bias + `up` epilogue), 3000 the weight-resident 7x7 / s2 stem (Cin = 4: NHWC4 image). */
size_t fpc_conv2d_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int Kh, int Kw);
/* the exact need of one request (same bm / bn / nsplit as the fpc_conv2d call): <= the bound above, which reserves 32
 * split-K slices of the whole output; fpc_conv2d accepts either size */
size_t fpc_conv2d_workspace_bytes_for(int B, int Ho, int Wo, int Cin, int Cout, int Kh, int Kw, int bm, int bn, int nsplit);
int fpc_conv2d_plan(int B, int Ho, int Wo, int Cin, int Cout, int Kh, int Kw, int bm, int bn, int nsplit,
                    int* out4 /* bm, bn, nsplit, P32 */);
int fpc_conv2d(const float* in, int64_t sb, int64_t sh, int64_t sw, int64_t sc, const float* w_oihw,
               const float* scale, const float* shift, const float* res, const float* up, float* out,
               float* gn_part, int B, int Hi, int Wi, int Cin, int Cout, int Kh, int Kw, int stride, int pad,
               int relu, int bm, int bn, int nsplit, void* ws, size_t ws_bytes, fpc_stream_t stream);

/* Weight gradient of a convolution for the training step (BASELINE.json configs[4]; the reference leaves the
 * convolutions' backward to cuDNN through autograd, F/lib/pose_regressor.py:709-743 under Lightning's backward).
 * x: input [B,Hi,Wi,Cin] channel-last (channel stride 1; element strides sb, sh, sw multiples of 4; 16-byte aligned),
 * dy: output gradient NHWC contiguous [B,Ho,Wo,Cout]; dw: OIHW contiguous [Cout,Cin,Kh,Kw], overwritten.
 * Cin % 64 == 0 and Cout % 4 == 0 are required (FPC_EINVAL otherwise); any stride / padding.  Split over the pixels,
 * summed in a fixed order: deterministic.  The data gradient of a stride-1 convolution is fpc_conv2d on the flipped,
 * transposed weights (fastposecnn_amd/lib/train_conv.py). */
size_t fpc_conv2d_wgrad_workspace_bytes(int B, int Ho, int Wo, int Cin, int Cout, int Kh, int Kw);
int fpc_conv2d_wgrad(const float* x, int64_t sb, int64_t sh, int64_t sw, const float* dy, float* dw, int B, int Hi,
                     int Wi, int Cin, int Cout, int Kh, int Kw, int stride, int pad, void* ws, size_t ws_bytes,
                     fpc_stream_t stream);
/* The same sums with split-precision matrix products (see "split precision" above: three bf16 pieces per f32 operand,
 * the six leading piece products accumulated in f32), about twice the f32 matrix rate.  Same workspace and shape rules,
 * deterministic; differs from fpc_conv2d_wgrad in the last bits only (ABI 8). */
int fpc_conv2d_wgrad_split(const float* x, int64_t sb, int64_t sh, int64_t sw, const float* dy, float* dw, int B, int Hi,
                           int Wi, int Cin, int Cout, int Kh, int Kw, int stride, int pad, void* ws, size_t ws_bytes,
                           fpc_stream_t stream);

/* Bilinear upsampling with align_corners = True (the x2 steps of the FPN segmentation blocks and the x4 of the heads:
 * torch.nn.functional.interpolate / nn.UpsamplingBilinear2d in the reference's network, F/lib/pose_regressor.py:608-666),
 * forward and its exact adjoint, ATen's source-index arithmetic.  scale: 2 or 4.  The tensor that is READ is given by its
 * element strides (any layout); the one WRITTEN is contiguous [B,C,.,.], channel-last when *_nhwc != 0.  The backward runs
 * one axis at a time through `scratch` (fpc_upsample_bilinear_bwd_scratch_floats floats).  Deterministic. */
int fpc_upsample_bilinear_fwd(const float* in, int64_t sb, int64_t sc, int64_t sh, int64_t sw, float* out, int B, int C, int h,
                              int w, int scale, int out_nhwc, fpc_stream_t stream);
size_t fpc_upsample_bilinear_bwd_scratch_floats(int B, int C, int h, int w, int scale);
int fpc_upsample_bilinear_bwd(const float* dout, int64_t sb, int64_t sc, int64_t sh, int64_t sw, float* din, float* scratch,
                              int B, int C, int h, int w, int scale, int din_nhwc, fpc_stream_t stream);

/* GroupNorm + ReLU on channel-last activations for the training step (the decoder blocks' GroupNorm(32, 128) -> ReLU,
 * segmentation_models_pytorch Conv3x3GNReLU, call sites F/lib/pose_regressor.py:608-666): x, y, dy, dx are [B, HW, C]
 * with C = 4 * groups (a group is one float4 of a pixel; groups divides 256 and is <= 64; FPC_EINVAL otherwise).
 * stats [B][groups][2] = mean, rstd (forward -> backward).  part: fpc_groupnorm4_relu_scratch_floats floats; after the
 * backward it holds [B][chunks][C][2] = per-chunk {sum dz, sum dz * xhat}, whose sums over the first two axes are
 * dbeta / dgamma.  Partial sums are combined in double in a fixed order: deterministic. */
size_t fpc_groupnorm4_relu_scratch_floats(int B, int HW, int C);
int fpc_groupnorm4_relu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, float* part, int B,
                            int HW, int C, int groups, float eps, fpc_stream_t stream);
int fpc_groupnorm4_relu_bwd(const float* x, const float* dy, const float* gamma, const float* beta, const float* stats, float* dx,
                            float* part, int B, int HW, int C, int groups, fpc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FPC_H_ */
