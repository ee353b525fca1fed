/* swin_hip.h -- C ABI of the MI355X (gfx950) Swin detection hot path.
 *
 * Drop-in boundary.  The reference has no FFI of its own: its
 * hot path calls torch ATen ops and the third-party mmcv-full extension from Python.
 * Every entry point below cites the reference call site(s) (file:line under the
 * upstream repository root) whose device work it replaces; INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless
 *     stated; the caller owns every buffer (no allocation, no host sync in here);
 *   - `stream` is a hipStream_t passed as void*; work is enqueued on it and the call
 *     returns immediately;
 *   - `dtype` selects the activation element type: SWIN_F32 (parity path, fp32 math)
 *     or SWIN_BF16 (16-bit storage, 16-bit MFMA products, fp32 accumulation/softmax/LN
 *     statistics).  Parameters (weights, biases, LN affine, bias table) are fp32.
 *     The library is built twice from the same sources: libswin_hip.so, in which "bf16" is
 *     bfloat16, and libswin_hip_f16.so (-DSWIN_HALF), in which it is IEEE half (the
 *     reference's apex O1 fp16, mmdet/apis/train.py:82-89); swin_hip_half_type() tells which;
 *   - activations are token-major: (B, H, W, C) row-major == (B*H*W, C);
 *   - return value: SWIN_OK or a SWIN_ERR_* code (nothing was launched on error).
 */
#ifndef SWIN_HIP_H
#define SWIN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWIN_OK 0
#define SWIN_ERR_BAD_ARG 1      /* null pointer / non-positive size */
#define SWIN_ERR_UNSUPPORTED 2  /* shape outside what the kernels implement */
#define SWIN_ERR_LAUNCH 3       /* hipGetLastError() after launch != hipSuccess */

#define SWIN_F32 0
#define SWIN_BF16 1             /* the library's 16-bit type: bfloat16 in libswin_hip.so, IEEE half in libswin_hip_f16.so (-DSWIN_HALF) */

#define SWIN_WINDOW 7           /* window_size of every config under configs/swin/ */
#define SWIN_HEAD_DIM 32        /* C / num_heads in Swin-T/S/B */
#define SWIN_ATTN_TILE 64       /* 49 tokens padded to the MFMA tile */

/* ABI version, bumped on any signature change (2: det_random_sample gained seed_dev; swin_set_scratch removed). */
int swin_hip_abi_version(void);
/* 0: this build's 16-bit type is bfloat16; 1: IEEE half */
int swin_hip_half_type(void);

/* ------------------------------------------------------------------------------------
 * LayerNorm over the channel dim.   Replaces nn.LayerNorm at
 * swin_transformer.py:211 (norm1), :253 (norm2), :295 (PatchMerging.norm),
 * :442 (PatchEmbed.norm), :620 (norm{i}).
 * x,y: (rows, C) dtype; gamma,beta: (C) f32; mean,rstd: (rows) f32 (saved for bwd; may be NULL).
 * ---------------------------------------------------------------------------------- */
int swin_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                       float* mean, float* rstd, int64_t rows, int C, float eps, int dtype, void* stream);
/* dx = LN'(dy) [+ dres]  : (rows,C) dtype.  dres (nullable) is the gradient arriving at x directly
 * (the residual branch), folded in here.  dx_scaled (nullable) = scale[b] * dx: the gradient of the
 * `y` operand of swin_add_layernorm_fwd.  dgamma/dbeta: (C) f32, ACCUMULATED (fp32 atomics). */
int swin_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                       const float* rstd, const void* dres, void* dx, void* dx_scaled, const float* scale,
                       int64_t rows_per_sample, float* dgamma, float* dbeta,
                       int64_t rows, int C, int dtype, void* workspace, void* stream);
/* workspace (nullable): swin_layernorm_bwd_workspace_bytes(rows, C, dtype) bytes of scratch, no initialisation needed.
 * With it the per-block partial sums of dgamma / dbeta are stored and summed by a second small kernel; without it they
 * are added with float atomics (hundreds of blocks on the same 2C addresses: several times slower). */
int64_t swin_layernorm_bwd_workspace_bytes(int64_t rows, int C, int dtype);

/* Fused residual + LayerNorm:  xo = x + scale[b] * y ;  n = LN(xo).
 * Replaces swin_transformer.py:252 (`shortcut + drop_path(x)`) followed by :253's norm2, and
 * :253's residual followed by the next block's :211 norm1.  scale: (B) f32 DropPath factors
 * (NULL = 1), rows_per_sample = H*W.  n/mean/rstd may be NULL (residual only). */
int swin_add_layernorm_fwd(const void* x, const void* y, const float* scale, int64_t rows_per_sample,
                           const float* gamma, const float* beta, void* xo, void* n,
                           float* mean, float* rstd, int64_t rows, int C, float eps, int dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Window attention core with cyclic shift, padding, window partition/reverse, relative
 * position bias and shift mask fused.   Replaces swin_transformer.py:214-231 (pad, roll,
 * window_partition), :129-150 (q*scale, q@k^T, +bias, +mask, softmax, @v), :237-247
 * (window_reverse, roll back, crop) and BasicLayer's mask construction :371-389.
 *
 *   qkv   : (B, H, W, 3C) dtype -- output of the qkv Linear on the UN-padded, UN-shifted
 *           token grid (channel = which*C + head*32 + d, swin_transformer.py:129)
 *   qkv_bias : (3C) f32 -- value of a padded token's q|k|v (pad is applied after LN1 so a
 *           padded token is 0 before the Linear, :211-218); may be NULL iff H%7==0 && W%7==0
 *   bias_exp : (nH, 64, 64) f32 expanded relative-position bias from swin_rel_bias_expand
 *   out   : (B, H, W, C) dtype, channel = head*32 + d (:150)
 *   lse   : (B*nW*nH, 64) f32 log-sum-exp per query (saved for bwd; may be NULL)
 *   shift : 0 or 3 (window_size//2, :346)
 * ---------------------------------------------------------------------------------- */
int swin_window_attn_fwd(const void* qkv, const float* qkv_bias, const float* bias_exp, void* out,
                         float* lse, int B, int H, int W, int C, int nH, int shift, float scale,
                         int dtype, void* stream);

/* Backward of the above.
 *   dout  : (B,H,W,C) dtype;  dqkv: (B,H,W,3C) dtype (fully written)
 *   dbias_exp : (nH,64,64) f32, ACCUMULATED -- reduce with swin_rel_bias_reduce
 *   dqkv_bias_pad : (3C) f32, ACCUMULATED: gradient reaching qkv.bias through padded tokens
 *                   (may be NULL iff no padding)
 *   workspace : >= swin_window_attn_bwd_workspace_bytes() bytes of device scratch (bf16 path: one
 *               (64,64) f32 slab per persistent wave for the bias gradient; may be NULL for SWIN_F32) */
int64_t swin_window_attn_bwd_workspace_bytes(int B, int H, int W, int nH, int dtype);
int swin_window_attn_bwd(const void* qkv, const float* qkv_bias, const float* bias_exp, const float* lse,
                         const void* dout, void* dqkv, float* dbias_exp, float* dqkv_bias_pad,
                         void* workspace, int B, int H, int W, int C, int nH, int shift, float scale,
                         int dtype, void* stream);

/* table (169, nH) f32  ->  bias_exp (nH, 64, 64) f32 laid out [head][key][query] with
 * -30000 for key >= 49 (folds the 49->64 padding mask into the bias).  Index rule of
 * swin_transformer.py:101-110.  */
int swin_rel_bias_expand(const float* table, float* bias_exp, int nH, void* stream);
/* the same for n tables (nH[i] heads each) in one launch: the expansions are a function of the parameters alone, so a training
 * loop rebuilds all of them once per optimizer step instead of once per block forward */
int swin_rel_bias_expand_multi(const float* const* tables, float* const* outs, const int* nH, int n, void* stream);
/* dbias_exp (nH,64,64) -> dtable (169,nH), ACCUMULATED. */
int swin_rel_bias_reduce(const float* dbias_exp, float* dtable, int nH, void* stream);
/* The small reductions behind a block backward as ONE launch (csrc/tail_reduce.hip); problem i of n, parallel arrays:
 *   kind 0 (column sums): src (rows, cols) f32; columns [0, a0) are ADDED into dst0, columns [a0, cols) into dst1 -- the
 *           [dgamma | dbeta] partial rows swin_layernorm_bwd leaves in its workspace (rows = workspace bytes / (8 C), a0 = C);
 *   kind 1 (relative position bias): src = the attention backward's workspace (rows slabs of `cols` floats, slab s belongs to
 *           head s % a0); ADDS the [key][query] tiles into dst0 = dtable (169, a0) by the index rule of swin_transformer.py:105-110
 *           and the pad-token sums into dst1 = dqkv_bias_pad (3, a1) (NULL: skipped).
 * swin_block_bwd uses this internally for norm2 / the next norm / the bias table (no (nH,64,64) intermediate, no memset). */
int swin_tail_reduce(const int* kind, const float* const* src, float* const* dst0, float* const* dst1, const int* rows,
                     const int* cols, const int* a0, const int* a1, int n, void* stream);

/* ------------------------------------------------------------------------------------
 * Elementwise pieces of the block.
 * ---------------------------------------------------------------------------------- */
/* y = gelu_erf(x + bias)  (nn.GELU, swin_transformer.py:34; bias = fc1.bias, may be NULL). */
int swin_bias_gelu_fwd(const void* x, const float* bias, void* y, int64_t rows, int C, int dtype, void* stream);
/* dx = dy * gelu'(x + bias);  dbias (C) f32 += column sums of dx (NULL to skip; ACCUMULATES) */
int swin_bias_gelu_bwd(const void* dy, const void* x, const float* bias, void* dx, float* dbias, int64_t rows, int C,
                       int dtype, void* stream);

/* PatchMerging gather + LayerNorm: swin_transformer.py:284-295.
 * x (B,H,W,C) -> y (B,ceil(H/2),ceil(W/2),4C) normalised; concat order x0(even,even), x1(odd,even),
 * x2(even,odd), x3(odd,odd); zero pad for odd H/W.  mean/rstd (rows_out) saved for bwd. */
int swin_patch_merge_ln_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                            float* rstd, int B, int H, int W, int C, float eps, int dtype, void* stream);
int swin_patch_merge_ln_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                            const float* rstd, void* dx, float* dgamma, float* dbeta,
                            int B, int H, int W, int C, int dtype, void* workspace, void* stream);
/* workspace (nullable): swin_layernorm_bwd_workspace_bytes(B*ceil(H/2)*ceil(W/2), 4*C, dtype) bytes. */

/* PatchEmbed im2row: img (B,3,Hi,Wi) f32 NCHW -> rows (B*ceil(Hi/4)*ceil(Wi/4), 48) dtype, zero pad
 * right/bottom (swin_transformer.py:433-438; column index = c*16 + ky*4 + kx = conv weight layout). */
int swin_patch_im2row(const float* img, void* rows, int B, int Hi, int Wi, int dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * FPN top-down step: lat_fine += nearest_upsample(lat_coarse)   (fpn.py:188-191), arbitrary target
 * size (F.interpolate(size=..., mode='nearest')).  channels_last = 0: (N,C,H,W) memory;
 * 1: (N,H,W,C) memory (the token-major layout the backbone produces).
 * ---------------------------------------------------------------------------------- */
int fpn_upsample_add_fwd(void* fine, const void* coarse, int N, int C, int Hf, int Wf, int Hc, int Wc,
                         int channels_last, int dtype, void* stream);
/* dcoarse += sum of dfine over each coarse cell's footprint */
int fpn_upsample_add_bwd(const void* dfine, void* dcoarse, int N, int C, int Hf, int Wf, int Hc, int Wc,
                         int channels_last, int dtype, void* stream);
/* out-of-place forms (round 2): out = fine + nearest_upsample(coarse) leaves `fine` untouched (the caller needs no clone);
 * the backward writes dcoarse from scratch (the caller needs no memset). */
int fpn_upsample_add_out_fwd(const void* fine, const void* coarse, void* out, int N, int C, int Hf, int Wf, int Hc, int Wc,
                             int channels_last, int dtype, void* stream);
int fpn_upsample_add_out_bwd(const void* dfine, void* dcoarse, int N, int C, int Hf, int Wf, int Hc, int Wc,
                             int channels_last, int dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * bf16 MFMA implicit-GEMM convolution, 3x3 / pad 1 / stride 1, channels-last.  Replaces the conv
 * library calls behind FPN's output convs (fpn.py:195-197), the RPN conv (rpn_head.py:43) and the FCN
 * mask-head convs (fcn_mask_head.py:119-121).
 *   x (N,H,W,Cin) bf16;  w (Cout,3,3,Cin) bf16 (= channels_last memory of the (Cout,Cin,3,3) parameter);
 *   bias (Cout) f32 or NULL;  y (N,H,W,Cout) bf16 = conv(x,w)+bias, ReLU if relu != 0.
 *   Cin % 64 == 0, Cout % 4 == 0.  The data gradient is the same call on dy with the 180-degree rotated,
 *   in/out-transposed weight.
 * ---------------------------------------------------------------------------------- */
int conv3x3_nhwc_bf16(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int Cin,
                      int Cout, int relu, void* stream);

/* Weight gradients (contraction over the token / pixel index, split over the grid, fp32 atomics):
 *   wgrad_linear_bf16      : dw (N1,N2) f32 += dy (T,N1)^T x (T,N2)   -- autograd of nn.Linear
 *                            (swin_transformer.py:129,151,33,36,296); N1 % 8 == 0, N2 % 8 == 0
 *   wgrad_conv3x3_nhwc_bf16: dw (Cout,3,3,Cin) f32 += gradient of conv3x3_nhwc_bf16's weight (implicit im2col)
 * dbias (N1 | Cout) f32 += column sums of dy (the bias gradient), NULL to skip.
 * All outputs ACCUMULATE (the caller zeroes them, or passes the parameter's fp32 gradient buffer). */
int wgrad_linear_bf16(const void* dy, const void* x, float* dw, float* dbias, int64_t T, int N1, int N2, void* stream);
int wgrad_conv3x3_nhwc_bf16(const void* dy, const void* x, float* dw, float* dbias, int N, int H, int W, int Cin,
                            int Cout, void* stream);
/* Grouped Linear weight gradients.  swin_wgrad_record has wgrad_linear_bf16's meaning but only RECORDS the problem (per device);
 * swin_wgrad_flush launches everything recorded as a few grouped kernels on `stream` -- a stage's worth of the backbone's
 * four-per-block weight gradients in one launch needs no or few splits of t (hence few float atomics) and streams long loops instead
 * of paying a latency chain per launch.  The caller keeps every recorded operand alive and unmodified until the flush, orders `stream`
 * behind their producers, and does not read the accumulators before the flush.  swin_wgrad_pending: recorded problems (and, through
 * *tiles, their 128 x 128 output tiles).  swin_block_bwd records its four weight gradients instead of launching them when iv[7] != 0.
 * Not in the reference: autograd computes each Linear's weight gradient where its backward node runs. */
int swin_wgrad_record(const void* dy, const void* x, float* dw, float* dbias, int64_t T, int N1, int N2);
int swin_wgrad_pending(int64_t* tiles);
int swin_wgrad_flush(void* stream);
/* The grouped launch for layers whose dimensions are multiples of 96 (csrc/wgrad96.hip; swin_wgrad_flush routes to it): n <= 32
 * problems, N1 % 96 == 0 and N2 % 96 == 0 -- qkv / proj / fc1 / fc2 / PatchMerging.reduction of Swin-T and Swin-S at every stage.
 * A wave owns a 96 x 96 piece of dW, a block a tile of up to 384 x 192; the launch's (problem, cluster of <= 8 tiles, stage) sequence
 * is cut into equal ranges, one per group of eight blocks that walk a cluster's tiles in step on one XCD.
 * dy / x / dw / db / T / N1 / N2: host arrays of n entries (db or db[i] NULL = no bias gradient).
 * SWIN_ERR_UNSUPPORTED for any other shape.  Outputs ACCUMULATE. */
int swin_wgrad96_group(const void* const* dy, const void* const* x, float* const* dw, float* const* db, const int64_t* T,
                       const int* N1, const int* N2, int n, void* stream);

/* ------------------------------------------------------------------------------------
 * mmcv.ops.RoIAlign / roi_align ('avg', aligned flag) -- call sites
 * base_roi_extractor.py:49-55, single_level_roi_extractor.py:93-97, structures.py:353-354.
 * rois (K,5) f32 [batch_idx,x1,y1,x2,y2]; arithmetic and output fp32.
 *   channels_last = 0: input (N,C,H,W), output (K,C,ph,pw)      (mmcv's memory layout)
 *   channels_last = 1: input (N,H,W,C), output (K,ph,pw,C)      (coalesced along C)
 *   in_dtype: element type of `input` (SWIN_F32 or SWIN_BF16, converted on load -- the
 *             reference's force_fp32 at single_level_roi_extractor.py:53).
 * ---------------------------------------------------------------------------------- */
int roi_align_fwd(const void* input, const float* rois, float* output, int N, int C, int H, int W,
                  int K, int ph, int pw, float spatial_scale, int sampling_ratio, int aligned,
                  int channels_last, int in_dtype, void* stream);
/* grad_input f32 in the same memory layout as the input, zeroed by the caller; fp32 atomics. */
int roi_align_bwd(const float* grad_output, const float* rois, float* grad_input, int N, int C, int H, int W,
                  int K, int ph, int pw, float spatial_scale, int sampling_ratio, int aligned,
                  int channels_last, void* stream);

/* SingleRoIExtractor.forward (single_level_roi_extractor.py:82-97) in one launch over an FPN pyramid in
 * channels-last memory: RoI k reads level lvl[k] (host-side map_roi_levels, :47-51); lvl[k] < 0 skips the RoI
 * (zero output row).  feats / grads are HOST arrays of n_levels (<= 4) device pointers. */
int roi_align_multilevel_fwd(const void* const* feats, const int* Hs, const int* Ws, const float* scales,
                             int n_levels, const float* rois, const int* lvl, void* output, int C, int K,
                             int ph, int pw, int sampling_ratio, int aligned, int in_dtype, int out_dtype, void* stream);
int roi_align_multilevel_bwd(float* const* grads, const int* Hs, const int* Ws, const float* scales, int n_levels,
                             const void* grad_output, const float* rois, const int* lvl, int C, int K, int ph,
                             int pw, int sampling_ratio, int aligned, int grad_dtype, void* stream);
/* The same gradient in GATHER form (round 3): every 8 x 8-pixel tile of every level is written once by the blocks that own it --
 * no float atomics, no zero fill by the caller (grads[l] (N,H_l,W_l,C) in out_dtype are FULLY written), no cast afterwards, and a
 * summation order independent of scheduling.  n_sets (<= 4) RoI sets pooled from the same pyramid are differentiated together
 * (the bbox head's 7x7 and the mask head's 14x14 RoIs of one stage): gouts[s] (K[s], ph[s], pw[s], C) channels-last in grad_dtype.
 * ph, pw <= 16, else SWIN_ERR_UNSUPPORTED.  workspace: roi_align_gather_workspace_bytes(...) bytes, ZEROED ONCE by the caller and
 * then reusable call after call (the kernels leave its counters zero). */
int64_t roi_align_gather_workspace_bytes(const int* Hs, const int* Ws, int n_levels, int N, int K_total);
int roi_align_multilevel_bwd_gather(void* const* grads, const int* Hs, const int* Ws, const float* scales, int n_levels, int N,
                                    int n_sets, const void* const* gouts, const float* const* rois, const int* const* lvls,
                                    const int* Ks, const int* phs, const int* pws, int C, int sampling_ratio, int aligned,
                                    int grad_dtype, int out_dtype, void* workspace, int64_t workspace_bytes, void* stream);
/* output / grad_output: (K, ph, pw, C) f32, or bf16 when out_dtype / grad_dtype = 1 (bf16 features only). */

/* ------------------------------------------------------------------------------------
 * mmcv.ops.nms device part -- call sites rpn_head.py:233, bbox_nms.py:84 (through batched_nms).
 * boxes_sorted (n,4) f32 ALREADY in descending-score order (stable sort done by the host
 * layer).  Writes keep_flags (n) uint8 (1 = kept) and *num_kept (device int32).
 * workspace: >= swin_nms_workspace_bytes(n) bytes of device memory.
 * The whole greedy reduction runs on the device (mmcv copies the bitmask to the host).
 * max_num > 0: stop after max_num kept boxes (mmcv nms's `max_num`; == slicing the result).
 * kept_pos (nullable): fixed-size (kept_cap) int32 list of the kept positions in sorted order, -1 padded --
 *                      lets a caller consume the result without a device->host sync for the dynamic count.
 * ---------------------------------------------------------------------------------- */
int64_t swin_nms_workspace_bytes(int64_t n);
int nms_sorted(const float* boxes_sorted, int64_t n, float iou_threshold, int offset, int max_num,
               uint8_t* keep_flags, int32_t* num_kept, int32_t* kept_pos, int kept_cap, void* workspace,
               void* stream);

/* nms_sorted for `batch` images of the same n in one pair of launches (leading batch dimension on every buffer;
 * workspace batch * swin_nms_workspace_bytes(n)): the per-image single-workgroup reductions run concurrently. */
int nms_sorted_batch(const float* boxes_sorted, int batch, int64_t n, float iou_threshold, int offset, int max_num,
                     uint8_t* keep_flags, int32_t* num_kept, int32_t* kept_pos, int kept_cap, void* workspace,
                     void* stream);
/* nms_sorted_batch for candidates that carry a class / level id whose boxes were offset apart (mmcv batched_nms, rpn_head.py:233:
 * boxes of different ids never overlap): the greedy scan of the whole list = the scans of the ids' sub-lists, which run side by
 * side (five lists of <= 2000 instead of one of 8780 for the RPN: 32 dependent steps instead of 138).  order (batch, n): source
 * index of sorted position i (nms_prepare_sorted_batch); idxs (batch, n) int64 in [0, groups) by SOURCE index; gmax: no group has
 * more members (the caller's bound); groups <= 8, iou_threshold > 0.  Same outputs, bit for bit, as nms_sorted_batch.
 * workspace: nms_grouped_workspace_bytes(batch, n, groups, gmax). */
int64_t nms_grouped_workspace_bytes(int batch, int64_t n, int groups, int gmax);
int nms_sorted_batch_grouped(const float* boxes_sorted, const int32_t* order, const int64_t* idxs, int batch, int64_t n, int groups,
                             int gmax, float iou_threshold, int offset, int max_num, uint8_t* keep_flags, int32_t* num_kept,
                             int32_t* kept_pos, int kept_cap, void* workspace, void* stream);

/* ---- training targets of the detector heads (csrc/det_targets.hip) --------------------------------------------
 * det_max_iou_assign replaces MaxIoUAssigner.assign (mmdet/core/bbox/assigners/max_iou_assigner.py:128-212 with
 * BboxOverlaps2D, iou2d_calculator.py) as called from anchor_head.py:215 and standard_roi_head.py:85:
 *   bboxes (n,4) f32 xyxy, gt_bboxes (g,4) f32, gt_labels (g) i64 or NULL ->
 *   assigned_gt_inds (n) i64 (-1 ignore / 0 background / k+1), max_overlaps (n) f32, assigned_labels (n) i64 or NULL.
 *   num_leading_gt: the first rows of bboxes are the gts themselves (add_gt_as_proposals, base_sampler.py:77-84) and
 *   match themselves; valid (n) u8 or NULL: rows with 0 get -1.  workspace: det_assign_workspace_bytes(n, g).
 * det_random_sample replaces RandomSampler.sample (random_sampler.py:31-78, base_sampler.py:54-96) with a fixed-size
 *   result: out_inds (num) i64, out_flags (num) u8 (bit 0 slot used, bit 1 positive), positives first. */
int64_t det_assign_workspace_bytes(int64_t n, int num_gts);
int det_max_iou_assign(const float* bboxes, int64_t n, const float* gt_bboxes, int num_gts, const int64_t* gt_labels,
                       float pos_iou_thr, float neg_iou_thr, float min_pos_iou, int match_low_quality,
                       int num_leading_gt, const uint8_t* valid, int64_t* assigned_gt_inds, float* max_overlaps,
                       int64_t* assigned_labels, void* workspace, void* stream);
int64_t det_random_sample_workspace_bytes(void);
int det_random_sample(const int64_t* assigned_gt_inds, int64_t n, int num, int num_pos_max, uint64_t seed,
                      const uint64_t* seed_dev, int64_t* out_inds, uint8_t* out_flags, void* workspace, void* stream);
/* seed_dev (nullable): DEVICE pointer to a 64-bit word that the kernels mix into `seed`.  A training step captured in a hipGraph
 * freezes the host seed of each call into the graph; the device word (rewritten before every replay: swin_set_u64) keeps the
 * samples different from step to step, as RandomSampler's torch.randperm does (random_sampler.py:54). */
int swin_set_u64(void* dst, uint64_t value, void* stream);

/* det_bbox_targets: targets of a fixed-size sample -- the gathers of anchor_head.py:221-247 / bbox_head.py:140-186 plus
 *   DeltaXYWHBBoxCoder.encode (delta_xywh_bbox_coder.py:82-130).  means / stds are HOST pointers to 4 floats.
 * det_delta2bbox: DeltaXYWHBBoxCoder.decode (delta_xywh_bbox_coder.py:189-237) for (n,4) rois / deltas. */
int det_bbox_targets(const float* bboxes, const int64_t* inds, const uint8_t* flags, const int64_t* assigned_gt_inds,
                     const float* gt_bboxes, int num_gts, const int64_t* assigned_labels, int64_t bg_label,
                     const float* means, const float* stds, int k, float* out_bboxes, float* out_deltas,
                     int64_t* out_gt_inds, int64_t* out_labels, void* stream);
int det_delta2bbox(const float* rois, const float* deltas, int64_t n, const float* means, const float* stds,
                   float max_h, float max_w, float wh_ratio_clip, float* out, void* stream);

/* det_roi_targets_pack: one image's rows of an R-CNN stage's batch-level training tensors from its fixed-size sample --
 *   standard_roi_head.py:83-93 (SamplingResult: boxes, pos_is_gt), bbox_head.py:140-186 (get_targets: labels, encoded or
 *   decoded regression targets), bbox2roi (transforms.py:117-137: the image index column) and, for the first km slots,
 *   mask_target.py:95-107 + structures.py:343-350 (rows [gt index + gt_offset, box clipped to (mask_h, mask_w)] for
 *   crop_and_resize), the mask head's labels (clamped below the background label) and validity.  Every out_* pointer
 *   addresses THIS image's first row; pos / valid / is_gt / mvalid are 0/1 bytes (torch.bool storage). */
int det_roi_targets_pack(const float* bboxes, const int64_t* inds, const uint8_t* flags, const int64_t* assigned_gt_inds,
                         const float* gt_bboxes, int num_gts, const int64_t* assigned_labels, int64_t bg_label,
                         const float* means, const float* stds, int k, int img, int num_leading_gt, int reg_decoded,
                         float* out_rois5, float* out_targets, int64_t* out_labels, uint8_t* out_pos, uint8_t* out_valid,
                         uint8_t* out_is_gt, int km, int gt_offset, float mask_h, float mask_w, float* out_feat_rois5,
                         float* out_mask_rois5, int64_t* out_mlabels, uint8_t* out_mvalid, void* stream);

/* det_paste_masks: test-time FCNMaskHead.get_seg_masks / _do_paste_mask (fcn_mask_head.py:169-300, :303-377):
 *   mask_logits (N, num_classes, mh, mw) f32|bf16, labels (N) i64, boxes (N,4) f32 in output-image coordinates ->
 *   out (N, img_h, img_w) u8 = (bilinear resample of sigmoid(logits[n, labels[n]]) into the box) >= thr.  is_prob != 0:
 *   the input already holds probabilities (CascadeRoIHead averages the stages' sigmoid masks, cascade_roi_head.py:383-396). */
int det_paste_masks(const void* mask_logits, const int64_t* labels, const float* boxes, int N, int num_classes,
                    int mh, int mw, int img_h, int img_w, float thr, int is_prob, int in_dtype, uint8_t* out,
                    void* stream);

/* swin_adamw_step: AdamW over all parameters in ONE launch (+ the bf16 operand copy of the GEMM/conv weights).
 * Replaces torch.optim.AdamW.step as configured by configs/swin/ *_coco.py:64-67 and the master->half copy of apex O1
 * (mmdet/apis/train.py:82-89).  segs: DEVICE array of {float* p; const float* g; float* m; float* v; bf16* shadow
 * (nullable); int64 n; int32 group; int32 pad} (56 bytes each); chunks: DEVICE array of int32 pairs (segment, chunk
 * index), one per swin_adamw_chunk_elems() elements of a segment; lr / weight_decay: HOST arrays of n_groups <= 8. */
int swin_adamw_step(const void* segs, const void* chunks, int n_chunks, const float* lr, const float* weight_decay,
                    int n_groups, float beta1, float beta2, float eps, float bias_correction1,
                    float bias_correction2, void* stream);
int swin_adamw_chunk_elems(void);
/* The same step with every per-step scalar in DEVICE memory, so that the launch's arguments never change and it can sit inside a
 * captured hipGraph (and so that fp16 loss scaling can skip a step without a host round trip).  state: 32 floats --
 *   [0..7] lr per group, [8..15] weight decay per group, [16] bias_correction1, [17] sqrt(bias_correction2),
 *   [18] grad_scale (every gradient is multiplied by it: 1 / loss scale), [19] skip (!= 0: parameters and moments untouched),
 *   [20] loss scale, [21] clean steps since the scale last changed, [22..31] reserved.
 * swin_adamw_set_state writes [0..17] from its ARGUMENTS with a one-block kernel (no host->device copy); the caller initialises
 * [18..21] (1, 0, 1, 0 without loss scaling). */
int swin_adamw_set_state(void* state, const float* lr, const float* weight_decay, int n_groups, float bias_correction1,
                         float bias_correction2, void* stream);
int swin_adamw_step_dev(const void* segs, const void* chunks, int n_chunks, const void* state, float beta1, float beta2,
                        float eps, void* stream);
/* fp16 dynamic loss scaling on the device, in the same state block (apex O1's dynamic LossScaler, mmdet/apis/train.py:82-89: 2^16,
 * halved on an inf / nan gradient -- that step is skipped --, doubled after 2000 clean steps).  Per step: swin_loss_scale_begin
 * (skip = 0, grad_scale = 1 / loss scale), swin_grad_check_finite per gradient bucket (g: 16-byte aligned; skip = 1 on a non-finite
 * value), the optimizer launch, swin_loss_scale_update.  The loss is multiplied by state[20] on the device. */
int swin_loss_scale_begin(void* state, void* stream);
int swin_grad_check_finite(const float* g, int64_t n, void* state, void* stream);
int swin_loss_scale_update(void* state, float growth, float backoff, int growth_interval, float min_scale, float max_scale,
                           void* stream);

/* swin_gemm_bf16: the plain GEMMs of the path (nn.Linear forward / data gradient, swin_transformer.py:33-36,129,151,296;
 * 1x1 convs; head FCs) on hipBLASLt with cached plans -- one library launch per call, no framework dispatch.
 *   c (M,N) bf16 = a (M,K) bf16 x op(b) [+ bias (N) bf16]; b_layout 0: b is (N,K) (c = a b^T), 1: b is (K,N) (c = a b).
 *   workspace: swin_gemm_workspace_bytes() bytes of device scratch. */
int64_t swin_gemm_workspace_bytes(void);
int swin_gemm_bf16(const void* a, const void* b, const void* bias, void* c, int64_t M, int N, int K, int b_layout,
                   void* workspace, void* stream);
/* swin_linear_hip_bf16: the same GEMM on the hand-written MFMA kernel of csrc/conv_gemm.hip (128 x 128 x 64 tiles, LDS-DMA):
 *   c (M,N) bf16 = a (M,K) bf16 x w (N,K)^T + bias (N) f32 (nullable), ReLU when relu != 0.  K % 64 == 0, N % 4 == 0. */
int swin_linear_hip_bf16(const void* a, const void* w, const float* bias, void* c, int64_t M, int N, int K, int relu,
                         void* stream);
/* The two GELU-fused GEMMs of Mlp (swin_transformer.py:32-38) for the widths csrc/ts_mlp.hip does not cover (C >= 256 and every Swin-B
 * width), on the same hand-written kernel, so that no pass over a T x 4C tensor is spent on the activation:
 *   swin_linear_gelu_hip_bf16 : hpre (M,N) = a (M,K) w (N,K)^T (no bias: the pre-activation the backward reads); h (M,N) = gelu_erf(that + bias)
 *   swin_linear_dgelu_hip_bf16: dhpre (M,N) = (dy (M,K) wt (N,K)^T) * gelu_erf'(hpre + bias), wt = the TRANSPOSED fc2 weight (4C, C)
 *   linear_t_layout_multi     : dsts[k] (cols, rows) = srcs[k] (rows, cols)^T, bf16, n matrices per call (HOST arrays) -- those
 *                               transposed weights, rebuilt once per optimizer step.   K % 64 == 0, N % 4 == 0; bias (N) f32. */
int swin_linear_gelu_hip_bf16(const void* a, const void* w, const float* bias, void* hpre, void* h, int64_t M, int N, int K, void* stream);
int swin_linear_dgelu_hip_bf16(const void* dy, const void* wt, const void* hpre, const float* bias, void* dhpre, int64_t M, int N, int K,
                               void* stream);
int linear_t_layout_multi(const void* const* srcs, void* const* dsts, const int* rows, const int* cols, int n, void* stream);
/* A plan's algorithm is chosen by timing the library's candidates on first use, so two data-parallel ranks may choose
 * differently.  swin_gemm_plans_export: records of 6 int64 {M, N, K, b_layout, has_bias, chosen candidate index} of the current
 * device's plans into HOST memory `out` (capacity `cap` records); returns the number of plans.  swin_gemm_plans_import: select the
 * recorded candidate for every plan that exists here; returns the number of plans changed.  (Rank 0 exports after warm-up, the
 * records are broadcast, every rank imports: ddp.sync_gemm_plans.) */
int swin_gemm_plans_export(int64_t* out, int cap);
int swin_gemm_plans_import(const int64_t* in, int n);

/* ---- loss kernels of the detector heads (csrc/det_losses.hip): value + input gradients, fixed-size samples ----------
 * det_rpn_loss_*:  AnchorHead.loss_single (anchor_head.py:375-434): sigmoid CE over the sampled anchors + L1 (beta 0) or
 *   SmoothL1(beta) (smooth_l1_loss.py:10-28; the Cascade configs' RPN uses beta 1/9) on the positives, divided by the
 *   batch's sample count.  cls (B,A), reg (B,A,4) f32|bf16; inds/flags (B*S) from
 *   det_random_sample, targets (B*S,4) from det_bbox_targets; out3 = {loss_cls, loss_bbox, n}.
 * det_bbox_loss_*: BBoxHead.loss (bbox_head.py:188-238): softmax CE, accuracy, regression on the positives; out4 =
 *   {loss_cls, acc %, loss_bbox, n_valid}.  bbox is (n, 4 nc), or (n, 4) when class_agnostic.  reg_mode 0: L1 (beta 0) /
 *   SmoothL1(beta) against encoded delta targets; reg_mode 2: reg_decoded_bbox=True with GIoULoss (iou_loss.py:78-101,
 *   iou2d_calculator.py:108-158): the deltas are decoded against rois (n,4) with means/stds (HOST float[4]) and compared
 *   with gt boxes in `targets`; eps is GIoULoss.eps.  rois/means/stds may be NULL in mode 0.
 * det_mask_loss_*: FCNMaskHead.loss / mask_cross_entropy (cross_entropy_loss.py): mean BCE of the labelled channel over
 *   the valid RoIs; out2 = {loss, n_valid}.  deconv_w 0: pred (n, nc, P) NCHW; deconv_w = mask width (P = deconv_w^2):
 *   pred in the row order of the ConvTranspose2d(k=2,s=2)-as-GEMM output, (roi, h, w, ky, kx, class) -- the logits
 *   before the 2x2 pixel shuffle of fcn_mask_head.py:122-126, so training skips the shuffle and layout copies.
 * Backward entries take grad_out aligned with the forward's out array; dcls/dreg (rpn) and dpred (mask) must be zeroed
 * by the caller, dcls/dbbox (bbox) are fully written.  det_bbox_loss_fwd's lse holds n + 4 * ceil(n / 16) floats: the rows'
 * log-sum-exp (kept for the backward) followed by the call's scratch. */
int det_rpn_loss_fwd(const void* cls, const void* reg, int B, int64_t A, int S, const int64_t* inds, const uint8_t* flags,
                     const float* targets, float beta, float* out3, int dtype, void* stream);
int det_rpn_loss_bwd(const void* cls, const void* reg, int B, int64_t A, int S, const int64_t* inds, const uint8_t* flags,
                     const float* targets, float beta, const float* out3, const float* grad_out, void* dcls, void* dreg,
                     int dtype, void* stream);
int det_bbox_loss_fwd(const void* cls, const void* bbox, int n, int num_classes, const int64_t* labels, const float* targets,
                      const uint8_t* flags, int reg_mode, int class_agnostic, float beta, float eps, const float* rois,
                      const float* means, const float* stds, float* out4, float* lse, int dtype, void* stream);
int det_bbox_loss_bwd(const void* cls, const void* bbox, int n, int num_classes, const int64_t* labels, const float* targets,
                      const uint8_t* flags, int reg_mode, int class_agnostic, float beta, float eps, const float* rois,
                      const float* means, const float* stds, const float* out4, const float* lse, const float* grad_out,
                      void* dcls, void* dbbox, int dtype, void* stream);
int det_mask_loss_fwd(const void* pred, int n, int num_classes, int P, int deconv_w, const float* target,
                      const int64_t* labels, const uint8_t* valid, float* out2, float* per_roi, int dtype, void* stream);
int det_mask_loss_bwd(const void* pred, int n, int num_classes, int P, int deconv_w, const float* target,
                      const int64_t* labels, const uint8_t* valid, const float* out2, const float* grad_out, void* dpred,
                      int dtype, void* stream);

/* swin_set_aux_stream: per device, a second hipStream_t (NULL = none) for the small reductions at the end of
 *   swin_layernorm_bwd (parameter gradients), swin_window_attn_bwd (bias-gradient slabs) and swin_rel_bias_reduce: while it
 *   is set they are enqueued there behind an event recorded on the call's own stream, so they leave the data-gradient chain.
 *   The caller joins the two streams before reading those results and keeps the calls' workspaces untouched until then.
 *   swin_block_bwd sets it to its table entry 55 for the duration of the call. */
int swin_set_aux_stream(void* side);
/* swin_fork_stream: `side` waits for everything enqueued on `main` so far (an event from a per-device ring is recorded on
 *   `main` and waited for on `side`; no host synchronisation).  The join is the same call with the arguments swapped.
 *   NULL is the legacy default stream, as everywhere in HIP. */
int swin_fork_stream(void* main, void* side);
/* swin_stream_create_low_priority: a non-blocking hipStream_t of the device's lowest priority for that off-chain work (so that
 *   the hardware prefers the main stream's workgroups when both have some); the stream lives as long as the process. */
int swin_stream_create_low_priority(void** out);

/* conv3x3_nhwc_bf16_gated: conv3x3_nhwc_bf16 whose output is zeroed where gate (N,H,W,Cout) bf16 is not positive -- a data
 *   gradient that already includes the ReLU backward (torch.ops.aten.threshold_backward) of the layer below, whose output
 *   `gate` is (the conv -> ReLU -> conv chains of fcn_mask_head.py:73-104).
 * narrow_dgrad_gated_bf16: dx (T,C) = [gate > 0] * dy (T,K) w (K,C) for K <= 64: data gradient of the RPN's 1x1 cls / reg
 *   heads (rpn_head.py:41-47) fused with the ReLU backward of rpn_conv. */
int conv3x3_nhwc_bf16_gated(const void* x, const void* w, const float* bias, const void* gate, void* y, int N, int H, int W,
                            int Cin, int Cout, void* stream);
/* The same convolution (gate may be NULL; with a gate relu is ignored by the callers above) in its halo-staged form (csrc/conv_halo.hip):
 * a tile of 256 consecutive pixels stages its three input row segments ONCE per 32-channel block and streams only weight tiles.
 * conv3x3_nhwc_bf16 / _ws / _gated route the many-pixel maps here themselves; this entry always uses it (parity tests, tools).
 * nt: 2 (tile 256 x 128) or 4 (256 x 256, Cout % 256 == 0), 0 = choose.  Cin % 32 == 0, Cout % 8 == 0; else SWIN_ERR_UNSUPPORTED. */
int conv3x3_halo_nhwc_bf16(const void* x, const void* w, const float* bias, const void* gate, void* y, int N, int H, int W, int Cin,
                           int Cout, int relu, int nt, void* stream);
int narrow_dgrad_gated_bf16(const void* dy, const void* w, const void* gate, void* dx, int64_t T, int K, int C, void* stream);

/* conv3x3_splitk_workspace_bytes / conv3x3_nhwc_bf16_ws: the same convolutions for maps with so few 128 x 128 output tiles that one
 *   launch leaves most CUs idle for the whole 9 * Cin contraction (the coarse pyramid levels of fpn.py:195-197 and rpn_head.py:43:
 *   126 / 32 / 10 tiles at P4 / P5 / P6 of a 2 x 800 x 1280 batch, 40 us each whatever the size).  With `workspace` (device, at least
 *   conv3x3_splitk_workspace_bytes(...) bytes; that function returns 0 for shapes that are not split) the contraction is split over
 *   blockIdx.y into fp32 partial slabs and a second launch adds them and applies bias / ReLU / gate (gate != NULL: as
 *   conv3x3_nhwc_bf16_gated, relu ignored).  workspace == NULL: exactly conv3x3_nhwc_bf16(_gated).  Same result up to the order of
 *   the fp32 sum. */
int64_t conv3x3_splitk_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int conv3x3_nhwc_bf16_ws(const void* x, const void* w, const float* bias, const void* gate, void* y, int N, int H, int W, int Cin,
                         int Cout, int relu, void* workspace, int64_t workspace_bytes, void* stream);

/* conv_dgrad_layout_multi: for n 3x3 conv weights resident as (Cout,3,3,Cin) bf16, the (Cin,3,3,Cout) weights of their
 *   data-gradient convolutions (rot180, in/out swapped: what conv_transpose / the reference's cudnn backward-data use
 *   implicitly), all in one launch.  srcs / dsts / couts / cins are HOST arrays of n entries. */
int conv_dgrad_layout_multi(const void* const* srcs, void* const* dsts, const int* couts, const int* cins, int n, void* stream);

/* swin_block_fwd / swin_block_bwd: the whole SwinTransformerBlock (swin_transformer.py:204-255) and its backward as ONE
 * call each -- the library's own kernels launched in sequence from native code (csrc/block_runner.hip lists the
 * pointer-table layouts).  p: HOST array of device pointers, iv: {B,H,W,C,nH,shift}, fv: {scale[, eps]}.  No allocation,
 * no synchronisation; every buffer (saved activations, temporaries, gradient accumulators, workspaces) is the caller's.
 * swin_block_bwd's table entry 55 may name a second hipStream_t for the four weight-gradient GEMMs (null: `stream`): they are
 * enqueued there behind events recorded on `stream`, so they overlap the data-gradient chain; the caller joins the streams
 * before the accumulators are read and keeps the operands alive until then. */
int swin_block_fwd(const void* const* p, const int64_t* iv, const float* fv, void* stream);
int swin_block_bwd(const void* const* p, const int64_t* iv, const float* fv, void* stream);

/* det_rpn_flatten_*: per-level fused RPN head outputs (B, hw[l], CH) [A cls | 4A deltas | pad] <-> the anchor-major
 * concatenation over levels cls_all (B, sum hw*A), reg_all (B, sum hw*A, 4) that AnchorHead.loss (anchor_head.py:474-486)
 * and RPNHead._get_bboxes (rpn_head.py:119-125) consume; one kernel each way.  ys / dys: HOST arrays of device pointers. */
int det_rpn_flatten_fwd(const void* const* ys, const int* hw, int L, int B, int A, int CH, void* cls_all, void* reg_all,
                        int dtype, void* stream);
int det_rpn_flatten_bwd(void* const* dys, const int* hw, int L, int B, int A, int CH, const void* dcls_all,
                        const void* dreg_all, int dtype, void* stream);

/* det_regress_by_class: BBoxHead.regress_by_class (bbox_head.py:409-436) as used between the stages of CascadeRoIHead
 *   (cascade_roi_head.py:274-284 training, :316-323 testing): rois (n,4) f32; labels (n) i64 or NULL -- NULL, background
 *   (== num_classes) or negative labels take argmax(cls[:, :num_classes]); cls (n, nc+1), bbox (n, 4 nc | 4) f32|bf16;
 *   out (n,4) = decode(roi, deltas of the label) clipped to (max_h, max_w) when max_w > 0. */
int det_regress_by_class(const float* rois, const int64_t* labels, const void* cls, const void* bbox, int64_t n,
                         int num_classes, int class_agnostic, const float* means, const float* stds, float max_h,
                         float max_w, float* out, int dtype, void* stream);

/* ---- token-stationary fused MLP (csrc/ts_mlp.hip): Mlp.forward (swin_transformer.py:32-38: fc1 -> exact-erf GELU -> fc2)
 *   and its backward, bf16 in / fp32 accumulate, for C in {96, 192} (stages 1-2, where the 4C hidden activation is the
 *   step's largest HBM stream).  x, y, dy, dx (T,C) bf16; w1 (4C,C), w2 (C,4C) bf16; b1 (4C), b2 (C) f32.
 *   swin_mlp_fwd_bf16: y = fc2(gelu(fc1(x) + b1)) + b2; nothing of size T x 4C is written.
 *   swin_mlp_bwd_bf16: recomputes fc1(x) + b1, writes dx = ((dy w2) * gelu') w1 and, for the weight-gradient GEMMs
 *   (wgrad_linear_bf16), h = gelu(.) and dhpre = (dy w2) * gelu' as (T,4C) bf16. */
int swin_mlp_fwd_bf16(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, int64_t T,
                      int C, void* stream);
int swin_mlp_bwd_bf16(const void* x, const void* dy, const void* w1, const float* b1, const void* w2, void* dx, void* h,
                      void* dhpre, int64_t T, int C, void* stream);
/* swin_mlp_fwd_bf16 with the block's second residual and the next LayerNorm in its epilogue (swin_transformer.py:253, :211):
 *   x2 (T,C) = x1 + dp[row / rows_per_sample] * Mlp(x);  nn = LayerNorm(x2; gamma, beta, eps), mean / rstd (T) f32.
 *   gamma NULL: residual only (nn / mean / rstd unused).  dp NULL: scale 1.  C in {96, 192}.  Same rounding points as
 *   swin_mlp_fwd_bf16 + swin_add_layernorm_fwd. */
int swin_mlp_add_ln_fwd_bf16(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* x1,
                             const float* dp, int64_t rows_per_sample, const float* gamma, const float* beta, void* x2, void* nn,
                             float* mean, float* rstd, int64_t T, int C, float eps, void* stream);
/* swin_mlp_bwd_bf16 with the backward of norm2 and of the first residual in its epilogue (swin_transformer.py:252 differentiated):
 *   dx (T,C) = LayerNorm-backward(dn2; x1, mean, rstd, gamma) + dres (dres NULL: none);  dy = dx * dp[row / rows_per_sample] (dy NULL:
 *   not wanted; dp NULL: 1);  partials: swin_mlp_ln_bwd_partial_rows(T, C) rows of [dgamma | dbeta] (2 C floats), one per thread
 *   block -- add them with swin_tail_reduce (kind 0).  dn2 itself is not stored.  h / dhpre as in swin_mlp_bwd_bf16.  C in {96, 192}. */
int64_t swin_mlp_ln_bwd_partial_rows(int64_t T, int C);
int swin_mlp_ln_bwd_bf16(const void* x, const void* dy2, const void* w1, const float* b1, const void* w2, void* h, void* dhpre,
                         const void* x1, const float* mean, const float* rstd, const float* gamma, const void* dres, const float* dp,
                         int64_t rows_per_sample, void* dx, void* dy, float* partials, int64_t T, int C, void* stream);
/* ... and with the backward of the block's NEXT norm and second residual as its prologue: the MLP half of a block's backward in one
 * launch.  dx1 (T,C) = LayerNorm-backward(dnn; x2, mean3, rstd3, gamma3) + dres3 is written (and consumed by the epilogue as its
 * residual gradient); dy2 = dx1 * dp1[...] is written when dp1 != NULL (the fc2 weight gradient's operand; else it equals dx1);
 * partials3: [dgamma3 | dbeta3] rows like `partials`. */
int swin_mlp_ln2_bwd_bf16(const void* x, const void* w1, const float* b1, const void* w2, void* h, void* dhpre, const void* x1,
                          const float* mean, const float* rstd, const float* gamma, const float* dp, int64_t rows_per_sample, void* dx,
                          void* dy, float* partials, const void* dnn, const void* x2, const float* mean3, const float* rstd3,
                          const float* gamma3, const void* dres3, const float* dp1, void* dx1, void* dy2, float* partials3, int64_t T,
                          int C, void* stream);
/* Token-stationary Linear layers of the attention branch for C in {96, 128, 192, 256} (csrc/ts_linear.hip; a wave owns 32 tokens):
 *   swin_ts_linear_bf16:      y (T,N) = [relu](x (T,C) w (N,C)^T + bias (N, 16-bit or NULL)), N % 64 == 0  -- the qkv projection
 *                             (swin_transformer.py:129), the FPN laterals of the narrow stages (fpn.py:171-174), the mask head's
 *                             2x2 deconvolution as a GEMM with its ReLU (fcn_mask_head.py:122-126)
 *   swin_ts_proj_add_ln_bf16: x1 = x + dp[row / rows_per_sample] * (o w^T + bias);  n2 = LayerNorm(x1)  -- proj, residual, DropPath
 *                             scale and norm2 (swin_transformer.py:150-151, 252-253) in one launch; mean / rstd (T) f32 for the backward.
 * Same rounding points as the library GEMM + swin_add_layernorm_fwd chain.  Other C: SWIN_ERR_UNSUPPORTED. */
int swin_ts_linear_bf16(const void* x, const void* w, const void* bias, void* y, int64_t T, int N, int C, int relu, void* stream);
int swin_ts_proj_add_ln_bf16(const void* o, const void* w, const void* bias, const void* x, const float* dp, int64_t rows_per_sample,
                             const float* gamma, const float* beta, void* x1, void* n2, float* mean, float* rstd, int64_t T, int C,
                             float eps, void* stream);

/* nms_prepare_sorted_batch / nms_gather_dets: the front and back end of mmcv.ops.batched_nms (rpn_head.py:233) around
 *   nms_sorted_batch for `batch` images of n candidates each (n <= 16384), one launch each: boxes (batch,n,4) f32, scores
 *   (batch,n) f32, idxs (batch,n) i64 -> boxes_sorted = boxes + idxs * (max coordinate of the image + 1) in stable
 *   descending-score order, order (batch,n) i32 = source index of every sorted slot; then dets (batch,kept_cap,5) =
 *   [box, score] of kept_pos's entries (-1: zero row), valid (batch,kept_cap) u8.  workspace: nms_prepare_workspace_bytes. */
int64_t nms_prepare_workspace_bytes(int batch, int64_t n);
int nms_prepare_sorted_batch(const float* boxes, const float* scores, const int64_t* idxs, int batch, int64_t n,
                             float* boxes_sorted, int32_t* order, void* workspace, void* stream);
int nms_gather_dets(const float* boxes, const float* scores, const int32_t* order, const int32_t* kept_pos, int batch,
                    int64_t n, int kept_cap, float* dets, uint8_t* valid, void* stream);

/* det_map_roi_levels: SingleRoIExtractor.map_roi_levels (single_level_roi_extractor.py:32-51): rois (K,5) f32 -> out (K) i32 =
 *   clamp(floor(log2(sqrt(w h) / finest_scale + 1e-6)), 0, num_levels-1); rows with valid[k] == 0 (valid may be NULL) get -1. */
int det_map_roi_levels(const float* rois, const uint8_t* valid, int64_t K, int num_levels, float finest_scale, int* out,
                       void* stream);

/* det_rpn_topk_decode: RPNHead._get_bboxes proposal selection (rpn_head.py:126-187) for all images and levels in one
 *   launch -- sigmoid, per-level top-`nms_pre` by (score descending, anchor index ascending: what a stable sort keeps,
 *   rpn_head.py:162-169), gather of deltas / anchors, DeltaXYWHBBoxCoder.decode with max_shape (:185-186), level ids
 *   (:174-180).  cls (B,total) / reg (B,total,4) f32|bf16 in (level,h,w,a) order; anchors (total,4) f32; level_sizes: HOST
 *   array.  Outputs (B, sum_l min(n_l, nms_pre)), each level's survivors in ascending anchor order (callers sort by
 *   score, stably, exactly as batched_nms does): scores f32, boxes f32 (B,.,4), ids i64.  workspace: 4 B per logit. */
int64_t det_rpn_topk_decode_workspace_bytes(int64_t B, int64_t total_anchors);
int det_rpn_topk_decode(const void* cls, const void* reg, const float* anchors, const int* level_sizes, int num_levels,
                        int64_t B, int nms_pre, const float* means, const float* stds, float max_h, float max_w,
                        void* workspace, float* out_scores, float* out_boxes, int64_t* out_ids, int dtype, void* stream);

/* ---- batch normalisation over channel-last rows (csrc/batchnorm.hip): the SyncBN of the Cascade configs' ConvFCBBoxHead
 *   (convfc_bbox_head.py:99-107, norm_cfg=dict(type='SyncBN')); x (R, C) f32|bf16, C % (16 / elt size) == 0.
 *   det_bn_stats: sums (2C+1) = {sum x, sum x^2 per channel, R}; the caller all-reduces sums over ranks for SyncBN.
 *   det_bn_finalize: mean_invstd (2C) from sums; running_mean/var (C, may be NULL) updated with `momentum` (unbiased var).
 *   det_bn_apply: y = (x - mean) * invstd * gamma + beta, ReLU when relu != 0.
 *   det_bn_bwd_reduce: sums (2C) = {sum dy', sum dy' xhat} of this rank (= dbeta, dgamma), dy' = dy masked by the ReLU.
 *   det_bn_bwd_apply: dx from the (all-reduced) sums and the global row count (DEVICE pointer to one float).
 *   workspace: det_bn_workspace_bytes(C) bytes. */
int64_t det_bn_workspace_bytes(int C);
int det_bn_stats(const void* x, int64_t R, int C, float* sums, void* workspace, int dtype, void* stream);
int det_bn_finalize(const float* sums, int C, float eps, float momentum, float* mean_invstd, float* running_mean,
                    float* running_var, void* stream);
int det_bn_apply(const void* x, void* y, int64_t R, int C, const float* mean_invstd, const float* gamma, const float* beta,
                 int relu, int dtype, void* stream);
int det_bn_bwd_reduce(const void* x, const void* dy, int64_t R, int C, const float* mean_invstd, const float* gamma,
                      const float* beta, int relu, float* sums, void* workspace, int dtype, void* stream);
int det_bn_bwd_apply(const void* x, const void* dy, void* dx, int64_t R, int C, const float* mean_invstd, const float* gamma,
                     const float* beta, int relu, const float* sums, const float* count, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SWIN_HIP_H */
