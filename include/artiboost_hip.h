/* artiboost_hip.h -- C ABI of libartiboost_hip.so (hand-written HIP kernels for gfx950 / MI355X).
 *
 * The reference (lixiny/ArtiBoost) is 100 % Python and has no FFI of its own; every "kernel" it runs lives in a
 * third-party wheel (cuDNN via torch, pyrender/OpenGL, manotorch).  Each entry point below therefore names the
 * reference *call site* whose arithmetic it replaces (paths relative to the reference root).  INTEGRATION.md shows
 * the ctypes stub a maintainer would add on the reference side.
 *
 * Conventions
 *  - plain pointers + sizes only; all pointers are DEVICE pointers unless the name ends in _host
 *  - every function enqueues work on `stream` (a hipStream_t passed as void*) and returns immediately
 *  - return value: 0 on success, >0 = hipError_t from the launch, <0 = argument error (AB_E*)
 *  - no global state; re-entrant per stream
 *  - activations are NHWC ("pixels x channels"); `dtype`: 0 = float32, 1 = bfloat16 (raw uint16 bits)
 */
#ifndef ARTIBOOST_HIP_H
#define ARTIBOOST_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AB_DT_F32 0
#define AB_DT_BF16 1
#define AB_DT_U8N 2         /* loaders only (`out_dtype` of ab_render_batch / ab_augment_batch): the padded image as ONE bf16 plane of the odd integers
                             * 2 v - 255 (v = the uint8 pixel after the jitter chain); the network input v / 255 - 0.5 (hodata.py:446) is that plane / 510.
                             * ab_conv2d_stem_fwd_x3 / _stem_wgrad_x3 take it as xpad_hi with xpad_lo == NULL: two MFMA passes, no rounding of the image */
#define AB_EINVAL (-1)
#define AB_ESHAPE (-2)
#define AB_EALIGN (-3)

/* library / device info: returns ABI version (integer, bumped on any signature change) */
int ab_abi_version(void);

/* measurement aid (no reference counterpart; bench.py's roofline leg): a one-thread launch that writes the device's constant-rate wall clock to
 * *slot (device int64); ab_wall_clock_khz(): its rate.  Captured into a hipGraph around a kernel, the slot difference is that kernel's
 * duration inside the replay.                                                                                                          */
int ab_wall_stamp(int64_t* slot, void* stream);
int ab_wall_clock_khz(void);

/* ---- M3: fused softmax + 3-D integral (soft-argmax) head ------------------------------------------------------
 * replaces: norm_heatmap('softmax') + max + renorm + view_to_bcdhw + integral_heatmap3d
 *           anakin/models/simplebaseline.py:16-40, 43-71, 183-189
 * logits : [B, H, W, C*DP] (NHWC of the reference's (B, C*D, H, W); channel = c*DP + d, d < D; DP >= D is the
 *          padded depth pitch -- the model uses DP = 32 so every class is one aligned 64/128-byte run), f32|bf16
 * part   : workspace, float [B, ntile, C, 8]  (ntile from ab_softargmax3d_ntiles)
 * uvd    : float [B, C, 3]  (u = width, v = height, d = depth, each in [0,1))
 * conf   : float [B, C]     (max softmax probability)
 * stat   : float [B, C, 2]  (global max, sum exp(x - max)) kept for backward                                  */
int ab_softargmax3d_ntiles(int H, int W);
int ab_softargmax3d_fwd(const void* logits, int dtype, int B, int C, int D, int DP, int H, int W,
                        float* part, float* uvd, float* conf, float* stat, void* stream);
/* dlogits[b,h,w,c*D+d] = p * ( g_uvd . (coord - uvd) ) / (1+1e-7) + g_conf * conf * (argmax? 1 : 0 - p)
 * g_conf may be NULL.  dlogits has the dtype/layout of logits (may alias logits: in-place is allowed).        */
int ab_softargmax3d_bwd(const void* logits, int dtype, int B, int C, int D, int DP, int H, int W,
                        const float* uvd, const float* conf, const float* stat,
                        const float* g_uvd, const float* g_conf, void* dlogits, void* stream);
/* fp32 logits in, dlogits out as split-bf16 planes (what the final layer's bf16x3 data / weight gradients consume)      */
int ab_softargmax3d_bwd_x3(const float* logits, int B, int C, int D, int DP, int H, int W, const float* uvd,
                           const float* conf, const float* stat, const float* g_uvd, const float* g_conf,
                           void* dl_hi, void* dl_lo, void* stream);
/* ... and dbias [C*DP] = column sums of dlogits over all B*H*W rows: the bias gradient of the final 1x1 layer that produced the
 * logits (anakin/models/simplebaseline.py:148, final_layer), reduced in the same pass in fixed order.
 * colpart: ab_softargmax3d_bwd_x3_bias_rows(B, H, W) x C*DP floats of scratch.                                              */
int ab_softargmax3d_bwd_x3_bias_rows(int B, int H, int W);
int ab_softargmax3d_bwd_x3_bias(const float* logits, int B, int C, int D, int DP, int H, int W, const float* uvd,
                                const float* conf, const float* stat, const float* g_uvd, const float* g_conf, void* dl_hi,
                                void* dl_lo, float* colpart, float* dbias, void* stream);
/* NORM_TYPE of IntegralDeconvHead (anakin/models/simplebaseline.py:16-40): norm_type 0 = "softmax" (the entry points above),
 * 1 = "sigmoid": w = sigmoid(x), conf = max w, uvd = sum(w * coord) / (sum(w) + 1e-7) -- evaluated in the log domain with the same
 * kernels (x' = log sigmoid(x); `stat` holds (log w_max, sum w / w_max)).  g_conf must be NULL for the sigmoid head.
 * ("divide_sum" is not implemented: raw weights can be negative, and the reference's own docstring advises against it.)
 * _bwd_x3_norm: dlogits as split planes; colpart + dbias non-NULL: also the bias gradient (scratch as for _bwd_x3_bias). */
int ab_softargmax3d_fwd_norm(const void* logits, int dtype, int B, int C, int D, int DP, int H, int W, int norm_type, float* part,
                             float* uvd, float* conf, float* stat, void* stream);
int ab_softargmax3d_bwd_norm(const void* logits, int dtype, int B, int C, int D, int DP, int H, int W, int norm_type, const float* uvd,
                             const float* conf, const float* stat, const float* g_uvd, const float* g_conf, void* dlogits,
                             void* stream);
int ab_softargmax3d_bwd_x3_norm(const float* logits, int B, int C, int D, int DP, int H, int W, int norm_type, const float* uvd,
                                const float* conf, const float* stat, const float* g_uvd, void* dl_hi, void* dl_lo, float* colpart,
                                float* dbias, void* stream);

/* ---- M1/M2: convolution stack (implicit GEMM on MFMA) ----------------------------------------------------------
 * replaces cuDNN behind nn.Conv2d / nn.ConvTranspose2d / nn.Linear:
 *   anakin/models/resnet.py:154 (stem), :41-44 conv3x3 in BasicBlock (:85-101), :181-184 (1x1 downsample)
 *   anakin/models/simplebaseline.py:161-170 (ConvTranspose2d 4x4 s2 p1), :95-101 (final 1x1 conv + bias)
 *   anakin/models/mlp.py:15-22 (Linear + ReLU)
 * Layouts: activations NHWC; weights "OHWI" = [Cout][kh][kw][Cin] (K-contiguous) in the activation dtype;
 * data-gradient weights "IHWO" = [Cin][kh][kw][Cout].  Cin (fwd) / Cout (dgrad) must be a multiple of 32 (bf16)
 * or 16 (f32).  stats (optional): float [ab_conv2d_stat_rows(...)][Cout][2] per-tile (sum, sum^2) partials of
 * the OUTPUT for the following training-mode BatchNorm (finalised by ab_bn_finalize).                           */
int ab_conv2d_stat_rows(int dtype, int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int stem);
int ab_conv2d_fwd(const void* x, const void* w, void* y, int dtype, int N, int H, int W, int Cin, int Cout,
                  int kh, int kw, int stride, int pad, const float* bias, float* stats, int relu, void* stream);
/* stem 7x7/2 pad 3 on a zero-bordered NHWC4 image [N, H+6, W+8, 4] (see ab_image_pad_nhwc4); w [Cout][7][8][4]  */
int ab_conv2d_stem_fwd(const void* xpad, const void* w, void* y, int dtype, int N, int H, int W, int Cout,
                       float* stats, void* stream);
/* dx of conv2d(x,w,stride,pad) (also == ConvTranspose2d forward).  H,W,Cin describe dx.  addend (optional, dx-shaped)
 * is added in the epilogue.  stats must be NULL.                                                                 */
/* rows of BatchNorm partials [rows][Cin][2] ab_conv2d_dgrad writes when `stats` is given (the forward of a transposed
 * convolution is this data gradient: simplebaseline.py:95-101); 0: not available for the shape, pass stats = NULL. */
int ab_conv2d_dgrad_stat_rows(int dtype, int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad);
int ab_conv2d_dgrad(const void* dy, const void* wt, void* dx, int dtype, int N, int H, int W, int Cin, int Cout,
                    int kh, int kw, int stride, int pad, const void* addend, float* stats, void* stream);
/* dw (float, OHWI) of conv2d; workspace of ab_conv2d_wgrad_workspace(N*Ho*Wo, Cout, kh*kw*Cin) bytes           */
long ab_conv2d_wgrad_workspace(int M, int Cout, int jtot);
int ab_conv2d_wgrad(const void* x, const void* dy, float* dw, int dtype, int N, int H, int W, int Cin, int Cout,
                    int kh, int kw, int stride, int pad, void* workspace, int accumulate, void* stream);
/* Deferred slab reduction: the *_deferred variants run the slab kernel and record the fixed-order reduction they would
 * launch; ab_wgrad_reduce_batch runs the recorded reductions of many layers in one launch (bit-identical results).  The
 * workspace of a deferred call must stay untouched until its descriptor has been consumed.                          */
#define AB_WGRAD_BATCH_MAX 48
typedef struct ab_wgrad_reduce_desc {
    const float* slabs;    /* [nslices][slab_elems] partial weight gradients */
    float* dst;            /* dW */
    long slab_elems;
    int nslices;           /* 0: nothing pending */
    int src_j, dst_j;      /* row lengths of a slab row and of the destination row (padded taps dropped) */
    int accumulate, stem_mask;
} ab_wgrad_reduce_desc;
int ab_conv2d_wgrad_deferred(const void* x, const void* dy, float* dw, int dtype, int N, int H, int W, int Cin, int Cout,
                             int kh, int kw, int stride, int pad, void* workspace, int accumulate,
                             ab_wgrad_reduce_desc* pending, void* stream);
int ab_conv2d_stem_wgrad_deferred(const void* xpad, const void* dy, float* dw, int dtype, int N, int H, int W, int Cout,
                                  void* workspace, ab_wgrad_reduce_desc* pending, void* stream);
int ab_conv2d_wgrad_x3_deferred(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* dw, int N, int H,
                                int W, int Cin, int Cout, int kh, int kw, int stride, int pad, void* workspace, int accumulate,
                                ab_wgrad_reduce_desc* pending, void* stream);
int ab_conv2d_stem_wgrad_x3_deferred(const void* xpad_hi, const void* xpad_lo, const void* dy_hi, const void* dy_lo, float* dw,
                                     int N, int H, int W, int Cout, void* workspace, ab_wgrad_reduce_desc* pending, void* stream);
int ab_wgrad_reduce_batch(const ab_wgrad_reduce_desc* desc, int n, void* stream);
/* Grouped weight gradient (round 6): G <= AB_WGRAD_GROUP_MAX convolutions of ONE shape (3x3 / stride 1 / pad 1, split-bf16 operands: the
 * layers of a ResNet stage, anakin/models/resnet.py:85-101,158-161) as one slab launch + one reduction launch.  items_host[p]: the planes of
 * x [N,H,W,Cin] and dy [N,H,W,Cout] and the destination dW [Cout,3,3,Cin] fp32 of problem p (device pointers in a HOST array).  Equal to G
 * ab_conv2d_wgrad_x3 calls up to the (fixed) summation order over pixel slices.  workspace >= ab_conv2d_wgrad_x3_group_workspace(...) bytes
 * (0: shape not handled: call ab_conv2d_wgrad_x3 per layer).                                                                              */
#define AB_WGRAD_GROUP_MAX 8
typedef struct ab_wgrad_group_item {
    const void* x_hi; const void* x_lo; const void* dy_hi; const void* dy_lo;
    float* dw;
} ab_wgrad_group_item;
long ab_conv2d_wgrad_x3_group_workspace(int G, int N, int H, int W, int Cin, int Cout);
int ab_conv2d_wgrad_x3_group(const ab_wgrad_group_item* items_host, int G, int N, int H, int W, int Cin, int Cout, void* workspace,
                             int accumulate, void* stream);
long ab_conv2d_stem_wgrad_workspace(int N, int H, int W, int Cout);
int ab_conv2d_stem_wgrad(const void* xpad, const void* dy, float* dw, int dtype, int N, int H, int W, int Cout,
                         void* workspace, void* stream);

/* ---- M1/M2 at the reference's precision: split-bf16 ("bf16x3") convolutions ------------------------------------
 * The reference runs these convolutions in fp32 (train/train_artiboost.py:39-41,91-96 -- no autocast; cuDNN fp32 behind
 * anakin/models/resnet.py:41-44,154,181-184 and simplebaseline.py:95-101,161-170).  Every fp32 operand v is given as
 * two bf16 planes hi = bf16(v), lo = bf16(v - hi) (ab_split_f32 makes them) and a product is hi*hi + hi*lo + lo*hi on the
 * bf16 MFMA with fp32 accumulation: 2^-17 operand precision instead of bf16's 2^-9, at 1/3 of the bf16 matrix peak
 * (the f32-input MFMA runs at 1/16).  Outputs, addends, BatchNorm partials and weight gradients are fp32.
 * Same layouts and meanings as ab_conv2d_fwd / _dgrad / _wgrad; Cin (fwd) / Cout (dgrad) % 32 == 0, wgrad: both % 64.
 * w_lo must lie 0 .. 2^31-1 bytes after w_hi (both planes of one allocation).                                       */
int ab_split_f32(const float* src, long n, void* hi, void* lo, void* stream);          /* n % 8 == 0 */
/* rows of BatchNorm partials a forward WITHOUT bias / relu writes for this shape (the specialised 3x3/s1, 3x3/s2 and 4x4/s2 kernels write one
 * row per tile of theirs).  ab_conv2d_fwd_x3 with stats AND a bias or ReLU runs the generic kernel, whose row count may differ on those
 * shapes: it then returns AB_EINVAL instead of writing past a buffer sized by this function.                                            */
int ab_conv2d_x3_stat_rows(int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad);
int ab_conv2d_fwd_x3(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, float* y, int N, int H, int W,
                     int Cin, int Cout, int kh, int kw, int stride, int pad, const float* bias, float* stats, int relu,
                     void* stream);
/* The final layer of IntegralDeconvHead and the first stage of its soft-argmax in ONE launch -- replaces final_layer (nn.Conv2d(256,
 * NCLASSES * DEPTH, 1): anakin/models/simplebaseline.py:95-101,173-175) followed by the softmax statistics of norm_heatmap / integral_heatmap3d
 * (:16-40, 43-71, 183-189).  x (hi, lo) [B,H,W,Cin]; w (hi, lo) [C*32][Cin] (channel = c*32 + d, d < D valid: DEPTH_PITCH 32, padding bins
 * carry zero weights); bias [C*32] or NULL; logits fp32 [B,H,W,C*32]; part fp32 [B, H*W/64, C, 8] = the rows ab_softargmax3d_stage2 merges
 * into uvd / conf / stat (same meaning as ab_softargmax3d_fwd's workspace).  _ok(): 1 when the register-resident GEMM takes the shape
 * (Cin % 64 == 0, Cin <= 256, H*W % 64 == 0, D <= 32); otherwise AB_ESHAPE and the caller runs ab_conv2d_fwd_x3 + ab_softargmax3d_fwd.  */
int ab_conv1x1_sam_fwd_x3_ok(int B, int H, int W, int Cin, int C, int D);
int ab_conv1x1_sam_fwd_x3(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias, float* logits,
                          int B, int H, int W, int Cin, int C, int D, float* part, void* stream);
int ab_softargmax3d_stage2(const float* part, int B, int C, int ntile, float* uvd, float* conf, float* stat, void* stream);
/* Eval-mode BasicBlock convolutions (anakin/models/resnet.py:85-101 under model.eval(): train/submit_reload.py:26-79, the TEST
 * pass of train/train_artiboost.py:224-240): 3x3 / stride 1 / pad 1 convolution with the BatchNorm that follows folded into the
 * epilogue -- bnp = [scale Cout | shift Cout | ..] from ab_bn_eval_params; out = relu?(conv * scale + shift + residual) as (hi, lo)
 * planes (+ fp32 copy when out_f32 != NULL); residual = (res_hi, res_lo) planes, or fp32 res_f32, or none.  _ok(): 1 when the
 * shape is taken (otherwise run ab_conv2d_fwd_x3 + ab_bn_apply_x3: same bits). */
int ab_conv2d_fwd_x3_evalbn_ok(int N, int H, int W, int Cin, int Cout);
int ab_conv2d_fwd_x3_evalbn(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, int N, int H, int W, int Cin,
                            int Cout, const float* bnp, const void* res_hi, const void* res_lo, const float* res_f32, int relu,
                            void* out_hi, void* out_lo, float* out_f32, void* stream);
/* ... and of the generic convolutions (strided 3x3, 1x1 downsample: resnet.py:85-101,181-184) / the head's ConvTranspose2d
 * (simplebaseline.py:161-172; dgrad form as ab_conv2d_dgrad_x3): out = relu?(conv * scale + shift) as fp32 `out_f32` OR as planes
 * (out_hi, out_lo) -- exactly one of the two; bit-identical to the conv followed by ab_bn_apply / ab_bn_apply_x3. */
int ab_conv2d_fwd_x3_affine(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, int N, int H, int W, int Cin,
                            int Cout, int kh, int kw, int stride, int pad, const float* scale, const float* shift, int relu,
                            float* out_f32, void* out_hi, void* out_lo, void* stream);
int ab_conv2d_dgrad_x3_affine(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, int N, int H, int W, int Cin,
                              int Cout, int kh, int kw, int stride, int pad, const float* scale, const float* shift, int relu,
                              float* out_f32, void* out_hi, void* out_lo, void* stream);
int ab_conv2d_dgrad_x3_stat_rows(int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad);
int ab_conv2d_dgrad_x3(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, float* dx, int N, int H,
                       int W, int Cin, int Cout, int kh, int kw, int stride, int pad, const float* addend, float* stats,
                       void* stream);
/* The two data gradients that leave the input of a down-sampling BasicBlock (anakin/models/resnet.py:85-101 backwards: conv1 3x3/s2
 * and downsample.0 1x1/s2 read the same x) in one launch: dx = dgrad(dy, wt; kh x kw / 2 / pad) + dgrad(dy2, wt2; 1x1 / 2 / 0)
 * [+ addend].  dy2 planes [N,H/2,W/2,Cout], wt2 planes [Cin][1][1][Cout]; kh, kw <= 3, H, W even.                            */
int ab_conv2d_dgrad_x3_pair(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, const void* dy2_hi,
                            const void* dy2_lo, const void* wt2_hi, const void* wt2_lo, float* dx, int N, int H, int W, int Cin,
                            int Cout, int kh, int kw, int pad, const float* addend, void* stream);
/* ab_conv2d_dgrad_x3_pair whose result arrives at relu(bn(bn_y) [+ residual]) -- the output of the stage below (anakin/models/resnet.py:85-101,
 * 178-192 backwards): dz = the MASKED gradient, bn_part[rows][Cin][2] = per-tile (sum dz, sum dz*xhat) for ab_bn_bwd_x3(part, rows); mask as
 * in ab_conv2d_dgrad_x3_bn.  rows = ab_conv2d_dgrad_x3_pair_bn_rows(...); 0 = not handled (ab_conv2d_dgrad_x3_pair + the full ab_bn_bwd_x3). */
int ab_conv2d_dgrad_x3_pair_bn_rows(int N, int H, int W, int Cin, int Cout, int kh, int kw, int pad);
int ab_conv2d_dgrad_x3_pair_bn(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, const void* dy2_hi,
                               const void* dy2_lo, const void* wt2_hi, const void* wt2_lo, float* dz, int N, int H, int W, int Cin,
                               int Cout, int kh, int kw, int pad, const float* bn_y, const void* bn_out_hi, const float* bnp,
                               float* bn_part, void* stream);
/* Data gradient of a 3x3/s1 conv whose result arrives at relu(bn(bn_y) [+ residual]) (BasicBlock, anakin/models/resnet.py:85-101,
 * backwards): dz = the MASKED gradient (mask: sign of bn_out_hi, the hi bf16 plane of the stored activation, or recomputed from
 * bn_y and bnp when bn_out_hi is NULL), bn_part[rows][Cin][2] = per-tile (sum dz, sum dz*xhat) for ab_bn_bwd_x3(part, rows).
 * rows = ab_conv2d_dgrad_x3_bn_rows(...); 0 = shape not handled here (use ab_conv2d_dgrad_x3 + the full ab_bn_bwd_x3).  Also taken: 1x1 / s1 / p0
 * with addend = bn_out_hi = NULL (final_layer of the head, simplebaseline.py:173-175 backwards, arriving at relu(bn(deconv_layers.3 output))).  */
int ab_conv2d_dgrad_x3_bn_rows(int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad);
int ab_conv2d_dgrad_x3_bn(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, float* dz, int N, int H,
                          int W, int Cin, int Cout, int kh, int kw, int stride, int pad, const float* addend,
                          const float* bn_y, const void* bn_out_hi, const float* bnp, float* bn_part, void* stream);
/* workspace: ab_conv2d_wgrad_x3_workspace(...) bytes (the split-bf16 kernels slice the pixels differently from the bf16 ones) */
long ab_conv2d_wgrad_x3_workspace(int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad);
int ab_conv2d_wgrad_x3(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, float* dw, int N, int H,
                       int W, int Cin, int Cout, int kh, int kw, int stride, int pad, void* workspace, int accumulate,
                       void* stream);

/* stem 7x7/2 on split planes of the zero-bordered NHWC4 image (ab_conv2d_stem_fwd / _wgrad at bf16x3 precision)        */
int ab_conv2d_stem_x3_stat_rows(int N, int H, int W);
int ab_conv2d_stem_fwd_x3(const void* xpad_hi, const void* xpad_lo, const void* w_hi, const void* w_lo, float* y, int N, int H,
                          int W, int Cout, float* stats, void* stream);
int ab_conv2d_stem_wgrad_x3(const void* xpad_hi, const void* xpad_lo, const void* dy_hi, const void* dy_lo, float* dw, int N,
                            int H, int W, int Cout, void* workspace, void* stream);
/* fp32 elementwise passes that write a convolution operand directly as split planes (no separate ab_split_f32 pass):
 * ab_bn_apply_x3 = ab_bn_apply on fp32 with out (optional fp32 copy, may be NULL) + planes; ab_bn_bwd_x3 = ab_bn_bwd /
 * ab_bn_bwd_apply on fp32 with dy as planes (nparts_given = 0: run the reduction into `part`; > 0: `part` holds that many
 * already reduced rows).  resnet.py:85-101, simplebaseline.py:171-172.                                             */
int ab_bn_apply_x3(const float* y, const float* res, const float* bnp, long M, int C, int relu, float* out, void* out_hi,
                   void* out_lo, void* stream);
/* ... the residual given as the raw output res_y of the block's downsample conv and ITS BatchNorm parameters res_bnp
 * (anakin/models/resnet.py:95-99: identity = downsample(x)): out = [relu]( bn(y) + bn_ds(res_y) ), bn_ds(res_y) never stored. */
int ab_bn_apply_x3_resbn(const float* y, const float* res_y, const float* bnp, const float* res_bnp, long M, int C, int relu,
                         float* out, void* out_hi, void* out_lo, void* stream);
/* ... the residual given as its (hi, lo) bf16 planes (a block input that exists only as the planes its producer wrote). */
int ab_bn_apply_x3_respl(const float* y, const void* res_hi, const void* res_lo, const float* bnp, long M, int C, int relu,
                         float* out, void* out_hi, void* out_lo, void* stream);
/* BatchNorm finalize + apply in ONE launch for the training forward (round 4; nn.BatchNorm2d in train mode, resnet.py:85-101 /
 * simplebaseline.py:152-175): part [nparts][C][2] = the per-tile (sum, sum of squares) of a convolution's epilogue, count = elements per
 * channel.  Writes bnp [4][C] and updates the running statistics exactly like ab_bn_finalize, then applies like ab_bn_apply_x3
 * (res fp32 or NULL), ab_bn_apply_x3_respl (res_hi / res_lo) or ab_bn_apply_x3_resbn (res + res_bnp).  Taken when
 * ab_bn_fin_apply_x3_ok(nparts, C) (nparts <= 64, C % 64 == 0); AB_ESHAPE otherwise (run ab_bn_finalize + ab_bn_apply_x3*).
 * ab_bn_bwd_x3 takes the same route internally for its finalize.                                                              */
int ab_bn_fin_apply_x3_ok(int nparts, int C);
int ab_bn_fin_apply_x3(const float* part, int nparts, long count, const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, float* bnp, const float* y, const float* res, const void* res_hi,
                       const void* res_lo, const float* res_bnp, long M, int C, int relu, float* out, void* out_hi, void* out_lo,
                       void* stream);
/* out (relu == 1): the stored activation as fp32, or -- out_is_hi_plane != 0 -- the hi plane (bf16) of its split form: the
 * ReLU mask only needs the sign, and the plane is half the bytes                                                     */
int ab_bn_bwd_x3(const float* dout, const void* out, int out_is_hi_plane, const float* y, const float* bnp, long M, int C,
                 int relu, float* part, int nparts_given, float* bwdp, float* dgamma, float* dbeta, void* dy_hi, void* dy_lo,
                 float* dz_out, void* stream);

/* ---- M1/M2: training-mode BatchNorm, ReLU, residual, pooling (HBM-bound NHWC kernels) ---------------------------
 * replaces nn.BatchNorm2d / ReLU / MaxPool2d / mean-pool: anakin/models/resnet.py:85-101,155-157,219;
 * anakin/models/simplebaseline.py:171-172.
 * bnp: float [4][C] = (gamma*invstd, beta-mean*gamma*invstd, mean, invstd).                                     */
int ab_col_stats_nparts(long M);
int ab_col_stats(const void* x, int dtype, long M, int C, float* part, void* stream);
int ab_bn_finalize(const float* part, int nparts, int C, long count, const float* gamma, const float* beta,
                   float eps, float momentum, float* running_mean, float* running_var, float* bnp, void* stream);
int ab_bn_eval_params(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                      float* bnp, void* stream);
/* ... for EVERY BatchNorm of a network in one launch (eval mode re-derives them at each forward: the running statistics may have
 * moved): desc_dev = device int32 [n][6] = (gamma, beta element offset in `flat`; running_mean, running_var offset in `stats`;
 * bnp offset in `out`; C).  max_c = the largest C. */
int ab_bn_eval_params_batch(const float* flat, const float* stats, const int32_t* desc_dev, int n, int max_c, float eps, float* out,
                            void* stream);
int ab_bn_apply(const void* y, const void* res, const float* bnp, int dtype, long M, int C, int relu, void* out,
                void* stream);
/* part: float [ab_col_stats_nparts(M)][C][2]; bwdp: float [2][C]; dz_out optional (gradient of the residual branch).
 * relu: 0 = no activation followed the BN; 1 = ReLU, mask taken from the stored activation `out` (needed when a residual
 * was added before the ReLU); 2 = ReLU, mask recomputed as y*bnp[0]+bnp[1] > 0 (`out` may be NULL and is not read).   */
int ab_bn_bwd(const void* dout, const void* out, const void* y, const float* bnp, int dtype, long M, int C, int relu,
              float* part, float* bwdp, float* dgamma, float* dbeta, void* dy, void* dz_out, void* stream);
int ab_relu_bwd(const void* dout, const void* out, int dtype, long n, void* dz, void* stream);
int ab_col_sum(const void* x, int dtype, long M, int C, float* part, float* out, void* stream);
int ab_col_sum_x3(const void* hi, const void* lo, long M, int C, float* part, float* out, void* stream);   /* of hi + lo */
int ab_add(const void* a, const void* b, int dtype, long n, void* out, void* stream);
/* idx: uint8 [N,H/2,W/2,C] winning tap (0..8, first maximum in row-major order); H,W describe the pool INPUT          */
int ab_maxpool3x3s2_fwd(const void* x, int dtype, int N, int H, int W, int C, void* out, void* idx, void* stream);
/* BatchNorm-backward reduction fused into the producer of its input gradient.  ab_conv2d_dgrad_bnstats is
 * ab_conv2d_dgrad (3x3 / stride 1 / pad 1, bf16) whose epilogue also accumulates, over the dx tile it stores, the sums
 * (sum dz, sum dz*xhat) of the BatchNorm that produced this conv's INPUT activation: bn_y = that BN's input (the conv
 * output it normalised), bnp = its (scale, shift, mean, invstd), dz = dx masked by the ReLU that followed (mask from the
 * stored activation bn_out when non-NULL -- required if a residual was added before the ReLU -- else recomputed from
 * bn_y).  bn_part: float [rows][Cin][2] with rows = ab_conv2d_dgrad_bnstats_rows(...) (0: no fused path for this shape;
 * use ab_conv2d_dgrad + ab_bn_bwd).  ab_bn_bwd_apply is the rest of ab_bn_bwd (finalize + apply) from such partials. */
int ab_conv2d_dgrad_bnstats_rows(int dtype, int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad);
int ab_conv2d_dgrad_bnstats(const void* dy, const void* wt, void* dx, int dtype, int N, int H, int W, int Cin, int Cout,
                            int kh, int kw, int stride, int pad, const void* addend, const void* bn_y, const void* bn_out,
                            const float* bnp, float* bn_part, void* stream);
int ab_bn_bwd_apply(const void* dout, const void* out, const void* y, const float* bnp, int dtype, long M, int C, int relu,
                    const float* part, int nparts, float* bwdp, float* dgamma, float* dbeta, void* dy, void* dz_out,
                    void* stream);
/* Stem fusion: out = maxpool3x3/2(relu(y*bnp[0]+bnp[1])) without materialising the activation, and its backward through
 * both BN passes (dpool: pooled gradient; part/bwdp as in ab_bn_bwd with M = N*H*W).                                 */
int ab_bn_relu_maxpool3x3s2_fwd(const void* y, const float* bnp, int dtype, int N, int H, int W, int C, void* out,
                                void* idx, void* stream);
int ab_bn_relu_maxpool3x3s2_fwd_x3(const float* y, const float* bnp, int N, int H, int W, int C, float* out, void* out_hi,
                                   void* out_lo, void* idx, void* stream);      /* fp32 + (hi, lo) bf16 planes of the pooled tensor */
int ab_bn_relu_maxpool_bwd(const void* dpool, const void* idx, const void* y, const float* bnp, int dtype, int N, int H,
                           int W, int C, float* part, float* bwdp, float* dgamma, float* dbeta, void* dy, void* stream);
/* ... for the split-bf16 path: ONE pass scatters the pooled gradient to full resolution, masks it with the recomputed ReLU and
 * reduces it (dz: fp32 [N,H,W,C] scratch; part: [ab_bn_relu_maxpool_bwd_x3_nparts(N,H,W,C)][C][2]), then finalize + a mask-free
 * apply pass that writes dy as planes.  nparts == 0: shape not handled (use ab_maxpool3x3s2_bwd + ab_bn_bwd_x3).               */
int ab_bn_relu_maxpool_bwd_x3_nparts(int N, int H, int W, int C);
int ab_bn_relu_maxpool_bwd_x3(const float* dpool, const void* idx, const float* y, const float* bnp, int N, int H, int W,
                              int C, float* part, float* bwdp, float* dgamma, float* dbeta, float* dz, void* dy_hi,
                              void* dy_lo, void* stream);
/* The same pair with the backward's BatchNorm reduction run over the POOLED elements (the masked gradient is non-zero at window
 * winners only): the forward also writes ywin [N,H/2,W/2,C] fp32, the raw conv output at each winner; the backward reduces
 * (dpool, ywin) -- a quarter of the full-resolution tensors -- into part[ab_col_stats_nparts(N*H/2*W/2)][C][2].
 * fwd: `out` may be NULL (the pooled activation as planes only).                                                            */
int ab_bn_relu_maxpool3x3s2_fwd_x3w(const float* y, const float* bnp, int N, int H, int W, int C, float* out, void* out_hi,
                                    void* out_lo, void* idx, float* ywin, void* stream);
int ab_bn_relu_maxpool_bwd_x3w(const float* dpool, const void* idx, const float* ywin, const float* y, const float* bnp, int N, int H,
                               int W, int C, float* part, float* bwdp, float* dgamma, float* dbeta, void* dy_hi, void* dy_lo,
                               void* stream);
int ab_maxpool3x3s2_bwd(const void* idx, const void* dout, int dtype, int N, int H, int W, int C, void* dx, void* stream);
int ab_avgpool_fwd(const void* x, int dtype, int N, int HW, int C, float* out, void* stream);
int ab_avgpool_bwd(const float* g, int dtype, int N, int HW, int C, void* dx, int accumulate, void* stream);
int ab_cast_f32_bf16(const float* src, long n, void* dst, void* stream);
int ab_transpose_oki(const float* src, int O, int K, int I, int dtype, void* dst, void* stream);
/* The same re-layout for many tensors in one launch.  desc_dev: DEVICE array of ntensors descriptors; tensor t owns the
 * workgroups [tile_begin, tile_begin + ceil(O/64)*K*ceil(I/64)); total_tiles = end of the last range.  Requires
 * I % 4 == 0 and O % 8 == 0 (16-byte vectors).                                                                      */
typedef struct ab_transpose_desc {
    const void* src;   /* float [O][K][I] */
    void* dst;         /* dtype [I][K][O] */
    int32_t O, K, I;
    int32_t tile_begin;
} ab_transpose_desc;
int ab_transpose_oki_batch(const ab_transpose_desc* desc_dev, int ntensors, long total_tiles, int dtype, void* stream);
/* bf16 outputs as split planes (hi at desc.dst, lo = bf16(v - hi) at desc.dst + lo_offset_elems): the IHWO weight copies of
 * the bf16x3 data gradients */
int ab_transpose_oki_batch_x3(const ab_transpose_desc* desc_dev, int ntensors, long total_tiles, long lo_offset_elems, void* stream);
int ab_image_pad_nhwc4(const float* img_nchw, int dtype, int N, int H, int W, void* out, void* stream);

/* ---- T1: global-norm clip + Adam on the flat parameter buffer ---------------------------------------------------
 * replaces torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step: train/train_artiboost.py:91-96,
 * anakin/utils/netutils.py:26-33.  part: float[1024] workspace, total_norm: float[1] (device).  hyper (optional):
 * device float[3] = {lr, 1-beta1^step, sqrt(1-beta2^step)} overriding lr/step (per-step values under graph replay). */
int ab_grad_norm(const float* grad, long n, float* part, float* total_norm, void* stream);
/* x[0..n) *= s in place: the 1 / world_size after a SUM all-reduce of the flat gradient -- the averaging torch's
 * DistributedDataParallel does around train/train_artiboost.py:88 `final_loss.backward()` (the reference wraps the model in DDP at :249-257).
 * Kept in this library because RCCL's ReduceOp.AVG kernels use packed fp32. */
int ab_scale_f32(float* x, long n, float s, void* stream);
int ab_clip_adam(float* param, const float* grad, float* m, float* v, long n, const float* total_norm,
                 float max_norm, float lr, float beta1, float beta2, float eps, int step, const float* hyper,
                 void* lp, void* stream);
/* ab_clip_adam that also refreshes the split-bf16 weight planes (hi = bf16(p), lo = bf16(p - hi)) in the same pass */
int ab_clip_adam_x3(float* param, const float* grad, float* m, float* v, long n, const float* total_norm,
                    float max_norm, float lr, float beta1, float beta2, float eps, int step,
                    const float* hyper, void* lp_hi, void* lp_lo, void* stream);

/* ---- M4 + L1-L3 + V1: fused pose assembly + criterion, forward and backward --------------------------------------
 * replaces (one wave per sample, deterministic): anakin/models/hybridbaseline.py:49-96 (uvd->xyz, 6D->R, corners),
 * anakin/criterions/jointloss.py:25-67, ordinal.py:144-227, 262-306, criterion.py:57-67 and the per-sample EPE of
 * anakin/metrics/val_metric.py:84-106.  Random draws are inputs (host RNG, reference order): views [nv,3] float,
 * pair index lists int64.  weights8_host (HOST pointer) = {LAMBDA_JOINTS_3D, LAMBDA_CORNERS_3D, LAMBDA_JOINTS_LEVEL,
 * LAMBDA_PART_LEVEL, LAMBDA_SCENE_LEVEL, LAMBDAS[JointsLoss], LAMBDAS[HandOrdLoss], LAMBDAS[SceneOrdLoss]}.
 * outputs: joints_abs [B,21,3], corners_abs [B,8,3], rotmat [B,3,3], uvd2d [B,30,3] (optional), sample_part [B,8]
 * (per-sample partial sums; [5],[6] = joint / corner EPE in mm), losses float[8] = joints_3d_loss, corners_3d_loss,
 * joint_ord_loss, part_ord_loss, scene_ord_loss, final_loss, mean EPE joints, mean EPE corners;
 * g_kp3d [B,22,3], g_box6d [B,6] = d final_loss / d input (NULL g_kp3d: forward only).                          */
/* M4 alone -- the pose assembly of an EVAL-mode forward (no targets, no criterion): anakin/models/hybridbaseline.py:49-96
 * (uvd -> xyz with the crop intrinsics, 6-D -> R, canonical corners -> camera frame, 2-D re-projection).  One launch instead of the
 * ~50 small tensor ops of the module's forward.  joints_abs [B,21,3], corners_abs [B,8,3], rotmat [B,3,3]; optional: uvd2d
 * [B,30,3] (hybridbaseline.py:80-84), joints_rel / corners_rel (minus the predicted joint `center_idx`), boxroot [B,3]. */
int ab_pose_assemble(const float* kp3d, const float* box6d, int box_stride, const float* root_joint, const float* cam_intr,
                     const float* corners_can, int B, int center_idx, float res_w, float res_h, float* joints_abs,
                     float* corners_abs, float* rotmat, float* uvd2d, float* joints_rel, float* corners_rel, float* boxroot,
                     void* stream);
int ab_pose_loss(const float* kp3d, const float* box6d, int box_stride, const float* root_joint,
                 const float* cam_intr, const float* corners_can, const float* joints_3d, const float* corners_3d,
                 const float* joints_vis, const float* corners_vis, const float* hand_views, int nvh,
                 const int64_t* j0, const int64_t* j1, int njp, const int64_t* p0, const int64_t* p1, int npp,
                 const float* scene_views, int nvs, const int64_t* s0, const int64_t* s1, int nsp, int B,
                 int center_idx, float res_w, float res_h, const float* weights8_host, float* joints_abs,
                 float* corners_abs, float* rotmat, float* uvd2d, float* sample_part, float* losses, float* g_kp3d,
                 float* g_box6d, void* stream);
/* ab_pose_loss with SymCornerLoss (anakin/criterions/symcornerloss.py:49-102, use_ho3d_ycb = False) in the same kernel:
 * loss = mean_b min_k mean((sym_k(corners_gt) - pred)^2) over the object's symmetry set, vis-masked; its value goes to
 * loss_out[0], losses[5] (final) and the corner gradients include weight * lambda * loss.  sym == NULL or K == 0: absent. */
typedef struct ab_symcorner {
    const float* R;            /* device [nobj][K][3][3] (identity-padded)                       */
    const float* t;            /* device [nobj][K][3], metres                                     */
    int32_t K;
    const int64_t* obj_idx;    /* device [B], 1-based                                             */
    const float* obj_transf;   /* device [B][4][4]                                                */
    float lambda;              /* LAMBDA_SYM_CORNERS_3D                                           */
    float weight;              /* the Criterion LAMBDA of this loss                               */
    float* loss_out;           /* device [1], may be NULL                                         */
} ab_symcorner;
int ab_pose_loss_sym(const float* kp3d, const float* box6d, int box_stride, const float* root_joint,
                     const float* cam_intr, const float* corners_can, const float* joints_3d, const float* corners_3d,
                     const float* joints_vis, const float* corners_vis, const float* hand_views, int nvh,
                     const int64_t* j0, const int64_t* j1, int njp, const int64_t* p0, const int64_t* p1, int npp,
                     const float* scene_views, int nvs, const int64_t* s0, const int64_t* s1, int nsp, int B,
                     int center_idx, float res_w, float res_h, const float* weights8_host, const ab_symcorner* sym,
                     float* joints_abs, float* corners_abs, float* rotmat, float* uvd2d, float* sample_part, float* losses,
                     float* g_kp3d, float* g_box6d, void* stream);

/* ---- R3/R4/R5: batched online synthesis (rasterise + z-buffer + shade + background + colour jitter + crop) --------
 * replaces: anakin/utils/renderer.py:101-136 (Renderer.__call__: pyrender/OpenGL draw, background putmask),
 *           anakin/utils/frender_utils.py:36-46,116-118 (per-frame vertex upload, stale normals),
 *           anakin/artiboost/render_infra.py:14-59 (render server + two queue hops per image),
 *           anakin/artiboost/rendered_dataset.py:256-270 + anakin/utils/img_augment.py:6-80 (PIL jitter, affine, to_tensor)
 * ab_scene: HOST struct of DEVICE pointers to the immutable assets (one packed table for all object meshes).
 *   The hand mesh has V_dup >= 778 render vertices (UV-seam duplicates, as a textured trimesh has): hand_faces int32 [1538,3]
 *   index them, hand_uv [V_dup,2] / hand_normals [V_dup,3] are per render vertex, hand_map int32 [V_dup] gives the MANO
 *   vertex whose position a render vertex takes (renderer.py:17-28 get_mapping, :107 update_verts); NULL = identity.
 * samples : device array of 96-byte records {int32 obj_id, hand_tex_id, bg_id, bg_x0, bg_y0, bg_w, bg_h; float light;
 *           float obj_pose[16] (row-major 4x4)}.  hand_verts: float [B,778,3] camera frame.
 * The background is the record's crop rectangle resized to the render size with cv2.resize's INTER_LINEAR fixed-point
 * arithmetic (renderer.py:136); ab_scene.bg holds RGBX texels, uint8 [nbg, bgs, bgs, 4].
 * order/factor: int32/float [B,4] colour-jitter op ids (0 brightness, 1 saturation, 2 hue, 3 contrast) and factors in
 * application order.  inv_affine: float [B,6] output-pixel-centre -> render-pixel map (PIL AFFINE data).
 * blur_radius: float [B] (device) or NULL: PIL ImageFilter.GaussianBlur radius applied to the composited render before
 * the jitter (rendered_dataset.py:257-258, radius = U(0,1) * 0.1); each radius must be < 1.41 (box radius < 1 px).
 * out_pad: zero-bordered NHWC4 [B, oh+6, ow+8, 4] in out_dtype (interior written; border must already be zero);
 * out_chw: optional float [B,3,oh,ow] (the reference's `image` tensor).  keys_out (optional) uint64 [B,H,W]:
 * depth24<<32 | face id, ~0 = background.  rgbx_out (optional) uint8 [B,H,W,4] pre-jitter render.
 * ab_scene.srgb2lin: float [256], lin2srgb: uint8 [4096] (4-byte aligned: a tile with triangles copies it to LDS as words). */
typedef struct ab_scene {
    const void* hand_faces; const void* hand_normals; const void* hand_uv; const void* hand_map; const void* hand_tex; int hts;
    const void* obj_verts; const void* obj_normals; const void* obj_uv; const void* obj_faces;
    const void* obj_vert_off; const void* obj_face_off; const void* obj_tex; int ots;
    const void* bg; int bgs; const void* srgb2lin; const void* lin2srgb;
    float fx, fy, cx, cy; int W, H;
} ab_scene;
long ab_render_workspace_bytes(int B, int W, int H, int max_faces);
int ab_render_batch(const ab_scene* scene_host, const void* samples, const float* hand_verts, const int32_t* order,
                    const float* factor, const float* inv_affine, const float* blur_radius, int B, int max_faces,
                    int ow, int oh, int out_dtype, void* out_pad, float* out_chw, void* workspace, void* keys_out,
                    void* rgbx_out, void* stream);
/* The GaussianBlur stage of ab_render_batch on its own (PIL ImageFilter.GaussianBlur on the RGB bytes of B RGBX images,
 * W and H multiples of 32): radius float [B] on the device, each < 1.41; out must not alias rgbx.                  */
int ab_gaussian_blur(const void* rgbx, int B, int W, int H, const float* radius, void* out, void* stream);
/* SURVEY section 8f-3, the real-data half of MixedDataset: the augmentation chain of HOdata.__getitem__
 * (anakin/datasets/hodata.py:336-337,435-446) for B decoded frames of one size, RGBX uint8 [B,H,W,4] on the device:
 * optional Image.FLIP_LEFT_RIGHT (flip uint8 [B] or NULL), GaussianBlur (blur_radius [B] or NULL; needs W, H % 32 == 0),
 * colour jitter (order / factor as ab_render_batch), inverse-affine nearest crop to ow x oh, to_tensor - 0.5.
 * Outputs as ab_render_batch (zero-bordered NHWC4 in out_dtype and/or float CHW).                                   */
long ab_augment_workspace_bytes(int B, int W, int H);
int ab_augment_batch(const void* rgbx, int B, int W, int H, const int32_t* order, const float* factor,
                     const float* inv_affine, const float* blur_radius, const uint8_t* flip, int ow, int oh,
                     int out_dtype, void* out_pad, float* out_chw, void* workspace, void* stream);
/* SURVEY section 8f-3, the decode in front of that chain: baseline JPEG files -> RGB(X) frames on the device, bit-identical to what
 * `Image.open(path).convert("RGB")` returns in the reference's DataLoader workers (anakin/datasets/ho3d.py:228-231, dexycb.py:226-229,
 * fhb.py:257-260: Pillow / libjpeg-turbo defaults, JDCT_ISLOW + fancy up-sampling).  Huffman decode (self-synchronising, one thread per
 * `sub_bytes` of scan data, 16 <= sub_bytes <= 128), de-quantisation + integer IDCT, chroma up-sampling and colour conversion all run on the device; the host
 * only reads the marker segments (artiboost_amd/jpeg.py) into:
 *   data   the files' bytes (device; each image's scan is addressed through its descriptor)
 *   desc   int32 [n][AB_JPEG_DESC_INTS]: 0 scan offset in data, 1 scan bytes, 2 width, 3 height, 4 components (1 | 3), 5 hmax, 6 vmax,
 *          7.. per component (h, v, quant table, DC table, AC table) x 3, 22 restart interval (MCUs), 23 first row of `segs`, 24 segments,
 *          25 first subsequence, 26 subsequences, 27 first block, 28 blocks, 29 byte offset of its sample planes in the workspace,
 *          30 output offset (pixels), 31 output row pitch (pixels), 32 MCUs per row, 33 MCU rows, 34 blocks per MCU,
 *          35 index of its 4 quantisation tables in qtabs, 36 index of its 8 Huffman tables in htabs
 *   segs   int32 [.][4] per restart interval (one per image without restart markers): byte offset in the scan, bytes, first subsequence,
 *          first block (both relative to the image)
 *   qtabs  uint16 [.][4][64] in natural (row-major) order;  htabs  uint8 [.][8][16 + 256]: DC tables 0-3, AC tables 0-3 as in the DHT segment
 * max_blocks / max_width / max_height: the largest descriptor fields 28 / 2 / 3 of the batch.
 * out: uint8, out_channels 3 (RGB) or 4 (RGBX, X = 0).  Supported files: SOF0 / SOF1, 8 bit, grey or YCbCr 4:4:4 / 4:2:2 / 4:2:0, restart
 * intervals; anything else is refused by the host parser (the caller keeps Pillow for those files, as the reference does for all).      */
#define AB_JPEG_DESC_INTS 40
/* workspace: coefficients, per-subsequence states, sample planes, the unstuffed scans, decode tables.  n_tables: sets of 8 Huffman tables in
 * htabs; data_bytes: bytes of `data`; total_segs: rows of segs; max_subseq: the largest descriptor field 26.                              */
long ab_jpeg_workspace_bytes(long total_blocks, long total_subseq, long plane_bytes, long data_bytes, long total_segs, int n, int n_tables);
int ab_jpeg_decode_batch(const void* data, const int32_t* desc, const int32_t* segs, const void* qtabs, const void* htabs, int n,
                         int n_tables, int sub_bytes, long total_blocks, long total_subseq, long plane_bytes, long data_bytes,
                         long total_segs, int max_blocks, int max_width, int max_height, int max_subseq, int out_channels, void* out,
                         void* workspace, void* stream);
/* SURVEY section 8f-3, the same decode for .png frames -- HO3D v2, the dataset of BASELINE.json's configs[1..3], stores rgb/NNNN.png
 * (anakin/datasets/ho3d.py:181, decoded at :228-231 by `Image.open(path).convert("RGB")`): the scanline reconstruction of the PNG
 * specification (filter types None / Sub / Up / Average / Paeth, what Pillow's libImaging/ZipDecode.c applies behind zlib) and the sample
 * selection of convert("RGB"), bit-identical.  The zlib inflate of the IDAT streams stays on the host (a thread pool, artiboost_amd/png.py).
 *   raw    the inflated scanlines of the n images (device): per image `height` lines of 1 filter byte + width * bpp sample bytes;
 *          the buffer must be readable for 8 bytes past the last line (samples are fetched 4 / 8 bytes at a time)
 *   desc   int32 [n][AB_PNG_DESC_INTS]: 0 / 1 byte offset of the image in raw (low / high word), 2 width, 3 height, 4 bytes per pixel
 *          (3 RGB8, 4 RGBA8, 6 RGB16, 8 RGBA16, 1 grey8), 5 byte offsets of the R, G, B samples inside a pixel as c0 | c1 << 8 | c2 << 16
 *          (16-bit samples: the high byte, as Pillow keeps), 6 output offset (pixels), 7 output row pitch (pixels)
 *   max_width / max_bpp: the largest descriptor fields 2 / 4 of the batch
 *   out    uint8, out_channels 3 (RGB) or 4 (RGBX, X = 0);  status (device int32 or NULL): bit 0 <- a filter byte above 4 was met
 * Interlaced, palette, 1/2/4-bit, 16-bit grey and grey + alpha files are refused by the host parser (the caller keeps Pillow for those).  */
#define AB_PNG_DESC_INTS 8
int ab_png_unfilter_batch(const void* raw, const int32_t* desc, int n, int max_width, int max_bpp, int out_channels, void* out, int* status,
                          void* stream);
/* Small-batch fp32 linear layers (the box-rotation MLP, anakin/models/mlp.py:11-25; nn.Linear weights [N][K]):
 *   fwd   y[M][N]  = act(x[M][K] W^T + bias)            (relu != 0: ReLU)
 *   dgrad gx[M][K] = (g[M][N] W) masked by act_out > 0   (act_out NULL: no mask); takes wt = W transposed, [K][N]
 *   wgrad dw[N][K] = g^T x, db[N] = column sums of g     (db may be NULL)      reduction lengths % 4 == 0.          */
int ab_linear_fwd(const float* x, const float* w, const float* bias, int M, int N, int K, int relu, float* y, void* stream);
int ab_linear_dgrad(const float* g, const float* wt, const float* act_out, int M, int N, int K, float* gx, void* stream);
int ab_linear_wgrad(const float* g, const float* x, int M, int N, int K, float* dw, float* db, void* stream);

/* ---- grasp refiner (anakin/artiboost/refiner.py) ------------------------------------------------
 * ab_nearest_dist replaces point2point_signed(hand_verts, verts_object) (refiner.py:21-83; nearest neighbour from the
 * third-party chamfer_distance CUDA extension) together with the rotated-object temporary of refiner.py:196-199:
 *   dist[b][i] = min_j || x[b][i] - rot[b] * y[j] ||   (+ per-vertex affine dist * scale[i] + shift[i], the eval-mode
 *   BatchNorm1d(778) of refiner.py:267, when scale != NULL), idx_out[b][i] = arg min (first minimum; optional).
 * x [B,P1,3]; ypts [nobj,P2,3] selected by obj_idx [B] (int64), or [B,P2,3] when obj_idx == NULL; rot [B,3,3] or NULL;
 * dist has row pitch ld >= P1.
 * ab_linear_fused: one layer of the RefineNet MLP (refiner.py:227-319, ResBlock :286-319), fp32:
 *   y[m][n] = act(((x W^T + bias) * scale[n] + shift[n]) + residual[m][n]), act 0 none / 1 ReLU / 2 leaky(slope);
 *   W [N][K], K % 4 == 0 (pad the features), scale/shift/residual/bias may be NULL, ldr / ldy = row pitches.        */
int ab_nearest_dist(const float* x, const float* ypts, const int64_t* obj_idx, const float* rot, int B, int P1, int P2,
                    const float* scale, const float* shift, float* dist, int ld, int32_t* idx_out, void* stream);
int ab_linear_fused(const float* x, const float* w, const float* bias, const float* scale, const float* shift,
                    const float* residual, int ldr, int M, int N, int K, int act, float slope, float* y, int ldy,
                    void* stream);
/* The colour-jitter chain of ab_render_batch on its own (anakin/utils/img_augment.py:6-80 on a PIL image): B RGBX images of
 * npix pixels, order int32 [B][4] (0 brightness, 1 saturation, 2 hue, 3 contrast), factor float [B][4]; out (RGBX, X =
 * 255) may alias rgbx; lsum_ws: B x 8 bytes of device scratch.                                                        */
int ab_color_jitter(const void* rgbx, int B, int npix, const int32_t* order, const float* factor, void* out,
                    void* lsum_ws, void* stream);

/* ---- R1: MANO linear-blend skinning ------------------------------------------------------------------------------
 * replaces manotorch.manolayer.ManoLayer.forward (third party, un-pinned git dependency, requirements.txt:178) at
 * anakin/artiboost/preprocessor.py:25,62, refiner.py:138,193,216,265, grasp_engine.py:90-95; maths as in the in-tree
 * anakin/postprocess/iknet/manolayer.py:182-276 (center_idx=None).  pose [B,48] axis-angle, betas [B,10];
 * model tables: v_template [778,3], shapedirs [778,3,10], posedirs [778,3,135], J_regressor [16,778],
 * weights [778,16], hands_mean [45].  Outputs verts [B,778,3], joints [B,21,3], T_abs [B,16,4,4] (optional).     */
int ab_mano_lbs(const float* pose, const float* betas, const float* v_template, const float* shapedirs,
                const float* posedirs, const float* J_regressor, const float* weights, const float* hands_mean,
                int B, float* verts, float* joints, float* T_abs, void* stream);

/* ---- Argument contracts of the dispatcher ops (torch.ops.artiboost_hip.*, libartiboost_torch.so) ------------------------------------------
 * The C entry points above take raw pointers and trust their caller.  Their PyTorch-dispatcher form (ops that "validate
 * with TORCH_CHECK") is generated from this header by artiboost_amd/gen_torch_ops.py and checks, before the C call, for EVERY op:
 *   - a tensor passed for a device pointer is a contiguous HIP tensor of the current device; one passed for a host pointer (`*_host`, the
 *     ab_* descriptor structs) is a contiguous CPU tensor;
 *   - a typed pointer fixes the dtype: float* float32, int32_t* / int* int32, int64_t* int64, uint8_t* uint8;
 * and, per op, the clauses of its `@check` line below (C expressions over the op's integer arguments; co() = convolution output size):
 *   names >= expr        every named tensor holds at least `expr` elements (a wrong B / H / C no longer overruns a buffer: RuntimeError)
 *   bytes names >= expr  ... at least `expr` bytes (workspaces)
 *   bf16|u8|i32|f32: names     dtype of `void*` arguments
 *   dt(code): names            dtype given by the op's AB_DT_* argument `code`
 *   strided: names             arguments the op addresses through a row-pitch argument (ld*, *_stride): non-contiguous views are accepted
 * A violated clause raises RuntimeError naming the op, the argument and both sizes; nothing is launched.
 * @check ab_softargmax3d_fwd: dt(dtype): logits; logits >= B*H*W*C*DP; part >= B*ab_softargmax3d_ntiles(H,W)*C*8; uvd >= B*C*3; conf >= B*C; stat >= B*C*2
 * @check ab_softargmax3d_bwd: dt(dtype): logits dlogits; logits dlogits >= B*H*W*C*DP; uvd g_uvd >= B*C*3; conf g_conf >= B*C; stat >= B*C*2
 * @check ab_softargmax3d_bwd_x3: bf16: dl_hi dl_lo; logits dl_hi dl_lo >= B*H*W*C*DP; uvd g_uvd >= B*C*3; conf g_conf >= B*C; stat >= B*C*2
 * @check ab_split_f32: bf16: hi lo; src hi lo >= n
 * @check ab_cast_f32_bf16: bf16: dst; src dst >= n
 * @check ab_conv2d_fwd_x3: bf16: x_hi x_lo w_hi w_lo; x_hi x_lo >= N*H*W*Cin; w_hi w_lo >= Cout*kh*kw*Cin; y >= N*co(H,kh,stride,pad)*co(W,kw,stride,pad)*Cout; bias >= Cout; stats >= ab_conv2d_x3_stat_rows(N,H,W,Cin,Cout,kh,kw,stride,pad)*Cout*2
 * @check ab_conv1x1_sam_fwd_x3: bf16: x_hi x_lo w_hi w_lo; x_hi x_lo >= B*H*W*Cin; w_hi w_lo >= C*32*Cin; bias >= C*32; logits >= B*H*W*C*32; part >= B*(H*W/64)*C*8
 * @check ab_softargmax3d_stage2: part >= B*ntile*C*8; uvd >= B*C*3; conf >= B*C; stat >= B*C*2
 * @check ab_conv2d_fwd_x3_evalbn: bf16: x_hi x_lo w_hi w_lo res_hi res_lo out_hi out_lo; x_hi x_lo >= N*H*W*Cin; w_hi w_lo >= Cout*9*Cin; bnp >= 2*Cout; res_hi res_lo res_f32 out_hi out_lo out_f32 >= N*H*W*Cout
 * @check ab_conv2d_fwd_x3_affine: bf16: x_hi x_lo w_hi w_lo out_hi out_lo; x_hi x_lo >= N*H*W*Cin; w_hi w_lo >= Cout*kh*kw*Cin; scale shift >= Cout; out_f32 out_hi out_lo >= N*co(H,kh,stride,pad)*co(W,kw,stride,pad)*Cout
 * @check ab_conv2d_dgrad_x3: bf16: dy_hi dy_lo wt_hi wt_lo; dy_hi dy_lo >= N*co(H,kh,stride,pad)*co(W,kw,stride,pad)*Cout; wt_hi wt_lo >= Cin*kh*kw*Cout; dx addend >= N*H*W*Cin
 * @check ab_conv2d_dgrad_x3_pair: bf16: dy_hi dy_lo wt_hi wt_lo dy2_hi dy2_lo wt2_hi wt2_lo; dy_hi dy_lo dy2_hi dy2_lo >= N*(H/2)*(W/2)*Cout; wt_hi wt_lo >= Cin*kh*kw*Cout; wt2_hi wt2_lo >= Cin*Cout; dx addend >= N*H*W*Cin
 * @check ab_conv2d_dgrad_x3_pair_bn: bf16: dy_hi dy_lo wt_hi wt_lo dy2_hi dy2_lo wt2_hi wt2_lo bn_out_hi; dy_hi dy_lo dy2_hi dy2_lo >= N*(H/2)*(W/2)*Cout; wt_hi wt_lo >= Cin*kh*kw*Cout; wt2_hi wt2_lo >= Cin*Cout; dz bn_y bn_out_hi >= N*H*W*Cin; bnp >= 4*Cin; bn_part >= ab_conv2d_dgrad_x3_pair_bn_rows(N,H,W,Cin,Cout,kh,kw,pad)*Cin*2
 * @check ab_conv2d_dgrad_x3_bn: bf16: dy_hi dy_lo wt_hi wt_lo bn_out_hi; dy_hi dy_lo >= N*co(H,kh,stride,pad)*co(W,kw,stride,pad)*Cout; wt_hi wt_lo >= Cin*kh*kw*Cout; dz addend bn_y bn_out_hi >= N*H*W*Cin; bnp >= 4*Cin; bn_part >= ab_conv2d_dgrad_x3_bn_rows(N,H,W,Cin,Cout,kh,kw,stride,pad)*Cin*2
 * @check ab_conv2d_wgrad_x3: bf16: x_hi x_lo dy_hi dy_lo; x_hi x_lo >= N*H*W*Cin; dy_hi dy_lo >= N*co(H,kh,stride,pad)*co(W,kw,stride,pad)*Cout; dw >= Cout*kh*kw*Cin; bytes workspace >= ab_conv2d_wgrad_x3_workspace(N,H,W,Cin,Cout,kh,kw,stride,pad)
 * @check ab_conv2d_stem_fwd_x3: bf16: xpad_hi xpad_lo w_hi w_lo; xpad_hi xpad_lo >= N*(H+6)*(W+8)*4; w_hi w_lo >= Cout*7*8*4; y >= N*(H/2)*(W/2)*Cout; stats >= ab_conv2d_stem_x3_stat_rows(N,H,W)*Cout*2
 * @check ab_conv2d_wgrad_x3_group: bytes workspace >= ab_conv2d_wgrad_x3_group_workspace(G,N,H,W,Cin,Cout)
 * @check ab_conv2d_stem_wgrad_x3: bf16: xpad_hi xpad_lo dy_hi dy_lo; xpad_hi xpad_lo >= N*(H+6)*(W+8)*4; dy_hi dy_lo >= N*(H/2)*(W/2)*Cout; dw >= Cout*7*8*4
 * @check ab_conv2d_fwd: dt(dtype): x w y; x >= N*H*W*Cin; w >= Cout*kh*kw*Cin; y >= N*co(H,kh,stride,pad)*co(W,kw,stride,pad)*Cout; bias >= Cout
 * @check ab_conv2d_dgrad: dt(dtype): dy wt dx addend; dy >= N*co(H,kh,stride,pad)*co(W,kw,stride,pad)*Cout; wt >= Cin*kh*kw*Cout; dx addend >= N*H*W*Cin
 * @check ab_bn_apply_x3: bf16: out_hi out_lo; y res out out_hi out_lo >= M*C; bnp >= 2*C
 * @check ab_bn_apply_x3_respl: bf16: res_hi res_lo out_hi out_lo; y res_hi res_lo out out_hi out_lo >= M*C; bnp >= 2*C
 * @check ab_bn_fin_apply_x3: bf16: res_hi res_lo out_hi out_lo; part >= nparts*C*2; gamma beta running_mean running_var >= C; bnp >= 4*C; res_bnp >= 2*C; y res res_hi res_lo out out_hi out_lo >= M*C
 * @check ab_bn_bwd_x3: bf16: dy_hi dy_lo; dout y dy_hi dy_lo dz_out >= M*C; bnp >= 4*C; dgamma dbeta >= C
 * @check ab_bn_finalize: part >= nparts*C*2; gamma beta running_mean running_var >= C; bnp >= 4*C
 * @check ab_bn_eval_params: gamma beta rm rv >= C; bnp >= 4*C
 * @check ab_col_stats: dt(dtype): x; x >= M*C; part >= ab_col_stats_nparts(M)*C*2
 * @check ab_bn_relu_maxpool3x3s2_fwd_x3w: bf16: out_hi out_lo; y >= N*H*W*C; bnp >= 2*C; out out_hi out_lo ywin >= N*(H/2)*(W/2)*C; bytes idx >= N*(H/2)*(W/2)*C
 * @check ab_avgpool_fwd: dt(dtype): x; x >= N*HW*C; out >= N*C
 * @check ab_image_pad_nhwc4: dt(dtype): out; img_nchw >= N*3*H*W; out >= N*(H+6)*(W+8)*4
 * @check ab_grad_norm: grad >= n; total_norm >= 1
 * @check ab_scale_f32: x >= n
 * @check ab_clip_adam: bf16: lp; param grad m v lp >= n; total_norm >= 1
 * @check ab_clip_adam_x3: bf16: lp_hi lp_lo; param grad m v lp_hi lp_lo >= n; total_norm >= 1
 * @check ab_pose_assemble: strided: box6d; kp3d >= B*22*3; root_joint >= B*3; cam_intr >= B*9; corners_can >= B*24; joints_abs joints_rel >= B*63; corners_abs corners_rel >= B*24; rotmat >= B*9; uvd2d >= B*90
 * @check ab_pose_loss: strided: box6d g_box6d; kp3d g_kp3d >= B*22*3; root_joint >= B*3; cam_intr >= B*9; corners_can corners_3d >= B*24; joints_3d >= B*63; joints_vis >= B*21; corners_vis >= B*8; hand_views >= nvh*3; scene_views >= nvs*3; j0 j1 >= njp; p0 p1 >= npp; s0 s1 >= nsp; weights8_host >= 8
 * @check ab_render_batch: bytes samples >= B*96; hand_verts >= B*778*3; order factor >= B*4; inv_affine >= B*6; blur_radius >= B; dt(out_dtype): out_pad; out_pad >= B*(oh+6)*(ow+8)*4; out_chw >= B*3*oh*ow
 * @check ab_gaussian_blur: u8: rgbx out; rgbx out >= B*W*H*4; radius >= B
 * @check ab_augment_batch: u8: rgbx; rgbx >= B*W*H*4; order factor >= B*4; inv_affine >= B*6; blur_radius flip >= B; dt(out_dtype): out_pad; out_pad >= B*(oh+6)*(ow+8)*4; out_chw >= B*3*oh*ow; bytes workspace >= ab_augment_workspace_bytes(B,W,H)
 * @check ab_color_jitter: u8: rgbx out; rgbx out >= B*npix*4; order factor >= B*4; bytes lsum_ws >= B*8
 * @check ab_jpeg_decode_batch: u8: data out; bytes data >= data_bytes; desc >= n*AB_JPEG_DESC_INTS; segs >= total_segs*4; bytes qtabs >= n*4*64*2; bytes htabs >= n_tables*8*272; bytes workspace >= ab_jpeg_workspace_bytes(total_blocks,total_subseq,plane_bytes,data_bytes,total_segs,n,n_tables)
 * @check ab_png_unfilter_batch: u8: raw out; desc >= n*AB_PNG_DESC_INTS; status >= 1
 * @check ab_linear_fwd: x >= M*K; w >= N*K; bias >= N; y >= M*N
 * @check ab_linear_dgrad: g act_out >= M*N; wt >= K*N; gx >= M*K
 * @check ab_linear_wgrad: g >= M*N; x >= M*K; dw >= N*K; db >= N
 * @check ab_nearest_dist: strided: dist; x >= B*P1*3; rot >= B*9; obj_idx >= B; scale shift >= P1; idx_out >= B*P1
 * @check ab_pose_loss_sym: strided: box6d g_box6d; kp3d g_kp3d >= B*22*3; root_joint >= B*3; cam_intr >= B*9
 * @check ab_linear_fused: strided: residual y; x >= M*K; w >= N*K; bias scale shift >= N
 * @check ab_mano_lbs: pose >= B*48; betas >= B*10; v_template >= 778*3; shapedirs >= 778*3*10; posedirs >= 778*3*135; J_regressor >= 16*778; weights >= 778*16; hands_mean >= 45; verts >= B*778*3; joints >= B*21*3; T_abs >= B*16*16
 */

#ifdef __cplusplus
}
#endif
#endif
