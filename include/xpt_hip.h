/* xpt_hip.h -- C ABI of libxpt_hip.so: the MI355X (gfx950) kernels of the
 * self-supervised depth/pose training hot path of goodgodgd/xpt-mde-2021.
 *
 * The reference has NO plugin / operator / FFI interface for this path: it is pure
 * Python on TensorFlow ops.  Each entry point below therefore cites the
 * reference Python callable (file:line, relative to the reference checkout) whose
 * arithmetic it replaces; INTEGRATION.md shows the ctypes binding.
 *
 * Conventions (all entry points)
 *   - every pointer is a DEVICE pointer to contiguous float32 owned by the caller
 *     (PyTorch's allocator in the shipped host code); nothing is allocated here,
 *     no global state, re-entrant;
 *   - images use the reference's axis order: [batch, numsrc, height, width, C]
 *     (channels last), depth [batch, height, width(,1)], pixel order row-major;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it and
 *     capturable into a hipGraph (no sync / malloc inside);
 *   - return value: 0 on success, XPT_ERR_* (< 0) on a bad argument or launch error;
 *     never throws;
 *   - reductions are deterministic: per-workgroup partial sums go to a caller-provided
 *     workspace and are summed in a fixed order (no float atomics).
 */
#ifndef XPT_HIP_H_
#define XPT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XPT_OK 0
#define XPT_ERR_NULL (-1)      /* a required pointer is NULL            */
#define XPT_ERR_SHAPE (-2)     /* non-positive or inconsistent dimension */
#define XPT_ERR_ARG (-3)       /* bad enum / flag value                  */
#define XPT_ERR_WORKSPACE (-4) /* workspace too small                    */
#define XPT_ERR_LAUNCH (-5)    /* hipGetLastError() != hipSuccess        */

/* photometric methods: model/loss_and_metric/loss_util.py:6 (L1), :29 (L2), :52 (SSIM) */
#define XPT_PHOTO_L1 0
#define XPT_PHOTO_L2 1
#define XPT_PHOTO_SSIM 2

/* library / ABI version and the gfx target the code objects were built for */
int xpt_abi_version(void);
const char* xpt_build_arch(void);

/* ------------------------------------------------------------------ K0: pose algebra
 * replaces utils/convert_pose.py:32-71 pose_rvec2matr_batch_tf (negated-skew Rodrigues).
 * pose [M,6] (tx,ty,tz,u1,u2,u3) -> T [M,4,4].  bwd: dT [M,4,4] -> dpose [M,6]. */
int xpt_pose_rvec2matr_fwd(const float* pose, float* T, int M, void* stream);
int xpt_pose_rvec2matr_bwd(const float* pose, const float* dT, float* dpose, int M, void* stream);

/* ------------------------------------------------------------------ K1 / a15: image pyramid
 * replaces tf.image.resize(bilinear, half-pixel centres, no antialias) at an exact integer
 * down-scale factor (synthesize_base.py:74-85 resize_source_images, util_funcs.py:163-175
 * multi_scale_like_depth).  img [M,H,W,C] -> out [M,H/scale,W/scale,C]; scale in {1,2,4,8,..}. */
int xpt_resize_down_fwd(const float* img, float* out, int M, int H, int W, int C, int scale, void* stream);
/* Every image the loss stage reads from a snippet batch image5d [B,S,H,W,3] (TotalLoss.append_data, losses.py:57-104:
 * sources = frames 0..S-2, target = frame S-1, each resized to every depth scale, synthesize_base.py:74-85 and
 * util_funcs.py:163-175) in one launch: job j = frames first_frame[j] .. +nframes[j]-1 of every snippet resized by the
 * exact factor scale[j] (1 = dense copy, else even) -> out[j] [B*nframes[j], H/scale[j], W/scale[j], 3]; at most 10 jobs.
 * Same values as xpt_resize_down_fwd on the dense frames. */
int xpt_image_pyramids(const float* image5d, int B, int S, int H, int W, int njobs, const int* first_frame,
                       const int* nframes, const int* scale, float* const* out, void* stream);

/* ------------------------------------------------------------------ K2+K3: view synthesis
 * replaces SynthesizeSingleScale.synthesize_batch_view (synthesize_base.py:88-178:
 * pixel_meshgrid, pixel2cam, transform_to_source, cam2pixel) fused with
 * BilinearInterpolation.__call__ (bilinear_interp.py:7-147).
 *   src   [B,N,h,w,3]  source images already resized to this scale
 *   depth [B,h,w]      target depth at this scale (0 = invalid pixel)
 *   T     [B,N,4,4]    target->source pose matrices
 *   K     [B,3,3]      UNSCALED intrinsic; rows 0,1 are divided by `scale` inside
 *                      (scale_intrinsic, synthesize_base.py:66-71)
 *   synth [B,N,h,w,3]  out
 * bwd: dsynth [B,N,h,w,3] -> ddepth [B,h,w], dT [B,N,4,4] (last row 0).
 *   workspace: xpt_warp_bwd_workspace_floats(B,N,h,w) floats. */
int xpt_warp_fwd(const float* src, const float* depth, const float* T, const float* K, float* synth,
                 int B, int N, int h, int w, float scale, void* stream);
size_t xpt_warp_bwd_workspace_floats(int B, int N, int h, int w);
int xpt_warp_bwd(const float* src, const float* depth, const float* T, const float* K, const float* dsynth,
                 float* ddepth, float* dT, float* workspace, size_t workspace_floats,
                 int B, int N, int h, int w, float scale, void* stream);

/* ------------------------------------------------------------------ K3 alone: bilinear sampler
 * replaces BilinearInterpolation.__call__(image, pixel_coords, valid_mask)
 * (bilinear_interp.py:7-32), also the sampler of FlowBilinearInterpolation (:166-181).
 *   image  [B,N,h,w,C], coords [B,N,ncoord,h*w] rows (u,v[,1]), ncoord in {2,3}
 *   valid_mask [B,h*w] or NULL (zero = invalid), out [B,N,h,w,C].
 * bwd: dout -> dcoords [B,N,ncoord,h*w] (row 2, if present, is zero).  C <= 16. */
int xpt_bilinear_fwd(const float* image, const float* coords, const float* valid_mask, float* out,
                     int B, int N, int h, int w, int C, int ncoord, void* stream);
int xpt_bilinear_bwd(const float* image, const float* coords, const float* valid_mask, const float* dout,
                     float* dcoords, int B, int N, int h, int w, int C, int ncoord, void* stream);

/* ------------------------------------------------------------------ K4/K5: photometric losses
 * replaces photometric_loss_l1 / _l2 / _ssim (loss_util.py:6-25, 29-48, 52-96).
 *   synth [B,N,h,w,3], target [B,h,w,3]
 *   map   [B,N,h,w,3] or NULL : per-pixel loss (the reduce=False result)
 *   loss  [B] or NULL         : mean over (N,h,w,3)  (the reduce=True result)
 *   workspace: xpt_photo_workspace_floats(B,N,h,w) floats (needed when loss != NULL).
 * bwd: exactly one of gloss [B] (grad of the reduce=True result) / gmap [B,N,h,w,3]
 *   (grad of the per-pixel map) is non-NULL -> dsynth [B,N,h,w,3].
 *   SSIM bwd needs workspace of xpt_photo_workspace_floats(B,N,h,w) floats. */
size_t xpt_photo_workspace_floats(int B, int N, int h, int w);
int xpt_photo_fwd(int method, const float* synth, const float* target, float* map, float* loss,
                  float* workspace, size_t workspace_floats, int B, int N, int h, int w, void* stream);
int xpt_photo_bwd(int method, const float* synth, const float* target, const float* gloss, const float* gmap,
                  float* dsynth, float* workspace, size_t workspace_floats, int B, int N, int h, int w,
                  void* stream);

/* ------------------------------------------------------------------ K2-K5 fused: warp + L1 + SSIM "march"
 * One pass per scale over (depth, target, sources) that replaces synthesize_batch_view (synthesize_base.py:88-178),
 * BilinearInterpolation (bilinear_interp.py:7-147), photometric_loss_l1 and photometric_loss_ssim
 * (loss_util.py:6-25, 52-96) WITHOUT materialising the synthesized views (pass synth = NULL) -- or also writing
 * them (synth [B,N,h,w,3]) when a caller needs the images.
 *   src [B,N,h,w,3], depth [B,h,w], T [B,N,4,4], K [B,3,3] unscaled (+ scale), target [B,h,w,3]
 *   -> loss_l1 [B], loss_ssim [B]  (the reduce=True results: means over N*h*w*3); both may be NULL, then only the
 *      per-wave partial sums are left in the workspace (used to time the main kernel alone)
 * bwd: g_l1 [B], g_ssim [B] (gradients of those means) -> ddepth [B,h,w], dT [B,N,4,4] (last row 0); the views are
 *   re-synthesized on the fly.  workspace: xpt_photo_fused_workspace_floats(B,N,h,w) floats for both directions.
 * ALGORITHMIC bytes per batch element (P = h*w): fwd P(16 + 12N) [+ 12NP with synth], bwd P(20 + 12N). */
/* launch-plan knobs (process-wide, for benchmarking): the row chunks (32, 16, 8 ... min_rows) shrink until the launch has
 * at least this many waves */
int xpt_photo_fused_tune(int fwd_min_waves, int bwd_min_waves, int min_rows);
/* forward variant (process-wide): 1 = hand-pipelined row loop (default), 0 = compiler-scheduled; same results */
int xpt_photo_fused_variant(int pipelined);
size_t xpt_photo_fused_workspace_floats(int B, int N, int h, int w);
int xpt_photo_fused_fwd(const float* src, const float* depth, const float* T, const float* K, const float* target,
                        float* synth, float* loss_l1, float* loss_ssim, float* workspace, size_t workspace_floats,
                        int B, int N, int h, int w, float scale, void* stream);
int xpt_photo_fused_bwd(const float* src, const float* depth, const float* T, const float* K, const float* target,
                        const float* g_l1, const float* g_ssim, float* ddepth, float* dT, float* workspace,
                        size_t workspace_floats, int B, int N, int h, int w, float scale, void* stream);

/* All scales of the loss pyramid in ONE march launch (+ one finishing launch) -- SynthesizeMultiScale.__call__
 * (model/synthesize/synthesize_base.py:13-20) followed by PhotometricLossMultiScale (model/loss_and_metric/losses.py:179-195)
 * over every scale.  Arrays hold nscales <= 4 entries (scale 0 first): src[s] [B,N,h_s,w_s,3], depth[s] [B,h_s,w_s],
 * target[s] [B,h_s,w_s,3], scale[s] = the divisor of the full-resolution intrinsic.  losses [2 nscales][B]: row s = the
 * photometric L1 of scale s, row nscales + s = its SSIM loss (NULL: per-wave partials only, no finishing launch).  workspace >= sum of the per-scale
 * xpt_photo_fused_workspace_floats.  Per scale the arithmetic is that of xpt_photo_fused_fwd / _bwd. */
int xpt_photo_fused_ms_fwd(int nscales, const float* const* src, const float* const* depth, const float* T, const float* K,
                           const float* const* target, float* losses, float* workspace, size_t workspace_floats,
                           int B, int N, const int* h, const int* w, const float* scale, void* stream);

/* g_l1[s] / g_ssim[s] [B] -> ddepth[s] [B,h_s,w_s]; dT [B,N,4,4] = the sum over the scales.  N must be 4 or 1. */
int xpt_photo_fused_ms_bwd(int nscales, const float* const* src, const float* const* depth, const float* T, const float* K,
                           const float* const* target, const float* const* g_l1, const float* const* g_ssim,
                           float* const* ddepth, float* dT, float* workspace, size_t workspace_floats, int B, int N,
                           const int* h, const int* w, const float* scale, void* stream);

/* Second generation of the multi-scale march launches (csrc/xpt_march.hip): same arguments, same results to fp32
 * rounding (losses 1e-6 relative, gradients 1e-5 of their scale), same workspace size as xpt_photo_fused_ms_*; the row
 * body is re-written for the instruction classes gfx950 issues at full rate and the workgroups are numbered XCD by XCD.
 * Replaces the same reference callables (synthesize_base.py:13-20, 88-178; bilinear_interp.py:7-147;
 * loss_util.py:6-25, 52-96; losses.py:179-195).  xpt_photo_march_tune: launch-plan knobs as xpt_photo_fused_tune. */
int xpt_photo_march_tune(int fwd_min_waves, int bwd_min_waves, int min_rows);
/* xpt_photo_march_plan: bwd_variant 1 = the pipelined row step of round 4 (the next row's tap gathers in flight behind the
 * SSIM-coefficient and gradient stages of the current one; the waves with the most row steps get the higher issue priority),
 * 2 = the same without wave priorities, 0 = the round-3 row step; rows_s*: rows per chunk of the
 * backward launch for pyramid scale s (0 = the automatic, occupancy-balanced choice). */
int xpt_photo_march_plan(int bwd_variant, int rows_s0, int rows_s1, int rows_s2, int rows_s3);
int xpt_photo_march_ms_fwd(int nscales, const float* const* src, const float* const* depth, const float* T, const float* K,
                           const float* const* target, float* losses, float* workspace, size_t workspace_floats,
                           int B, int N, const int* h, const int* w, const float* scale, void* stream);
/* backward AND forward values of a training step in one pass: losses [2 nscales][B] as xpt_photo_fused_ms_fwd writes them
 * (NULL: gradients only); the upstream gradients g_l1 / g_ssim must be known when it is launched (they are loss weights
 * over the batch size: TotalLoss.__call__, losses.py:44-55). */
int xpt_photo_march_ms_fwdbwd(int nscales, const float* const* src, const float* const* depth, const float* T, const float* K,
                              const float* const* target, const float* const* g_l1, const float* const* g_ssim,
                              float* losses, float* const* ddepth, float* dT, float* workspace, size_t workspace_floats,
                              int B, int N, const int* h, const int* w, const float* scale, void* stream);
int xpt_photo_march_ms_bwd(int nscales, const float* const* src, const float* const* depth, const float* T, const float* K,
                           const float* const* target, const float* const* g_l1, const float* const* g_ssim,
                           float* const* ddepth, float* dT, float* workspace, size_t workspace_floats, int B, int N,
                           const int* h, const int* w, const float* scale, void* stream);

/* ------------------------------------------------------------------ K6 (+a4): edge-aware smoothness
 * replaces SmoothenessLossMultiScale.smootheness_loss (losses.py:409-440) for one scale
 * (the caller divides by the scale, losses.py:401-402).
 *   disp [B,h,w] (or depth when input_is_depth != 0: disp = (1/d)*[d>1e-5],
 *   util_funcs.py:157-160, fused), image [B,h,w,3] -> loss [B].
 *   grad_factor = opts.IMAGE_GRADIENT_FACTOR (config-example.py:67).
 * bwd: gloss [B] -> dinput [B,h,w] (w.r.t. disp, or depth when input_is_depth). */
size_t xpt_smooth_workspace_floats(int B, int h, int w);
int xpt_smooth_fwd(const float* disp, const float* image, float* loss, float* workspace, size_t workspace_floats,
                   int B, int h, int w, float grad_factor, int input_is_depth, void* stream);
int xpt_smooth_bwd(const float* disp, const float* image, const float* gloss, float* dinput,
                   int B, int h, int w, float grad_factor, int input_is_depth, void* stream);
/* Every scale of SmoothenessLossMultiScale.__call__ (losses.py:391-407) in one forward pair / one backward launch:
 * nscales (1..4) arrays disp[s] [B,h_s,w_s], image[s] [B,h_s,w_s,3] -> losses [nscales][B] (one buffer); same values as
 * xpt_smooth_fwd / xpt_smooth_bwd per scale.  workspace_floats >= the sum of xpt_smooth_workspace_floats(B,h_s,w_s). */
int xpt_smooth_ms_fwd(int nscales, const float* const* disp, const float* const* image, float* losses,
                      float* workspace, size_t workspace_floats, int B, const int* h, const int* w, float grad_factor,
                      int input_is_depth, void* stream);
int xpt_smooth_ms_bwd(int nscales, const float* const* disp, const float* const* image, const float* const* gloss,
                      float* const* dinput, int B, const int* h, const int* w, float grad_factor, int input_is_depth,
                      void* stream);

/* ------------------------------------------------------------------ a13: the merge of TotalLoss.__call__ (losses.py:44-55)
 * terms[t] [batch] (t < n <= 64: one per loss type and scale), c [n] (scale weight x type weight / global batch),
 * amat [types][n] (per-type read-out) -> total [1] = sum_t c[t] sum_b terms[t][b], by_type [types]; one launch, fixed
 * summation order.  bwd: g_total [1] -> grads [n][batch] = c[t] g_total. */
int xpt_merge_total_fwd(int n, const float* const* terms, const float* c, const float* amat, float* total,
                        float* by_type, int batch, int types, void* stream);
int xpt_merge_total_bwd(int n, const float* c, const float* g_total, float* grads, int batch, void* stream);

/* ------------------------------------------------------------------ a14: fused Adam (Keras semantics)
 * replaces tf.optimizers.Adam(lr).apply_gradients (model/model_util/optimizers.py:7-13,
 * model/train_val.py:86) over FLAT fp32 buffers of n elements (16-byte aligned):
 *   lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr_t m/(sqrt(v)+eps)
 *   g is first multiplied by grad_scale; `step` is a DEVICE pointer to the float step count t >= 1
 *   (so the launch can be replayed from a hipGraph); zero_grad != 0 clears g in the same pass;
 *   shadow_bf16 (may be NULL): n bfloat16 values that receive a bf16 copy of the updated parameters (the operands
 *   of the bf16 convolutions / GEMMs -- no per-layer cast launches in the forward pass). */
int xpt_adam_step(float* param, float* grad, float* m, float* v, long long n, const float* step, float lr,
                  float beta1, float beta2, float eps, float grad_scale, int zero_grad, void* shadow_bf16,
                  void* stream);
/* tf.optimizers.SGD(learning_rate) of optimizer_factory's "sgd_constant" (model/model_util/optimizers.py:10-11; momentum 0):
 * p -= lr * grad_scale * g over the same flat buffers; zero_grad / shadow_bf16 as xpt_adam_step. */
int xpt_sgd_step(float* param, float* grad, long long n, float lr, float grad_scale, int zero_grad, void* shadow_bf16,
                 void* stream);

/* ------------------------------------------------------------------ a2: depthwise convolution (NASNet separable convs)
 * replaces the depthwise half of every keras SeparableConv2D(use_bias=False) inside
 * tf.keras.applications.NASNetMobile (reference call site model/build_model/pretrained_nets.py:36-44), with the
 * preceding Activation('relu') of _separable_conv_block optionally fused (relu_in) and the ZeroPadding2D of the
 * stride-2 blocks folded into (pad_t, pad_l).
 *   x  [B,H,W,C]   NHWC activations, dtype 0 = float32, 1 = bfloat16 (fp32 accumulation)
 *   w  [C,k,k]     float32 (k in {3,5,7}), stride in {1,2}
 *   y  [B,OH,OW,C] = sum_{ky,kx} f(x[b, oy*stride+ky-pad_t, ox*stride+kx-pad_l, c]) w[c,ky,kx], f = relu | id
 * bwd_data:   dy -> dx [B,H,W,C] (x is only read for the relu mask; may be NULL when relu_in == 0)
 * bwd_weight: (x, dy) -> dw [C,k,k] float32; workspace xpt_dwconv_bwd_weight_workspace_floats() floats. */
int xpt_dwconv_fwd(const void* x, const float* w, void* y, int B, int H, int W, int C, int k, int stride, int pad_t,
                   int pad_l, int OH, int OW, int relu_in, int dtype, void* stream);
int xpt_dwconv_bwd_data(const void* x, const float* w, const void* dy, void* dx, int B, int H, int W, int C, int k,
                        int stride, int pad_t, int pad_l, int OH, int OW, int relu_in, int dtype, void* stream);
/* launch-plan knob (process-wide, for benchmarking): output groups per workgroup of the weight-gradient kernel (0, 4, 8,
 * 16, 32); -1 / -2: scalar / vectorised multi-layer kernels; -3 / -4: scalar / vectorised stride-2 data gradient;
 * -20 .. -23: lab (parts of the multi-layer backward off); -100 - cw: channels per workgroup of the small-map kernels
 * (-100 = off); -10000 - px: largest map they take; tile kernels: -20000 - (TH * 100 + TW) forward tiles (0 = off),
 * -30000 - code data-gradient tiles (1 = automatic, 0 = off), -40000 - cw channels per workgroup (0 = automatic),
 * -50000 - n forward from 2^n input elements on, -60000 - code / -70000 - code the same tiles for stride-1 layers
 * (0 = off, the default), -80000 - g narrowest channel group served (default 2) */
int xpt_dwconv_tune(int wrw_groups);
/* process-wide A/B switch (benchmarking): the kernels that number their workgroups image-major map image k of a batch of 8
 * to XCD k (each XCD's L2 keeps what it wrote across kernel boundaries); pure renumbering, same results.  Default on. */
int xpt_set_xcd_affinity(int on);
/* The 16-bit activation format of this build: 0 = bfloat16 (libxpt_hip.so), 1 = IEEE half (libxpt_hip_f16.so: the same sources
 * compiled with -DXPT_HALF_F16 for BASELINE configs[4], "fp16 convs + fp32 loss accumulation").  Wherever this header says
 * "bf16" / dtype 1 for an activation or packed-weight operand it means this format; fp32 operands, accumulators, losses and
 * optimizer state are the same in both builds.  The reference has no counterpart (it is fp32 end to end, train_val.py:78-92). */
int xpt_half_format(void);
size_t xpt_dwconv_bwd_weight_workspace_floats(int B, int OH, int OW, int C, int k);
int xpt_dwconv_bwd_weight(const void* x, const void* dy, float* dw, float* workspace, size_t workspace_floats,
                          int B, int H, int W, int C, int k, int stride, int pad_t, int pad_l, int OH, int OW,
                          int relu_in, int dtype, void* stream);

/* ------------------------------------------------------------------ a2/a3: per-channel conv epilogues on NHWC activations
 * y = act(f(x) * scale[c] + shift[c]),  scale = gamma*rsqrt(var+eps), shift = beta - mean*scale
 *   gamma == NULL : bias epilogue of CustomConv2D (model/model_util/layer_ops.py:31-35): scale = 1, shift = beta (the bias),
 *                   act = LeakyReLU(slope) (config-example.py:56-63) or linear (slope = 1);
 *   gamma != NULL : keras BatchNormalization in inference mode inside NASNetMobile (called without training=True,
 *                   model/train_val.py:82), f = relu when relu_in (the Activation('relu') that precedes it is fused).
 *   x, y [rows, C] (rows = B*H*W pixels of an NHWC tensor), dtype 0 = float32 / 1 = bfloat16; parameters float32.
 *   residual (fwd, may be NULL): y += residual [rows, C] -- the branch sum of a NASNet cell (keras layers.add) fused
 *                   into the BatchNorm that produces one of its operands; linear epilogues (slope == 1) only; its
 *                   gradient is dy itself.
 * bwd: dy [rows, C] with row pitch dy_pitch >= C elements (a channel slice of a wider tensor is read in place)
 *   -> dx (may be NULL), dbeta [C], dgamma [C] (NULL iff gamma is NULL); y is only read when slope != 1;
 *   workspace xpt_affine_act_bwd_workspace_floats(rows, C) floats; deterministic reduction. */
int xpt_affine_act_fwd(const void* x, const float* gamma, const float* beta, const float* mean, const float* var,
                       float eps, const void* residual, void* y, long long rows, int C, float slope, int relu_in,
                       int dtype, void* stream);
size_t xpt_affine_act_bwd_workspace_floats(long long rows, int C);
int xpt_affine_act_bwd(const void* x, const void* y, const void* dy, long long dy_pitch, const float* gamma,
                       const float* beta, const float* mean, const float* var, float eps, void* dx, float* dbeta,
                       float* dgamma, float* workspace, size_t workspace_floats, long long rows, int C, float slope,
                       int relu_in, int dtype, void* stream);

/* ------------------------------------------------------------------ a2: weight gradient of the pointwise (1x1) convolutions
 * replaces the filter gradient of every 1x1 Conv2D / the pointwise half of every SeparableConv2D inside
 * tf.keras.applications.NASNetMobile (reference call site model/build_model/pretrained_nets.py:36-44; gradients taken
 * by tape.gradient, model/train_val.py:85-86).
 *   dy [M, cout], x [M, cin]   bfloat16 rows of NHWC activations (M = B*H*W); pitch_* = row pitch in elements, so a
 *                              channel slice of a wider tensor is consumed in place;
 *   dw [cout, cin]             float32 = sum_m dy[m, co] x[m, ci]  (exact products, f32 accumulation on the matrix cores);
 *   workspace                  xpt_conv1x1_bwd_weight_workspace_floats() floats (split-K partial tiles);
 *   counters                   >= xpt_conv1x1_bwd_weight_counters() zero-initialised uint32, left zero on return; may be
 *                              shared by all calls issued to one stream.  Deterministic: partials are added in split order.
 * xpt_conv1x1_bwd_weight_tune: launch-plan knobs (waves per workgroup 4|16, row pairs per wave, workgroups per launch,
 * KiB of partial tiles per output tile); process-wide, for benchmarking -- the defaults are the measured optimum. */
int xpt_conv1x1_bwd_weight_tune(int waves, int pairs_per_wave, int max_blocks, int max_partial_kib);
/* deferred mode: MiB of split partials one layer may leave for xpt_reduce_partials (default 8; benchmarking) */
int xpt_conv1x1_bwd_weight_defer_cap(int mib);
size_t xpt_conv1x1_bwd_weight_workspace_floats(long long M, int cout, int cin);
int xpt_conv1x1_bwd_weight_counters(long long M, int cout, int cin);
int xpt_conv1x1_bwd_weight(const void* dy, const void* x, float* dw, float* workspace, size_t workspace_floats,
                           unsigned* counters, int n_counters, long long M, int cout, int cin, long long pitch_dy,
                           long long pitch_x, void* stream);

/* ------------------------------------------------------------------ a2: pointwise convolution + BatchNorm forward
 * replaces a 1x1 Conv2D / the pointwise half of a SeparableConv2D together with the BatchNormalization (inference
 * mode) and the `layers.add` that follow it inside tf.keras.applications.NASNetMobile (pretrained_nets.py:36-44):
 *   x [M, cin] bf16 rows (pitch_x elements apart), w [cout, cin] bf16 ->
 *   ypre [M, cout] bf16 = x w^T (fp32 accumulation on the matrix cores), y [M, cout] bf16 = BN(ypre) (+ residual [M, cout]).
 * Rows are read with the widest loads cin, pitch_x and the bases allow (16 bytes down to 2: 11-channel layers). */
int xpt_pwconv_tune(int ksplit_min_cin, int ksplit_max_tiles);   /* launch-plan knob: split the k loop over the 4 waves of a
                                                                   * workgroup from this many input channels on, up to this
                                                                   * many 32x32 output tiles (0 tiles: never) */
int xpt_pwconv_bn_fwd(const void* x, const void* w, const float* gamma, const float* beta, const float* mean,
                      const float* var, float eps, const void* residual, void* ypre, void* y, long long M, int cin,
                      int cout, long long pitch_x, void* stream);

/* ------------------------------------------------------------------ a2/a3: dense k x k convolutions (implicit GEMM, bf16 matrix cores)
 * replaces keras Conv2D(filters, k, strides, padding="same") + bias + LeakyReLU as built by CustomConv2D
 * (model/model_util/layer_ops.py:5-36) for PoseNetImproved (model/build_model/pose_net.py:57-91) and the depth decoder
 * (model/build_model/depth_net.py:101-109, 137-167), the UpSampling2D(2, "nearest") in front of the decoder's first
 * convolution of each level (depth_net.py:76-84), and tape.gradient of those layers (model/train_val.py:85-86).
 * All activations NHWC bf16 with a pixel pitch (in elements) so that channel slices / padded tensors are used in place.
 *
 * xpt_conv_pack_weights: ONE launch converts the fp32 master weights into the kernels' bf16 operand layouts,
 *   fwd [N][KH*KW][Cp] (Cp = C rounded up to 8, zero filled) and bwd [Cp][KH*KW][Np] (Np = N rounded up to 8).
 *   jobs = device array of xpt_conv_pack_job (first_block = running sum of T * ceil(max(N, Np) / 64) * ceil(Cp / 64)
 *   over the preceding jobs: one workgroup per tap and 64 x 64 channel tile; nblocks = the total).
 * xpt_conv2d_fwd: y[b,oh,ow,n] = act(bias[n] + sum_{kh,kw,c} x[b, oh*stride + kh - pad_t, ow*stride + kw - pad_l, c] w[n,kh,kw,c]),
 *   zero outside the input (TF SAME: asymmetric pads are given explicitly), act = LeakyReLU(slope) (1 = linear);
 *   x has PH x PW physical pixels of C channels (C % 8 == 0, pad channels must be finite); upsample = 1: the taps index
 *   the nearest-2x up-sampling of x (2PH x 2PW) without materialising it.
 * xpt_conv2d_bwd_data: dx[b,ih,iw,c] = sum_{kh,kw,n} g[b,(ih+pad_t-kh)/stride,(iw+pad_l-kw)/stride,n] w[n,kh,kw,c]
 *   (integral, in-range positions only; stride 1 or 2); g has Np readable channels; fold2x2 = 1: x was consumed through
 *   upsample = 1, the gradients of the four children of a pixel are summed (IH x IW = physical extent).
 * xpt_conv2d_bwd_weight_partials: split-K partial sums [splits][N][KH][KW][Cr] fp32 of
 *   dw[n,kh,kw,c] = sum_{b,oh,ow} g[b,oh,ow,n] x[b, oh*stride+kh-pad_t, ow*stride+kw-pad_l, c]  (c < Cr <= C);
 *   splits = xpt_conv2d_bwd_weight_splits(...); finished by xpt_reduce_partials.  N % 8 == 0, C % 8 == 0.
 * xpt_restack_bf16: restack_on_channels (pose_net.py:44-50): image5d [B,S,H,W,3] fp32 -> [B,H,W,Cp] bf16,
 *   channel = frame*3 + c, channels >= 3S zero.
 * xpt_conv2d_tune / xpt_conv2d_bwd_weight_tune: launch-plan knobs (process-wide, benchmarking). */
typedef struct xpt_conv_pack_job {
  const float* src;          /* element (n,kh,kw,c) at n*sn + kh*sh + kw*sw + c*sc */
  void* fwd;
  void* bwd;                 /* may be NULL */
  long long sn, sc, sh, sw;
  int N, T, KW, C, Cp, Np;
  long long first_block;
} xpt_conv_pack_job;
int xpt_conv_pack_job_bytes(void);
int xpt_conv_pack_weights(const void* jobs, int njobs, long long nblocks, void* stream);
int xpt_conv2d_tune(int plan);
int xpt_conv2d_fwd(const void* x, const void* w, const float* bias, void* y, int B, int PH, int PW, int C,
                   long long xpitch, int N, int KH, int KW, int stride, int pad_t, int pad_l, int OH, int OW,
                   long long ypitch, int upsample, float slope, void* stream);
int xpt_conv2d_bwd_data(const void* g, const void* wb, void* dx, int B, int OH, int OW, int Np, long long gpitch,
                        int C, int KH, int KW, int stride, int pad_t, int pad_l, int IH, int IW, long long dxpitch,
                        int fold2x2, void* stream);
/* Split-K variants for the deep decoder layers (csrc/xpt_conv_splitk.hip; depth_net.py:101-109 dp_up4 / dp_up3: 1,056 ... 216
 * reduction channels on the 8 x 26 / 16 x 52 maps): 128-pixel x 128 (64)-channel LDS tiles, the reduction axis cut into
 * 1 ... 16 slices whose fp32 partial tiles [slice][pixel][channel] go to `workspace`, a finishing launch adds the slices in
 * order (+ bias + LeakyReLU, + the 2 x 2 fold of a nearest-2x input) -- deterministic, no atomics.  Stride 1 only.
 * xpt_conv2d_splitk_workspace_floats: floats of workspace the launch needs, 0 = this layer is not served (use
 *   xpt_conv2d_fwd / xpt_conv2d_bwd_data); pixels = B x OH x OW of the grid the launch enumerates (data gradient with
 *   fold2x2: B x 2 IH x 2 IW), out_channels / red_channels of THAT launch (data gradient: out = C, red = Np).
 * xpt_conv2d_splitk_tune: enable, forced slice count (0 = automatic), minimum reduction length, maximum pixels served. */
int xpt_conv2d_splitk_tune(int enable, int force_split, int min_k, int max_pixels);
size_t xpt_conv2d_splitk_workspace_floats(long long pixels, int out_channels, int red_channels, int taps, int stride);
int xpt_conv2d_fwd_splitk(const void* x, const void* w, const float* bias, void* y, int B, int PH, int PW, int C,
                          long long xpitch, int N, int KH, int KW, int pad_t, int pad_l, int OH, int OW, long long ypitch,
                          int upsample, float slope, float* workspace, size_t workspace_floats, void* stream);
int xpt_conv2d_bwd_data_splitk(const void* g, const void* wb, void* dx, int B, int OH, int OW, int Np, long long gpitch,
                               int C, int KH, int KW, int pad_t, int pad_l, int IH, int IW, long long dxpitch, int fold2x2,
                               float* workspace, size_t workspace_floats, void* stream);
/* Persistent, weight-stationary variants for the 3 x 3 stride-1 layers of the decoder's half- / full-resolution levels
 * (csrc/xpt_conv_stream.hip; depth_net.py:101-109 dp_up1 / dp_up0: <= 96 output channels, weight slab + input halo <= 80 KB of
 * LDS): a workgroup stages the layer's weights once and walks over 8 x 16-pixel tiles, the next tile's halo in flight while
 * the current one is multiplied and stored; zeros of the padding come from the buffer loads' range check.
 * xpt_conv2d_stream_serves: 1 when these entry points serve the layer (else use xpt_conv2d_fwd / xpt_conv2d_bwd_data);
 *   arguments as for xpt_conv2d_splitk_workspace_floats, upsample_or_fold = the forward's upsample / the gradient's fold2x2.
 * xpt_conv2d_stream_tune: enable, minimum tile count served, workgroups per CU, LDS budget (KiB); 0 keeps a value. */
int xpt_conv2d_stream_tune(int enable, int min_tiles, int wgs_per_cu, int max_lds_kib);
int xpt_conv2d_stream_serves(int B, int OH, int OW, int out_channels, int red_channels, int KH, int KW, int stride,
                             int upsample_or_fold);
int xpt_conv2d_fwd_stream(const void* x, const void* w, const float* bias, void* y, int B, int PH, int PW, int C,
                          long long xpitch, int N, int pad_t, int pad_l, int OH, int OW, long long ypitch, int upsample,
                          float slope, void* stream);
int xpt_conv2d_fwd_stream_k5s2(const void* x, const void* w, const float* bias, void* y, int B, int PH, int PW, int C,
                               long long xpitch, int N, int pad_t, int pad_l, int OH, int OW, long long ypitch, float slope,
                               void* stream);   /* 5 x 5 stride-2 forward (pose_net.py:60-61), served when xpt_conv2d_stream_serves says so */
int xpt_conv2d_bwd_data_stream(const void* g, const void* wb, void* dx, int B, int OH, int OW, int Np, long long gpitch, int C,
                               int pad_t, int pad_l, int IH, int IW, long long dxpitch, int fold2x2, void* stream);
int xpt_conv2d_bwd_weight_tune(int max_partial_mib, int target_blocks);
int xpt_conv2d_bwd_weight_splits(int B, int C, int N, int KH, int KW, int stride, int OH, int OW);
int xpt_conv2d_bwd_weight_partials(const void* g, const void* x, float* partials, size_t partial_floats, int B, int PH,
                                   int PW, int C, int Cr, long long xpitch, int N, long long gpitch, int KH, int KW,
                                   int stride, int pad_t, int pad_l, int OH, int OW, int upsample, void* stream);
int xpt_restack_bf16(const float* image5d, void* out, int B, int S, int H, int W, int Cp, void* stream);

/* ------------------------------------------------------------------ a2: decoder prediction heads
 * replaces the Conv2D(1, 3, padding="same", activation linear) of get_scaled_depth (model/build_model/depth_net.py:87-92)
 * and its tape.gradient: pre [B,H,W] fp32 = bias + conv3x3_same(x [B,H,W,C] bf16 with pixel pitch xpitch, w [3][3][C] fp32),
 * C in {16, 32, 64, 128}.  bwd: g [B,H,W] fp32 -> dx [B,H,W,C] bf16 dense and per-workgroup partials
 * [xpt_headconv_bwd_blocks()][9 C + 1] fp32 (dW [3][3][C], then dbias), finished by xpt_reduce_partials. */
int xpt_headconv_tune(int passes, int max_blocks);   /* launch-plan knob of xpt_headconv_bwd (pixel passes per workgroup, most workgroups) */
int xpt_headconv_bwd_blocks(int B, int H, int W, int C);
int xpt_headconv_fwd(const void* x, long long xpitch, const float* w, const float* bias, float* pre, int B, int H, int W,
                     int C, void* stream);
int xpt_headconv_bwd(const void* x, long long xpitch, const float* w, const float* g, void* dx, float* partials,
                     size_t partial_floats, int B, int H, int W, int C, void* stream);
/* the same with the gradient the feature map received from its OTHER consumer (the next decoder level's first convolution,
 * depth_net.py:137-167) added in: dx = addend + head data gradient; addend bf16 [B,H,W,C] with pixel pitch addend_pitch, or NULL. */
int xpt_headconv_bwd_add(const void* x, long long xpitch, const float* w, const float* g, const void* addend,
                         long long addend_pitch, void* dx, float* partials, size_t partial_floats, int B, int H, int W, int C,
                         void* stream);

/* ------------------------------------------------------------------ a2/a4: depth head activation
 * InverseSigmoid (model/build_model/model_factory.py:134-138) and safe_reciprocal_number (utils/util_funcs.py:157-160):
 *   depth = safe_rcp(sigmoid(x) + 0.01), disp = safe_rcp(depth), safe_rcp(v) = (1 / v) [v > 1e-5]; float32, n elements.
 * bwd: gx = d depth/dx (g_depth + d disp/d depth g_disp); g_depth or g_disp may be NULL (not both). */
int xpt_depth_head_fwd(const float* x, float* depth, float* disp, long long n, void* stream);
int xpt_depth_head_bwd(const float* x, const float* g_depth, const float* g_disp, float* gx, long long n, void* stream);
/* The decoder's four prediction scales (depth_net.py:137-167) in one launch each way: nscales (1..4) arrays of n[s]
 * elements.  bwd: g_depth[s] / g_disp[s] may be NULL; a scale with both NULL gets gx[s] = 0. */
int xpt_depth_head_ms_fwd(int nscales, const float* const* x, float* const* depth, float* const* disp, const long long* n,
                          void* stream);
int xpt_depth_head_ms_bwd(int nscales, const float* const* x, const float* const* g_depth, const float* const* g_disp,
                          float* const* gx, const long long* n, void* stream);

/* tf.keras.layers.GlobalAveragePooling2D closing PoseNet (model/build_model/pose_net.py:45) on an NHWC map x [B,HW,C]
 * (dtype 0 float32 / 1 bfloat16) -> y float32 [B,C]; bwd: g float32 [B,C] -> dx [B,HW,C] of that dtype = g / HW. */
int xpt_global_avgpool_fwd(const void* x, float* y, int B, int HW, int C, int dtype, void* stream);
int xpt_global_avgpool_bwd(const float* g, void* dx, int B, int HW, int C, int dtype, void* stream);

/* resize_image (model/model_util/layer_ops.py:43-50; tf.image.resize bilinear, half-pixel centres) at an exact factor 2 on
 * the decoder's one-channel raw predictions (depth_net.py:87-92): src float32 [M,h,w] -> out [M,2h,2w] (dtype 0 float32 /
 * 1 bfloat16: it is concatenated with bf16 features); bwd: g [M,2h,2w] read with a pixel pitch of g_pitch elements (a
 * channel slice of the NHWC concatenation's gradient, in place) -> dsrc float32 [M,h,w] (gather form: no zero fill). */
int xpt_upsample2x_fwd(const float* src, void* out, long long M, int h, int w, int dtype, void* stream);
int xpt_upsample2x_bwd(const void* g, long long g_pitch, float* dsrc, long long M, int h, int w, int dtype, void* stream);
/* the same with the gradient src received from its OTHER consumer (the depth activation of the same raw prediction,
 * depth_net.py:87-92) added in: dsrc = addend + adjoint(g); addend float32 [M,h,w] or NULL. */
int xpt_upsample2x_bwd_add(const void* g, long long g_pitch, const float* addend, float* dsrc, long long M, int h, int w, int dtype,
                           void* stream);

/* ------------------------------------------------------------------ f-4: PWC-Net correlation cost volume
 * tfa.layers.CorrelationCost(kernel_size=1, max_displacement=md, stride_1=1, stride_2=s2, pad=md, channels_last)
 * as called by PWCNet.correlation (model/build_model/flow_net.py:181-196; md = 128 >> level, s2 = max(md / 4, 1)):
 *   out[b,y,x, ty*D + tx] = (1/C) sum_c left[b,y,x,c] * right[b, y + (ty-rad)*s2, x + (tx-rad)*s2, c],  rad = md / s2,
 *   D = 2 rad + 1, zero where the displaced pixel is outside the image.  left / right [B,H,W,C] dense NHWC,
 *   out [B,H,W,D*D]; dtype 0 float32 / 1 bfloat16 (all three tensors), fp32 accumulation.
 * bwd: dleft, dright [B,H,W,C] from gout [B,H,W,D*D] (gathers, deterministic).  D*D <= 256; bwd: C <= 1024 (C % 4 == 0) or C <= 256. */
int xpt_corr_cost_channels(int max_disp, int stride2);
int xpt_corr_cost_fwd(const void* left, const void* right, void* out, int B, int H, int W, int C, int max_disp,
                      int stride2, int dtype, void* stream);
int xpt_corr_cost_bwd(const void* left, const void* right, const void* gout, void* dleft, void* dright, int B, int H,
                      int W, int C, int max_disp, int stride2, int dtype, void* stream);

/* ------------------------------------------------------------------ a2: the cells' 3x3 average pooling
 * keras AveragePooling2D((3,3), strides (1,1), padding='same') inside NASNetMobile (divisor = number of in-image taps),
 * times `scale` (add([avg(p), avg(p)]) of the normal cell = scale 2).  in [B,H,W,C] NHWC with row pitch in_pitch >= C
 * elements, out dense; dtype 0 float32 / 1 bfloat16; adjoint = 1 applies the transposed operator (the backward). */
int xpt_avgpool3_same(const void* in, long long in_pitch, void* out, int B, int H, int W, int C, float scale, int adjoint,
                      int dtype, void* stream);

/* the same for n (<= 6) independent layers of one shape in one launch (arrays of n pointers; residual[j] may be NULL) */
int xpt_pwconv_bn_multi_fwd(int n, const void* const* x, const void* const* w, const float* const* gamma,
                            const float* const* beta, const float* const* mean, const float* const* var, float eps,
                            const void* const* residual, void* const* ypre, void* const* y, long long M, int cin, int cout,
                            long long pitch_x, void* stream);
/* ... with SIBLING layers (keras nasnet._normal_a_cell: x1 = add([left1, right1]), both ending in a pointwise convolution
 * + BatchNormalization of one shape): where sib_x[j] is not NULL job j computes
 *   y_j = bn_j(x_j w_j^T) + bf16(bn'_j(sib_x_j sib_w_j^T)) (+ residual_j)
 * in the same workgroups -- the sum never goes through memory -- and sib_ypre[j] receives the sibling's convolution output
 * (its BatchNorm backward needs it).  Bit for bit what two launches (siblings first, then the main layers with the sibling
 * outputs as residuals) produce.  The sib_* arrays hold n entries (entries of jobs without a sibling are NULL) and may all
 * be NULL.  pitch_y: row pitch of every y[j] in elements (0 = cout): the jobs of the "spatial" _adjust_block write the two
 * halves of ONE [M, 2 cout] tensor (y[1] = y[0] + cout, pitch_y = 2 cout; no residuals then); ypre stays dense. */
int xpt_pwconv_bn_multi_fwd_sib(int n, const void* const* x, const void* const* w, const float* const* gamma,
                                const float* const* beta, const float* const* mean, const float* const* var, float eps,
                                const void* const* residual, void* const* ypre, void* const* y, const void* const* sib_x,
                                const void* const* sib_w, const float* const* sib_gamma, const float* const* sib_beta,
                                const float* const* sib_mean, const float* const* sib_var, void* const* sib_ypre,
                                long long M, int cin, int cout, long long pitch_x, long long pitch_y, void* stream);

/* ------------------------------------------------------------------ a14: a new batch into the captured step's inputs
 * The step is a replayed hipGraph reading static buffers (xpt_mde_2021_amd/model/train_val.py, replacing the per-step feed of
 * model/train_val.py:78-92 `features`): dst[i][0 .. bytes[i]) = src[i][0 .. bytes[i]) for n (1..8) device buffers in ONE
 * launch (no src may overlap a dst). */
int xpt_multi_copy(const void* const* src, void* const* dst, const long long* bytes, int n, void* stream);

/* ------------------------------------------------------------------ a2: the decoder's channel concatenation
 * replaces tf.concat([upconv, skip, up-sampled prediction], axis=-1) of upconv_with_skip_connection
 * (model/build_model/depth_net.py:104-107).  out [rows, Ct] bf16 (dense, Ct % 8 == 0, 16-byte aligned) = the n (1..4) bf16
 * inputs [rows, channels[i]] (row pitches in elements) side by side and zeros in the remaining channels (the pad to the
 * 8-channel groups xpt_conv2d_fwd reads). */
int xpt_concat_channels(const void* const* inputs, const long long* pitches, const int* channels, int n, void* out,
                        long long rows, int Ct, void* stream);

/* ------------------------------------------------------------------ a2: gradient fan-in of a multiply used activation
 * out [rows, C] = sum_i inputs[i] [rows, C] (row pitch pitches[i] >= C elements; n = 2..8; dtype 0 float32 / 1 bfloat16,
 * fp32 accumulation in input order).  Replaces the chain of pairwise adds autograd (tape.gradient, train_val.py:85)
 * performs for a NASNet cell input that feeds up to six branches. */
int xpt_sum_rows(const void* const* inputs, const long long* pitches, int n, void* out, long long rows, int C, int dtype,
                 void* stream);

/* ------------------------------------------------------------------ a2: the elementwise tail of a NASNet-A cell
 * Replaces the keras layers that end _normal_a_cell / _reduction_a_cell of tensorflow.keras.applications.nasnet (built by
 * the reference at model/build_model/pretrained_nets.py:11-44) together with the Activation('relu') every consumer of a
 * cell output starts with:  out[:, s F:(s+1) F] = relu(sum_k scale * (pooled ? AveragePooling2D(3,1,'same')(in) : in)).
 * src / pitch / pooled / scale hold two slots per slice (index s*2 + k, k < nterms[s] <= 2); all tensors are NHWC rows
 * [B H W][channels]; dtype 0 float32 / 1 bfloat16. */
int xpt_cell_tail_fwd(int nslices, const int* nterms, const void* const* src, const long long* pitch, const int* pooled,
                      const float* scale, void* out, long long out_pitch, int B, int H, int W, int F, int dtype,
                      void* stream);

/* its backward in one launch: gm [M, nslices F] = [out > 0] * (sum of the ngrads <= 4 consumer gradients); dense input
 * gradient j < ndense <= 4: dense_out[j] [M, F] = sum of bterms[j] <= 3 terms (slice / pooled / scale at index j*3 + k) of
 * the masked gradient (pooled: through the adjoint of the average pooling). */
int xpt_cell_tail_bwd(int ngrads, const void* const* grads, const long long* gpitch, const void* out, long long out_pitch,
                      void* gm, int nslices, int ndense, void* const* dense_out, const int* bterms, const int* slice,
                      const int* pooled, const float* scale, int B, int H, int W, int F, int dtype, void* stream);

/* ------------------------------------------------------------------ a2: "spatial" adjust block and reduction-cell pools
 * keras nasnet._adjust_block when p has twice the cell's resolution: p1 = AveragePooling2D((1,1), strides 2)(p) = p[2i,2j],
 * p2 = the same after ZeroPadding2D(((0,1),(0,1))) + Cropping2D(((1,0),(1,0))) = p[2i+1,2j+1] (0 outside).  One gather
 * builds out [B, ceil(H/2), ceil(W/2), 2C] = [p1 | p2]; one scatter is its backward (d1 / d2 may be NULL). */
int xpt_adjust_gather(const void* in, long long in_pitch, void* out, int B, int H, int W, int C, int dtype, void* stream);
int xpt_adjust_scatter(const void* d1, long long pitch1, const void* d2, long long pitch2, void* out, int B, int H, int W,
                       int C, int dtype, void* stream);

/* keras nasnet._reduction_a_cell: ZeroPadding2D(correct_pad(h, 3)) -> MaxPooling2D(3, strides 2, 'valid') and
 * AveragePooling2D(3, strides 2, 'valid') of the same h, in one launch (arg: uint8 arg-max tap per output element) and
 * one backward launch (gmp / gap may be NULL).  pad_t / pad_l: the leading zero padding (0 or 1). */
int xpt_pool_pair_fwd(const void* in, long long in_pitch, void* mp, void* ap, void* arg, int B, int H, int W, int C, int OH,
                      int OW, int pad_t, int pad_l, int dtype, void* stream);
int xpt_pool_pair_bwd(const void* gmp, long long pitch_m, const void* gap, long long pitch_a, const void* arg, void* dh, int B,
                      int H, int W, int C, int OH, int OW, int pad_t, int pad_l, int dtype, void* stream);
/* ... with a second gradient of the max-pooled tensor (two consumers of it: added on load, no fan-in launch); gmp2 may be NULL */
int xpt_pool_pair_bwd2(const void* gmp, long long pitch_m, const void* gmp2, long long pitch_m2, const void* gap, long long pitch_a,
                       const void* arg, void* dh, int B, int H, int W, int C, int OH, int OW, int pad_t, int pad_l, int dtype,
                       void* stream);

/* ------------------------------------------------------------------ f-2: the in-step augmentation
 * TotalAugment over [CropAndResize(p_crop), HorizontalFlip(p_flip), ColorJitter(p_jit)] (model/model_util/augmentation.py:
 * 22-219; called inside the training step, model/train_val.py:79-81) in ONE launch: crop box from four uniforms (:94-109),
 * tf.image.crop_and_resize sampling (bilinear, corner aligned, zeros outside; nearest for the ground-truth depth), image
 * flip, saturation + gamma jitter, and the matching rewrites of intrinsics (:111-129, :169-173), ground-truth poses and the
 * stereo extrinsic (:175-186).  u [8] device uniforms in [0,1): crop y1, x1, y2, x2; flip; jitter; gamma; saturation;
 * params [8] out: the box, flip (0/1), jitter (0/1), gamma, saturation.  img / img_out [n_img, H, W, 3] float32 (second
 * pair: the right camera, or NULL); depth [n_depth, H, W] or NULL; K [B, 3, 3]; pose [n_pose, 4, 4] or NULL; stereo
 * [B, 4, 4] or NULL.  Outputs must not alias inputs. */
int xpt_augment(const float* u, float* params, const float* img0, float* img0_out, const float* img1, float* img1_out,
                int n_img, const float* depth, float* depth_out, int n_depth, const float* K0, float* K0_out, const float* K1,
                float* K1_out, int B, const float* pose0, float* pose0_out, const float* pose1, float* pose1_out, int n_pose,
                const float* stereo, float* stereo_out, int H, int W, float p_crop, float p_flip, float p_jit,
                float half_crop, void* stream);
/* Pinned draws for the replay check of a captured step (xpt_mde_2021_amd/model/train_val.py): every later xpt_augment launch
 * carries `pin` (device float[9], NULL to stop) and takes pin[1..8] as its uniforms while pin[0] > 0.5 -- decided on the
 * device at run time, so replays of one captured graph can be made to repeat their draws and to draw freshly again. */
int xpt_augment_pin(const float* pin);

/* Encoder input as PretrainedModel prepares it (model/build_model/pretrained_nets.py:36-43): image / 127.5 - 1, bilinear resize
 * (TF2 half-pixel centres) to (H+2, W+2), written as the NHWC bf16 tensor [B, H+2, W+2, 8] (channels 3..7 zero) the stem
 * convolution reads.  image: B frames [H, W, 3] float32, frame b at image + b * batch_stride elements. */
int xpt_stem_input(const float* image, long long batch_stride, void* out, int B, int H, int W, void* stream);

/* ------------------------------------------------------------------ f-3: the per-step depth metric of the training loop
 * get_depth_metric (model/train_val.py:180-200) = valid_depth_filter + median scaling + abs-rel
 * (evaluate/eval_utils.py:109-131) for every sample of the batch in one launch (radix selection instead of two sorts):
 * per_sample[b] = mean over mask of |gt - clip(pred * median(gt)/median(pred), min, max)| / gt,
 * mask = (gt > min_depth) & (gt < max_depth) & rows [r0, r1) & columns [c0, c1) (the Garg crop); 0 for an empty mask.
 * pred, gt [B, h, w] float32 contiguous. */
int xpt_depth_metric(const float* pred, const float* gt, float* per_sample, int B, int h, int w, int r0, int r1, int c0,
                     int c1, float min_depth, float max_depth, void* stream);

/* get_pose_metric (model/train_val.py:203-210) = PoseMetricNumpy (evaluate/eval_utils.py:15-87) for the batch in one launch:
 * out[0..2] = mean absolute-scale trajectory error, mean scale-aligned trajectory error, mean rotation error (radians) of
 * pred [B, N, 4, 4] (pose matrices, e.g. from xpt_pose_rvec2matr_fwd) against truth [B, N, 4, 4]. */
int xpt_pose_metric(const float* pred, const float* truth, float* out, int B, int N, void* stream);

/* ------------------------------------------------------------------ deferred parameter gradients (one finishing launch per step)
 * The *_partials entry points compute the same parameter gradients as xpt_affine_act_bwd / xpt_dwconv_bwd_weight /
 * xpt_conv1x1_bwd_weight (tape.gradient of the layer variables, model/train_val.py:85-86) but stop at the
 * per-workgroup partial sums, left in a caller-owned persistent workspace:
 *   affine:   partials[blocks][2][C]     row 0 -> dbeta (bias gradient), row 1 -> dgamma; blocks = xpt_affine_act_bwd_blocks()
 *   dwconv:   partials[chunks][C][k][k]  chunks = xpt_dwconv_bwd_weight_chunks()
 *   conv1x1:  partials[splits][cout][cin] splits = xpt_conv1x1_bwd_weight_splits()
 * xpt_reduce_partials then finishes all layers at once: block b of the launch serves job blockmap[b].x and the outputs
 * starting at blockmap[b].y (256 outputs per block when split_waves is 1 or 16, 64 when it is 4, 2048 when it is 32);
 *   dst[i] = sum over segments g < nseg, splits s < nsplit[g] of src[g][s * stride[g] + i],  i < n   (fixed order).
 * jobs and blockmap are device arrays built once by the host (the layer list of a model is static). */
typedef struct xpt_reduce_job {
  float* dst;
  long long n;
  int nseg;          /* 1..4 segments (a layer applied several times per step contributes one segment per use) */
  int split_waves;   /* 1: one output per thread (nsplit <= 8); 4: the 4 waves of a block share 64 outputs;
                      * 16 ("wide") and 32 ("flat", for few splits): float4 loads, needing n % 4 == 0, n >= 256 and
                      * dst, src[], stride[] 16-byte aligned */
  const float* src[4];
  long long stride[4];
  int nsplit[4];
} xpt_reduce_job;
int xpt_reduce_job_bytes(void);
int xpt_reduce_partials(const void* jobs, const void* blockmap, int nblocks, void* stream);
int xpt_affine_act_bwd_blocks(long long rows, int C);
int xpt_affine_act_bwd_partials(const void* x, const void* y, const void* dy, long long dy_pitch, const float* gamma,
                                const float* beta, const float* mean, const float* var, float eps, void* dx,
                                float* partials, size_t partial_floats, long long rows, int C, float slope, int relu_in,
                                int dtype, void* stream);
int xpt_dwconv_bwd_weight_chunks(int B, int OH, int OW, int C, int k, int stride);
int xpt_dwconv_bwd_weight_partials(const void* x, const void* dy, float* partials, size_t partial_floats, int B, int H,
                                   int W, int C, int k, int stride, int pad_t, int pad_l, int OH, int OW, int relu_in,
                                   int dtype, void* stream);
/* data gradient + weight-gradient partials of one depthwise layer in ONE launch (same results as xpt_dwconv_bwd_data and
 * xpt_dwconv_bwd_weight_partials; the two computations only share their inputs) */
int xpt_dwconv_bwd_both(const void* x, const float* w, const void* dy, void* dx, float* partials, size_t partial_floats,
                        int B, int H, int W, int C, int k, int stride, int pad_t, int pad_l, int OH, int OW, int relu_in,
                        int dtype, void* stream);
/* xpt_conv1x1_bn_bwd_partials for n (<= 6) layers of one shape (M, cout, cin, pitch_x) in one launch: arrays of n
 * pointers, pitch_dy per layer; every partial buffer sized as for the single call. */
int xpt_conv1x1_bn_multi_bwd_partials(int n, const void* const* dy, const long long* pitch_dy, const void* const* ypre,
                                      const void* const* x, const float* const* gamma, const float* const* var,
                                      const float* const* mean, float eps, void* const* g_out, float* const* w_partials,
                                      float* const* bn_partials, size_t w_partial_floats, size_t bn_partial_floats,
                                      long long M, int cout, int cin, long long pitch_x, void* stream);
/* The whole backward of conv1x1 -> BatchNorm in ONE launch: the partials of xpt_conv1x1_bn_bwd_partials_sum plus, when
 * dx is not NULL, the data gradient dx [M, cin] bf16 = ((dy + dy2 + dy3) * s) W (what torch.mm(g, W) did in a second
 * launch; tape.gradient w.r.t. the layer input, model/train_val.py:85-86), computed by extra workgroups of the same
 * launch straight from dy (g is not written).  w: the layer's bf16 weight [cout, cin], dense. */
int xpt_conv1x1_bwd_fused(const void* dy, const void* x, const void* w, void* dx, float* partials,
                          size_t partial_floats, long long M, int cout, int cin, long long pitch_dy, long long pitch_x,
                          void* stream);   /* no BatchNorm behind the convolution: dx = dy W */
int xpt_conv1x1_bn_bwd_fused(const void* dy, const void* dy2, const void* dy3, const void* ypre, const void* x,
                             const void* w, const float* gamma, const float* var, const float* mean, float eps, void* dx,
                             float* w_partials, size_t w_partial_floats, float* bn_partials, size_t bn_partial_floats,
                             long long M, int cout, int cin, long long pitch_dy, long long pitch_dy2, long long pitch_dy3,
                             long long pitch_x, void* stream);
/* The same for n (<= 6) layers of one shape; dx[j] NULL skips that layer's data gradient. */
int xpt_conv1x1_bn_multi_bwd_fused(int n, const void* const* dy, const long long* pitch_dy, const void* const* ypre,
                                   const void* const* x, const void* const* w, const float* const* gamma,
                                   const float* const* var, const float* const* mean, float eps, void* const* dx,
                                   float* const* w_partials, float* const* bn_partials, size_t w_partial_floats,
                                   size_t bn_partial_floats, long long M, int cout, int cin, long long pitch_x,
                                   void* stream);
/* ... with gradient FAN-IN per layer: the output gradient of layer j is dy[j] + dy2[j] + dy3[j] (its output feeds up to three
 * consumers of the cell: tape.gradient sums them, model/train_val.py:85; dy2[j] / dy3[j] and the arrays themselves may be
 * NULL), added on load in fp32 and rounded to bf16 once -- what xpt_sum_rows in front of the launch would produce. */
int xpt_conv1x1_bn_multi_bwd_fused_fan(int n, const void* const* dy, const long long* pitch_dy, const void* const* dy2,
                                       const long long* pitch_dy2, const void* const* dy3, const long long* pitch_dy3,
                                       const void* const* ypre, const void* const* x, const void* const* w,
                                       const float* const* gamma, const float* const* var, const float* const* mean,
                                       float eps, void* const* dx, float* const* w_partials, float* const* bn_partials,
                                       size_t w_partial_floats, size_t bn_partial_floats, long long M, int cout, int cin,
                                       long long pitch_x, void* stream);
/* Several depthwise layers of one activation shape and stride in one launch (the mutually independent branch
 * convolutions of a NASNet cell stage): forward y[j] = dwconv(f(x[j]), w[j]) with kernel size k[j] in {3,5,7} and leading
 * padding (pad_t[j], pad_l[j]); backward: dxin[u] = gradient of the u-th DISTINCT input summed over the jobs reading it
 * (input_of[j]), partials[j] = the weight-gradient partials of job j (layout / count as
 * xpt_dwconv_bwd_weight_partials). n <= 6. */
int xpt_dwconv_multi_fwd(const void* const* x, const float* const* w, void* const* y, const int* k, const int* pad_t,
                         const int* pad_l, int n, int B, int H, int W, int C, int stride, int OH, int OW, int relu_in,
                         int dtype, void* stream);
int xpt_dwconv_multi_bwd(const void* const* xin, void* const* dxin, int n_inputs, const void* const* dy,
                         const float* const* w, float* const* partials, const int* k, const int* pad_t,
                         const int* pad_l, const int* input_of, int n, int B, int H, int W, int C, int stride, int OH,
                         int OW, int relu_in, int dtype, void* stream);
int xpt_conv1x1_bwd_weight_splits(long long M, int cout, int cin);
/* conv1x1 -> BatchNormalization backward in ONE launch (the BN layer that follows every pointwise convolution of a
 * NASNet cell, keras nasnet._separable_conv_block / _adjust_block / cell heads): dy = gradient of the BN output,
 * ypre = the convolution output the BN saw.  Writes g_out [M, cout] bf16 = dy * gamma * rsqrt(var + eps) (operand of the
 * data-gradient GEMM), the filter-gradient partials w_partials [splits][cout][cin] of g^T x, and bn_partials
 * [splits][2][cout] (row 0 -> dbeta, row 1 -> dgamma); splits = xpt_conv1x1_bwd_weight_splits(). */
int xpt_conv1x1_bn_bwd_partials(const void* dy, const void* ypre, const void* x, const float* gamma, const float* var,
                                const float* mean, float eps, void* g_out, float* w_partials, size_t w_partial_floats,
                                float* bn_partials, size_t bn_partial_floats, long long M, int cout, int cin,
                                long long pitch_dy, long long pitch_x, void* stream);
/* the same with the output gradient arriving in up to three pieces (the layer's output feeds several branches of a cell:
 * dy + dy2 + dy3 summed on load in fp32 and rounded to bf16 once, i.e. the result of a separate xpt_sum_rows launch);
 * dy2 / dy3 may be NULL */
int xpt_conv1x1_bn_bwd_partials_sum(const void* dy, const void* dy2, const void* dy3, const void* ypre, const void* x,
                                    const float* gamma, const float* var, const float* mean, float eps, void* g_out,
                                    float* w_partials, size_t w_partial_floats, float* bn_partials,
                                    size_t bn_partial_floats, long long M, int cout, int cin, long long pitch_dy,
                                    long long pitch_dy2, long long pitch_dy3, long long pitch_x, void* stream);
int xpt_conv1x1_bwd_weight_partials(const void* dy, const void* x, float* partials, size_t partial_floats, long long M,
                                    int cout, int cin, long long pitch_dy, long long pitch_x, void* stream);

/* ------------------------------------------------------------------ input contract helper (host function, no GPU)
 * CRC-32C of a host buffer: the checksum of the TFRecord framing read by tfrecords/tfrecord_reader.py:61-75
 * (tf.data.TFRecordDataset); masked value = ((crc >> 15) | (crc << 17)) + 0xa282ead8. */
uint32_t xpt_crc32c(const void* data, size_t nbytes);
/* The reader's per-record work, natively (round 4; tfrecords/tfrecord_reader.py:61-108, tfr_util.py:8-77 of the reference):
 * xpt_tfrecord_index frames a shard held in memory (payload offset / length / stored masked CRC per record; header CRCs
 *   checked when verify_crc); returns the record count, max_records when the arrays were too small, -(k + 1) when record k is
 *   truncated or corrupt.
 * xpt_tfrecord_decode checks one payload's masked CRC-32C, walks the serialized tf.train.Example and copies, for every listed
 *   key, the first bytes_list value (exactly dst_bytes[i] bytes) / int64_list value (8) / float_list value (4) into dst[i].
 *   0 ok, -10 CRC mismatch, -11 malformed, -(100 + i) key i missing, -(1000 + i) key i has another size.  Thread-safe. */
long long xpt_tfrecord_index(const void* shard, size_t nbytes, int verify_crc, unsigned long long* payload_off,
                             unsigned long long* payload_len, unsigned int* payload_crc, long long max_records);
int xpt_tfrecord_decode(const void* payload, size_t nbytes, unsigned int stored_crc, int verify_crc, int nkeys,
                        const char* const* keys, void* const* dst, const size_t* dst_bytes);

/* ------------------------------------------------------------------ NASNet branch stage in one launch (xpt_sepconv.hip)
 * keras nasnet._separable_conv_block halves (Activation('relu') -> SeparableConv2D(k, stride 1, 'same') ->
 * BatchNormalization, inference statistics) of up to 6 branches of one cell stage, optionally with the sibling branch of
 * the cell's `add` (_normal_a_cell) or a residual: tensorflow.keras.applications NASNetMobile as instantiated at
 * model/build_model/pretrained_nets.py:36-44.  Arrays of n entries; *_b = the sibling branch of job j (x_b[j] NULL: none).
 * Same results, bit for bit, as xpt_dwconv_multi_fwd followed by xpt_pwconv_bn_multi_fwd (right branches first). */
int xpt_sepconv_bn_multi_fwd(int n, const void* const* x, const float* const* wdw, const void* const* wpw,
                             const float* const* gamma, const float* const* beta, const float* const* mean,
                             const float* const* var, const int* k, void* const* ydw, void* const* ypre,
                             const void* const* x_b, const float* const* wdw_b, const void* const* wpw_b,
                             const float* const* gamma_b, const float* const* beta_b, const float* const* mean_b,
                             const float* const* var_b, const int* k_b, void* const* ydw_b, void* const* ypre_b,
                             void* const* yb, const void* const* residual, void* const* y, float eps, int B, int H, int W,
                             int C, int cout, void* stream);

/* ------------------------------------------------------------------ captured-step audit (no reference counterpart)
 * Node census of a captured hipGraph (hipGraph_t as torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph() hands it out),
 * child graphs included: counts[6] = kernel, memcpy, memset, host, other nodes, total.  The trainers that replace the
 * reference's @tf.function step (model/train_val.py:95-102) refuse a captured step with a memset node: such nodes replay
 * wrongly on this runtime. */
int xpt_graph_node_census(void* hip_graph, int* counts);

#ifdef __cplusplus
}
#endif
#endif /* XPT_HIP_H_ */
