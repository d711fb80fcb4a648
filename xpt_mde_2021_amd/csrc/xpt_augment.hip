// xpt_augment.hip -- the training step's augmentation in ONE launch (gfx950).
//
// Replaces TotalAugment over [CropAndResize, HorizontalFlip, ColorJitter] (model/model_util/augmentation.py:22-219, called
// inside the step at model/train_val.py:79-81) for the images, the ground-truth depth and the small per-sample tensors
// (intrinsics, ground-truth poses, stereo extrinsic) of a feature dict.  As a chain of tensor ops it is ~60 launches and
// 0.49 ms of the 8.5 ms step (every elementwise pass over the [B,5,H,W,3] fp32 snippets moves 51 MB); here every output
// pixel is produced once:
//     crop box  (y1, x1, y2, x2) from four uniforms (augmentation.py:94-109: each side cropped with probability p by <= 10 %)
//     sample    tf.image.crop_and_resize semantics: corner-aligned bilinear (nearest for depth), zeros outside the image
//     flip      column W-1-j when u_flip < p_flip (images only: the reference leaves depth_gt alone, :147-166)
//     jitter    when u_jit < p_jit: x -> ((x+1)/2) saturation-adjusted (HSV S *= U(0.5,1.5), clipped), ^ gamma U(0.5,1.5), *2-1
// and one extra workgroup rewrites the intrinsics (cx' = (cx - x1 W)/(x2 - x1), fx' = fx/(x2 - x1), ...; flip: |W e_02 - K|)
// and negates row / column 0 of the ground-truth poses and of the stereo extrinsic for a flip (:111-129, :169-186).
// The eight uniforms come from ONE torch.rand call on the device generator (fresh values on every hipGraph replay).
#include "xpt_common.h"

namespace {

struct AugArgs {
  const float* u;              // [8] uniforms: crop y1, x1, y2, x2; flip; jitter; gamma; saturation
  const float* pin;            // null, or [9]: pin[0] > 0.5 -> the draws are pin[1..8] instead of u (xpt_augment_pin)
  float* params;               // [8] out: box (y1, x1, y2, x2), flip (0/1), jitter (0/1), gamma, saturation
  const float* img[2];         // [n_img, H, W, 3] (image5d, image5d_R)
  float* img_out[2];
  const float* depth;          // [n_depth, H, W] or null
  float* depth_out;
  const float* K[2];           // [B, 3, 3]
  float* K_out[2];
  const float* pose[2];        // [n_pose, 4, 4]
  float* pose_out[2];
  const float* stereo;         // [B, 4, 4]
  float* stereo_out;
  int n_img, n_depth, B, n_pose, H, W;
  float p_crop, p_flip, p_jit, half_crop;
  unsigned img_blocks, depth_blocks;     // workgroups per image tensor / for the depth
};

struct Draw {
  float y1, x1, y2, x2, gamma, sat;
  bool flip, jit;
};

__device__ inline float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

__device__ inline Draw decode(const AugArgs& a_) {
  AugArgs a = a_;
  if (a.pin != nullptr && a.pin[0] > 0.5f) a.u = a.pin + 1;       // pinned draws (the replay check of a captured step)
  Draw d;
  const float max1 = a.half_crop, min1 = -(1.f - a.p_crop) * a.half_crop / a.p_crop;
  const float min2 = 1.f - max1, max2 = 1.f - min1;
  d.y1 = clamp01(a.u[0] * (max1 - min1) + min1);
  d.x1 = clamp01(a.u[1] * (max1 - min1) + min1);
  d.y2 = clamp01(a.u[2] * (max2 - min2) + min2);
  d.x2 = clamp01(a.u[3] * (max2 - min2) + min2);
  d.flip = a.u[4] < a.p_flip;
  d.jit = a.u[5] < a.p_jit;
  d.gamma = a.u[6] + 0.5f;
  d.sat = a.u[7] + 0.5f;
  return d;
}

// tf.image.adjust_saturation on one pixel: with hue and value fixed, c' = v - (v - c) * s'/s
__device__ inline void jitter(float (&c)[3], float gamma, float sat) {
#pragma unroll
  for (int k = 0; k < 3; ++k) c[k] = (c[k] + 1.f) * 0.5f;
  const float v = fmaxf(fmaxf(c[0], c[1]), c[2]), mn = fminf(fminf(c[0], c[1]), c[2]);
  const float delta = v - mn;
  const float s = v > 0.f ? delta / fmaxf(v, 1e-12f) : 0.f;
  const float s_new = clamp01(s * sat);
  const float ratio = s > 0.f ? s_new / fmaxf(s, 1e-12f) : 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float t = v - (v - c[k]) * ratio;
    c[k] = powf(fmaxf(t, 0.f), gamma) * 2.f - 1.f;
  }
}

__global__ __launch_bounds__(256) void augment_kernel(AugArgs a) {
  const Draw d = decode(a);
  const int H = a.H, W = a.W;
  unsigned blk = blockIdx.x;
  const unsigned img_total = a.img_blocks * (a.img[1] ? 2u : 1u);
  if (blk < img_total) {
    const int which = blk >= a.img_blocks ? 1 : 0;
    if (which) blk -= a.img_blocks;
    const float* __restrict__ src = a.img[which];
    float* __restrict__ dst = a.img_out[which];
    const long long total = (long long)a.n_img * H * W;
    const float sy = (d.y2 - d.y1) / (float)(H > 1 ? H - 1 : 1), sx = (d.x2 - d.x1) / (float)(W > 1 ? W - 1 : 1);
    // (pixel index in 32 bits -- the launcher refuses 2^31 pixels or more --: the three 64-bit divisions of the decomposition
    //  were two thirds of this kernel's instructions)
    const unsigned total32 = (unsigned)total, step32 = a.img_blocks * 256u;
    for (unsigned p = blk * 256u + threadIdx.x; p < total32; p += step32) {
      unsigned j_, i_;
      const unsigned r = xpt_divmod(p, (unsigned)W, j_);
      const unsigned n = xpt_divmod(r, (unsigned)H, i_);
      const int j = (int)j_, i = (int)i_;
      const int jj = d.flip ? W - 1 - j : j;
      // corner-aligned sample position (crop_and_resize): y = (y1 + (y2 - y1) i / (H - 1)) (H - 1)
      const float fy = (d.y1 + sy * (float)i) * (float)(H - 1), fx = (d.x1 + sx * (float)jj) * (float)(W - 1);
      const float y0f = floorf(fy), x0f = floorf(fx);
      const int y0 = (int)y0f, x0 = (int)x0f;
      const float wy = fy - y0f, wx = fx - x0f;
      float c[3] = {0.f, 0.f, 0.f};
      const float* base = src + (long long)n * H * W * 3;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
        const float wgt = ((t >> 1) ? wy : 1.f - wy) * ((t & 1) ? wx : 1.f - wx);
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const float* q = base + ((long long)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)) * 3;
        const float w_ = ok ? wgt : 0.f;
        c[0] += q[0] * w_; c[1] += q[1] * w_; c[2] += q[2] * w_;
      }
      if (d.jit) jitter(c, d.gamma, d.sat);
      float* o = dst + (long long)p * 3;
      o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
    }
    return;
  }
  blk -= img_total;
  if (blk < a.depth_blocks) {                       // ground-truth depth: nearest sample of the same box, never flipped
    const long long total = (long long)a.n_depth * H * W;
    const float sy = (d.y2 - d.y1) / (float)(H > 1 ? H - 1 : 1), sx = (d.x2 - d.x1) / (float)(W > 1 ? W - 1 : 1);
    const unsigned total32 = (unsigned)total, step32 = a.depth_blocks * 256u;
    for (unsigned p = blk * 256u + threadIdx.x; p < total32; p += step32) {
      unsigned j_, i_;
      const unsigned r = xpt_divmod(p, (unsigned)W, j_);
      const unsigned n = xpt_divmod(r, (unsigned)H, i_);
      const int j = (int)j_, i = (int)i_;
      const float fy = (d.y1 + sy * (float)i) * (float)(H - 1), fx = (d.x1 + sx * (float)j) * (float)(W - 1);
      const int yy = (int)nearbyintf(fy), xx = (int)nearbyintf(fx);
      const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
      const float v = a.depth[(long long)n * H * W + (long long)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)];
      a.depth_out[p] = ok ? v : 0.f;
    }
    return;
  }
  // ---- the last workgroup: parameters and the small per-sample tensors
  if (threadIdx.x == 0) {
    a.params[0] = d.y1; a.params[1] = d.x1; a.params[2] = d.y2; a.params[3] = d.x2;
    a.params[4] = d.flip ? 1.f : 0.f; a.params[5] = d.jit ? 1.f : 0.f;
    a.params[6] = d.gamma; a.params[7] = d.sat;
  }
  const float xr = 1.f / (d.x2 - d.x1), yr = 1.f / (d.y2 - d.y1);
  for (int which = 0; which < 2; ++which) {
    if (!a.K[which]) continue;
    for (int e = threadIdx.x; e < a.B * 9; e += 256) {
      const int rc = e % 9, row = rc / 3, col = rc % 3;
      float v = a.K[which][e];
      if (row == 0 && col == 2) v -= d.x1 * (float)W;
      if (row == 1 && col == 2) v -= d.y1 * (float)H;
      if (row == 0) v *= xr;
      if (row == 1) v *= yr;
      if (d.flip) v = fabsf((row == 0 && col == 2 ? (float)W : 0.f) - v);
      a.K_out[which][e] = v;
    }
  }
  for (int which = 0; which < 2; ++which) {
    if (!a.pose[which]) continue;
    for (int e = threadIdx.x; e < a.n_pose * 16; e += 256) {
      const int rc = e % 16, row = rc / 4, col = rc % 4;
      const float v = a.pose[which][e];
      a.pose_out[which][e] = (d.flip && ((row == 0) != (col == 0))) ? -v : v;
    }
  }
  if (a.stereo) {
    for (int e = threadIdx.x; e < a.B * 16; e += 256) {
      const int rc = e % 16, row = rc / 4, col = rc % 4;
      const float v = a.stereo[e];
      a.stereo_out[e] = (d.flip && ((row == 0) != (col == 0))) ? -v : v;
    }
  }
}

// ------------------------------------------------------------------------------------------------ encoder input
// PretrainedModel's preprocessing (model/build_model/pretrained_nets.py:36-43): x = image / 127.5 - 1, bilinear resize to
// (H + 2, W + 2) (TF2 half-pixel centres, no antialias) so that the "valid" stem convolution returns H/2 x W/2 -- written
// straight in the layout that convolution reads: NHWC bf16 with the 3 channels padded to 8 (one 16-byte store per pixel).
// As tensor ops: div, sub, resize, cast, zero fill + copy of the channel pad = 6 launches.
__global__ __launch_bounds__(256) void stem_input_kernel(const float* __restrict__ img, long long batch_stride,
                                                         unsigned short* __restrict__ out, int B, int H, int W) {
  const int OH = H + 2, OW = W + 2;
  const long long total = (long long)B * OH * OW;
  const float ry = (float)H / (float)OH, rx = (float)W / (float)OW;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long long)gridDim.x * 256) {
    unsigned x_, y_;
    const unsigned r = xpt_divmod((unsigned)p, (unsigned)OW, x_);      // (pixel count below 2^31: checked by the launcher)
    const long long b = (long long)xpt_divmod(r, (unsigned)OH, y_);
    const int x = (int)x_, y = (int)y_;
    const float fy = fmaxf(ry * ((float)y + 0.5f) - 0.5f, 0.f), fx = fmaxf(rx * ((float)x + 0.5f) - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const float* base = img + b * batch_stride;
    const float* q00 = base + ((long long)y0 * W + x0) * 3;
    const float* q01 = base + ((long long)y0 * W + x1) * 3;
    const float* q10 = base + ((long long)y1 * W + x0) * 3;
    const float* q11 = base + ((long long)y1 * W + x1) * 3;
    unsigned short v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float a00 = q00[c] / 127.5f - 1.f, a01 = q01[c] / 127.5f - 1.f, a10 = q10[c] / 127.5f - 1.f,
                  a11 = q11[c] / 127.5f - 1.f;
      // at::upsample_bilinear2d: (1 - wy) * ((1 - wx) a00 + wx a01) + wy * ((1 - wx) a10 + wx a11)
      const float t = (1.f - wy) * ((1.f - wx) * a00 + wx * a01) + wy * ((1.f - wx) * a10 + wx * a11);
      v[c] = xpt_f2h(t);
    }
    uint4 pk;
    pk.x = v[0] | ((unsigned)v[1] << 16); pk.y = v[2]; pk.z = 0u; pk.w = 0u;
    *(uint4*)(out + p * 8) = pk;
  }
}

}  // namespace

/* Encoder input of NASNetMobile as the reference prepares it (pretrained_nets.py:36-43): out [B, H+2, W+2, 8] bf16 NHWC
 * (channels 3..7 zero) = resize_bilinear(image / 127.5 - 1, (H+2, W+2)); image: B frames of [H, W, 3] float32, frame b at
 * image + b * batch_stride (elements) -- the target frame of a snippet tensor is read in place. */
extern "C" int xpt_stem_input(const float* image, long long batch_stride, void* out, int B, int H, int W, void* stream) {
  XPT_CHECK_PTR(image); XPT_CHECK_PTR(out);
  if (B <= 0 || H <= 0 || W <= 0 || batch_stride < (long long)H * W * 3) return XPT_ERR_SHAPE;
  if (((uintptr_t)out) % 16 != 0) return XPT_ERR_ARG;
  const long long total = (long long)B * (H + 2) * (W + 2);
  if (total >= (1LL << 31)) return XPT_ERR_SHAPE;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(stem_input_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, image, batch_stride,
                     (unsigned short*)out, B, H, W);
  return xpt_launch_status();
}

/* One launch for TotalAugment([CropAndResize(p_crop), HorizontalFlip(p_flip), ColorJitter(p_jit)]):
 * u [8] uniforms in [0, 1) (device), params [8] out (box y1 x1 y2 x2, flip, jitter, gamma, saturation);
 * img / img_out: image5d (and image5d_R or NULL) as [n_img, H, W, 3] float32; depth / depth_out [n_depth, H, W] or NULL;
 * K / K_out [B, 3, 3] (second pair NULL without a right camera); pose / pose_out [n_pose, 4, 4] (NULL when the dataset has
 * no pose_gt); stereo / stereo_out [B, 4, 4] or NULL.  Outputs must not alias inputs. */
static const float* g_aug_pin = nullptr;

/* Pinned draws: every later xpt_augment launch carries `pin` (device float[9], or NULL to stop) and uses pin[1..8] as its
 * uniforms WHILE pin[0] > 0.5 (decided on the device, at run time): a captured training step can then be replayed with the
 * same draws -- its replay check compares replays with each other and with an eager step -- by flipping pin[0], without
 * re-capturing.  The buffer must outlive the captured graphs. */
extern "C" int xpt_augment_pin(const float* pin) {
  g_aug_pin = pin;
  return XPT_OK;
}

extern "C" int xpt_augment(const float* u, float* params, const float* img0, float* img0_out, const float* img1,
                           float* img1_out, int n_img, const float* depth, float* depth_out, int n_depth, const float* K0,
                           float* K0_out, const float* K1, float* K1_out, int B, const float* pose0, float* pose0_out,
                           const float* pose1, float* pose1_out, int n_pose, const float* stereo, float* stereo_out, int H,
                           int W, float p_crop, float p_flip, float p_jit, float half_crop, void* stream) {
  XPT_CHECK_PTR(u); XPT_CHECK_PTR(params); XPT_CHECK_PTR(img0); XPT_CHECK_PTR(img0_out);
  if (n_img <= 0 || H <= 0 || W <= 0 || B <= 0) return XPT_ERR_SHAPE;
  if ((img1 == nullptr) != (img1_out == nullptr) || (depth == nullptr) != (depth_out == nullptr) ||
      (K0 == nullptr) != (K0_out == nullptr) || (K1 == nullptr) != (K1_out == nullptr) ||
      (pose0 == nullptr) != (pose0_out == nullptr) || (pose1 == nullptr) != (pose1_out == nullptr) ||
      (stereo == nullptr) != (stereo_out == nullptr))
    return XPT_ERR_NULL;
  if (!(p_crop > 0.f) || (long long)n_img * H * W >= (1LL << 31) - (1LL << 24) || (long long)n_depth * H * W >= (1LL << 31) - (1LL << 24))
    return XPT_ERR_ARG;
  AugArgs a{};
  a.pin = g_aug_pin;
  a.u = u; a.params = params;
  a.img[0] = img0; a.img_out[0] = img0_out; a.img[1] = img1; a.img_out[1] = img1_out;
  a.depth = depth; a.depth_out = depth_out;
  a.K[0] = K0; a.K_out[0] = K0_out; a.K[1] = K1; a.K_out[1] = K1_out;
  a.pose[0] = pose0; a.pose_out[0] = pose0_out; a.pose[1] = pose1; a.pose_out[1] = pose1_out;
  a.stereo = stereo; a.stereo_out = stereo_out;
  a.n_img = n_img; a.n_depth = depth ? n_depth : 0; a.B = B; a.n_pose = n_pose; a.H = H; a.W = W;
  a.p_crop = p_crop; a.p_flip = p_flip; a.p_jit = p_jit; a.half_crop = half_crop;
  long long ib = ((long long)n_img * H * W + 255) / 256;
  a.img_blocks = (unsigned)(ib > 8192 ? 8192 : ib);
  long long db = depth ? ((long long)n_depth * H * W + 255) / 256 : 0;
  a.depth_blocks = (unsigned)(db > 2048 ? 2048 : db);
  const unsigned grid = a.img_blocks * (img1 ? 2u : 1u) + a.depth_blocks + 1u;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(augment_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return xpt_launch_status();
}
