// xpt_loss.hip -- photometric L1 / L2 / SSIM (K4, K5) and edge-aware smoothness (K6) for gfx950,
// stand-alone (per-pixel) form used behind the reference's separate callables.  The fused
// warp+loss march kernels live in xpt_fused.hip.  C ABI: include/xpt_hip.h.
#include "xpt_common.h"

using namespace xpt;

#define SSIM_C1 (0.01f * 0.01f)
#define SSIM_C2 (0.03f * 0.03f)

// error_mask = (mean_c synth == 0)  (loss_util.py:15-16 / 38-39 / 64-65)
__device__ inline bool black_pixel(const float y[3]) { return ((y[0] + y[1]) + y[2]) / 3.0f == 0.f; }

// 3x3 SAME window statistics around (i,j) of channel c; divisor excludes padding
// (tf.keras.layers.AveragePooling3D(pool=(1,3,3), padding="SAME"), loss_util.py:78).
struct SsimStat {
  float mux, muy, sx, sy, sxy, inv_cnt;
};

__device__ inline SsimStat ssim_stat(const float* __restrict__ yimg, const float* __restrict__ ximg, int h, int w,
                                     int i, int j, int c) {
  float Sx = 0.f, Sy = 0.f, Sxx = 0.f, Syy = 0.f, Sxy = 0.f;
  int cnt = 0;
#pragma unroll
  for (int di = -1; di <= 1; ++di) {
    const int ii = i + di;
    if (ii < 0 || ii >= h) continue;
#pragma unroll
    for (int dj = -1; dj <= 1; ++dj) {
      const int jj = j + dj;
      if (jj < 0 || jj >= w) continue;
      const long long o = ((long long)ii * w + jj) * 3 + c;
      const float x = ximg[o], y = yimg[o];
      Sx += x; Sy += y; Sxx += x * x; Syy += y * y; Sxy += x * y;
      ++cnt;
    }
  }
  SsimStat s;
  s.inv_cnt = 1.0f / (float)cnt;
  s.mux = Sx * s.inv_cnt;
  s.muy = Sy * s.inv_cnt;
  s.sx = Sxx * s.inv_cnt - s.mux * s.mux;
  s.sy = Syy * s.inv_cnt - s.muy * s.muy;
  s.sxy = Sxy * s.inv_cnt - s.mux * s.muy;
  return s;
}

__device__ inline float ssim_value(const SsimStat& s) {
  const float n = (2.f * s.mux * s.muy + SSIM_C1) * (2.f * s.sxy + SSIM_C2);
  const float d = (s.mux * s.mux + s.muy * s.muy + SSIM_C1) * (s.sx + s.sy + SSIM_C2);
  return n / d;
}

// ------------------------------------------------------------------ forward (all methods)
// grid (ceil(P/256), B*N).  part[bn][blk] = block sum of the per-pixel loss over 3 channels.
template <int METHOD>
__global__ void photo_fwd_kernel(const float* __restrict__ synth, const float* __restrict__ target,
                                 float* __restrict__ map, float* __restrict__ part, int N, int h, int w) {
  __shared__ float red[16];
  const int bn = blockIdx.y, b = bn / N;
  const int P = h * w;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const float* yimg = synth + (long long)bn * P * 3;
  const float* ximg = target + (long long)b * P * 3;
  float acc = 0.f;
  if (p < P) {
    const float y[3] = {yimg[3 * (long long)p], yimg[3 * (long long)p + 1], yimg[3 * (long long)p + 2]};
    const bool black = black_pixel(y);
    const int i = p / w, j = p - i * w;
    float e[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = ximg[3 * (long long)p + c];
      if (METHOD == XPT_PHOTO_L1) {
        e[c] = fabsf(y[c] - x);
      } else if (METHOD == XPT_PHOTO_L2) {
        e[c] = (y[c] - x) * (y[c] - x);
      } else {
        const SsimStat s = ssim_stat(yimg, ximg, h, w, i, j, c);
        e[c] = clampf((1.f - ssim_value(s)) * 0.5f, 0.f, 1.f);
      }
      if (black) e[c] = 0.f;
    }
    if (map) {
      float* m = map + ((long long)bn * P + p) * 3;
      m[0] = e[0]; m[1] = e[1]; m[2] = e[2];
    }
    acc = (e[0] + e[1]) + e[2];
  }
  if (part) {  // uniform branch
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) part[(long long)bn * gridDim.x + blockIdx.x] = s;
  }
}

// loss[b] = (sum_n sum_blk part) * inv_count, fixed order.
__global__ void photo_reduce_kernel(const float* __restrict__ part, float* __restrict__ loss, int B, int per_b,
                                    float inv_count) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* q = part + (long long)b * per_b;
  float s = 0.f;
  for (int k = 0; k < per_b; ++k) s += q[k];
  loss[b] = s * inv_count;
}

// ------------------------------------------------------------------ backward L1 / L2
template <int METHOD>
__global__ void photo_bwd_pointwise_kernel(const float* __restrict__ synth, const float* __restrict__ target,
                                           const float* __restrict__ gloss, const float* __restrict__ gmap,
                                           float* __restrict__ dsynth, int N, int h, int w, float inv_count) {
  const int bn = blockIdx.y, b = bn / N;
  const int P = h * w;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const long long o = ((long long)bn * P + p) * 3;
  const float y[3] = {synth[o], synth[o + 1], synth[o + 2]};
  const bool black = black_pixel(y);
  const float gs = gloss ? gloss[b] * inv_count : 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float x = target[((long long)b * P + p) * 3 + c];
    const float g = gmap ? gmap[o + c] : gs;
    const float df = y[c] - x;
    float d;
    if (METHOD == XPT_PHOTO_L1) d = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
    else d = 2.f * df;
    dsynth[o + c] = black ? 0.f : g * d;
  }
}

// ------------------------------------------------------------------ backward SSIM, pass A: per-pixel coefficients
// For output pixel p, channel c:  L_p = clip((1-ssim_p)/2, 0, 1) (0 where the pixel is black), and
//   dL_p/dy_q = g_p * (-1/2) * [ dssim/dmu_y + 2 y_q dssim/dEyy + x_q dssim/dExy ] / cnt_p  for q in win(p).
// coef[bn][p][3*c + {0,1,2}] = g_p * (-1/2) / cnt_p * {dssim/dmu_y, dssim/dEyy, dssim/dExy}.
__global__ void ssim_coef_kernel(const float* __restrict__ synth, const float* __restrict__ target,
                                 const float* __restrict__ gloss, const float* __restrict__ gmap,
                                 float* __restrict__ coef, int N, int h, int w, float inv_count) {
  const int bn = blockIdx.y, b = bn / N;
  const int P = h * w;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float* yimg = synth + (long long)bn * P * 3;
  const float* ximg = target + (long long)b * P * 3;
  const float y[3] = {yimg[3 * (long long)p], yimg[3 * (long long)p + 1], yimg[3 * (long long)p + 2]};
  const bool black = black_pixel(y);
  const int i = p / w, j = p - i * w;
  const float gs = gloss ? gloss[b] * inv_count : 0.f;
  float* co = coef + ((long long)bn * P + p) * 9;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const SsimStat s = ssim_stat(yimg, ximg, h, w, i, j, c);
    const float n1 = 2.f * s.mux * s.muy + SSIM_C1, n2 = 2.f * s.sxy + SSIM_C2;
    const float d1 = s.mux * s.mux + s.muy * s.muy + SSIM_C1, d2 = s.sx + s.sy + SSIM_C2;
    const float inv12 = 1.0f / (d1 * d2);
    const float ssim = n1 * n2 * inv12;
    const float val = (1.f - ssim) * 0.5f;
    float g = gmap ? gmap[((long long)bn * P + p) * 3 + c] : gs;
    if (black || !(val >= 0.f && val <= 1.f)) g = 0.f;  // where(mask) + clip_by_value gradient
    g *= -0.5f * s.inv_cnt;
    const float dmu = 2.f * s.mux * (n2 - n1) * inv12 - 2.f * s.muy * ssim * (1.0f / d1 - 1.0f / d2);
    const float dEyy = -ssim / d2;
    const float dExy = 2.f * n1 * inv12;
    co[3 * c + 0] = g * dmu;
    co[3 * c + 1] = g * dEyy;
    co[3 * c + 2] = g * dExy;
  }
}

// pass B: dsynth(q,c) = sum_{p in win(q)} (A_p + 2 B_p y_q + C_p x_q)
__global__ void ssim_gather_kernel(const float* __restrict__ synth, const float* __restrict__ target,
                                   const float* __restrict__ coef, float* __restrict__ dsynth, int N, int h, int w) {
  const int bn = blockIdx.y, b = bn / N;
  const int P = h * w;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= P) return;
  const int i = q / w, j = q - i * w;
  float SA[3] = {0.f, 0.f, 0.f}, SB[3] = {0.f, 0.f, 0.f}, SC[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int di = -1; di <= 1; ++di) {
    const int ii = i + di;
    if (ii < 0 || ii >= h) continue;
#pragma unroll
    for (int dj = -1; dj <= 1; ++dj) {
      const int jj = j + dj;
      if (jj < 0 || jj >= w) continue;
      const float* co = coef + ((long long)bn * P + (long long)ii * w + jj) * 9;
#pragma unroll
      for (int c = 0; c < 3; ++c) { SA[c] += co[3 * c]; SB[c] += co[3 * c + 1]; SC[c] += co[3 * c + 2]; }
    }
  }
  const long long o = ((long long)bn * P + q) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float y = synth[o + c], x = target[((long long)b * P + q) * 3 + c];
    dsynth[o + c] = SA[c] + 2.f * SB[c] * y + SC[c] * x;
  }
}

// =================================================================== K6: edge-aware smoothness
// smootheness_loss (losses.py:409-440) on disparity; input_is_depth fuses safe_reciprocal_number
// (util_funcs.py:157-160): disp = (1/d) * [d > 1e-5].
__device__ inline float to_disp(float v, int is_depth) {
  if (!is_depth) return v;
  return (v > 0.00001f) ? 1.0f / v : 0.f;
}

__device__ inline float edge_weight(const float* __restrict__ a, const float* __restrict__ b, float gf) {
  const float m = ((fabsf((a[0] - b[0]) * gf) + fabsf((a[1] - b[1]) * gf)) + fabsf((a[2] - b[2]) * gf)) / 3.0f;
  return expf(-m);
}

// part[b][blk][2] = (sum |gx wx|, sum |gy wy|)
__device__ __forceinline__ void smooth_fwd_body(const float* __restrict__ disp, const float* __restrict__ image,
                                                float* __restrict__ part, int h, int w, float gf, int is_depth,
                                                int nblk, float* red) {
  const int b = blockIdx.y;
  const int P = h * w;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  float acc[2] = {0.f, 0.f};
  if (p < P) {
    const int i = p / w, j = p - i * w;
    const float* dz = disp + (long long)b * P;
    const float* im = image + (long long)b * P * 3;
    const float d0 = to_disp(dz[p], is_depth);
    if (j < w - 1) {
      const float gx = d0 - to_disp(dz[p + 1], is_depth);
      acc[0] = fabsf(gx * edge_weight(im + 3 * (long long)p, im + 3 * (long long)(p + 1), gf));
    }
    if (i < h - 1) {
      const float gy = d0 - to_disp(dz[p + w], is_depth);
      acc[1] = fabsf(gy * edge_weight(im + 3 * (long long)p, im + 3 * (long long)(p + w), gf));
    }
  }
  block_sum_n<2>(acc, red);
  if (threadIdx.x == 0) {
    part[((long long)b * nblk + blockIdx.x) * 2 + 0] = acc[0];
    part[((long long)b * nblk + blockIdx.x) * 2 + 1] = acc[1];
  }
}

__global__ void smooth_fwd_kernel(const float* __restrict__ disp, const float* __restrict__ image,
                                  float* __restrict__ part, int h, int w, float gf, int is_depth) {
  __shared__ float red[4 * 2];
  smooth_fwd_body(disp, image, part, h, w, gf, is_depth, gridDim.x, red);
}

// All scales of the smoothness loss in one launch: blockIdx.z picks the scale, workgroups past a scale's own
// block count leave at once (the grid is sized for the largest scale).
struct SmoothMs {
  const float* disp[4];
  const float* image[4];
  float* part[4];        // forward partials / backward: d(input)
  const float* gloss[4];
  float* loss[4];
  int h[4], w[4], nblk[4];
  float gf;
  int is_depth, B;
};

__global__ void smooth_fwd_ms_kernel(SmoothMs a) {
  __shared__ float red[4 * 2];
  const int s = blockIdx.z;
  if ((int)blockIdx.x >= a.nblk[s]) return;
  smooth_fwd_body(a.disp[s], a.image[s], a.part[s], a.h[s], a.w[s], a.gf, a.is_depth, a.nblk[s], red);
}

// one wave per sample: lane l adds the partials l, l + 64, ... (all its loads independent), then a fixed shuffle tree --
// deterministic, and the same order in the per-scale and the multi-scale launch
__device__ __forceinline__ void smooth_reduce_body(const float* __restrict__ part, int nblk, float inv_x, float inv_y,
                                                   float* __restrict__ out) {
  float sx = 0.f, sy = 0.f;
  for (int k = threadIdx.x; k < nblk; k += 64) {
    sx += part[2 * k];
    sy += part[2 * k + 1];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sx += __shfl_down(sx, off, 64);
    sy += __shfl_down(sy, off, 64);
  }
  if (threadIdx.x == 0) *out = 0.5f * (sx * inv_x) + 0.5f * (sy * inv_y);
}

__global__ __launch_bounds__(64) void smooth_reduce_kernel(const float* __restrict__ part, float* __restrict__ loss, int B,
                                                          int nblk, float inv_x, float inv_y) {
  const int b = blockIdx.x;
  smooth_reduce_body(part + (long long)b * nblk * 2, nblk, inv_x, inv_y, loss + b);
}

// (the one-thread-per-sample loop this replaces walked up to 208 dependent iterations: 18 us)
__global__ __launch_bounds__(64) void smooth_reduce_ms_kernel(SmoothMs a) {
  const int s = blockIdx.y;
  const int b = blockIdx.x;
  const int nblk = a.nblk[s];
  const float inv_x = 1.0f / ((float)a.h[s] * (float)(a.w[s] - 1)), inv_y = 1.0f / ((float)(a.h[s] - 1) * (float)a.w[s]);
  smooth_reduce_body(a.part[s] + (long long)b * nblk * 2, nblk, inv_x, inv_y, a.loss[s] + b);
}

__device__ inline float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__device__ __forceinline__ void smooth_bwd_body(const float* __restrict__ disp, const float* __restrict__ image,
                                                const float* __restrict__ gloss, float* __restrict__ dinput, int h,
                                                int w, float gf, int is_depth, float inv_x, float inv_y) {
  const int b = blockIdx.y;
  const int P = h * w;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int i = p / w, j = p - i * w;
  const float* dz = disp + (long long)b * P;
  const float* im = image + (long long)b * P * 3;
  const float raw = dz[p];
  const float d0 = to_disp(raw, is_depth);
  float gxs = 0.f, gys = 0.f;
  if (j < w - 1) {
    const float wx = edge_weight(im + 3 * (long long)p, im + 3 * (long long)(p + 1), gf);
    gxs += sgn((d0 - to_disp(dz[p + 1], is_depth)) * wx) * wx;
  }
  if (j > 0) {
    const float wx = edge_weight(im + 3 * (long long)(p - 1), im + 3 * (long long)p, gf);
    gxs -= sgn((to_disp(dz[p - 1], is_depth) - d0) * wx) * wx;
  }
  if (i < h - 1) {
    const float wy = edge_weight(im + 3 * (long long)p, im + 3 * (long long)(p + w), gf);
    gys += sgn((d0 - to_disp(dz[p + w], is_depth)) * wy) * wy;
  }
  if (i > 0) {
    const float wy = edge_weight(im + 3 * (long long)(p - w), im + 3 * (long long)p, gf);
    gys -= sgn((to_disp(dz[p - w], is_depth) - d0) * wy) * wy;
  }
  float g = gloss[b] * 0.5f * (gxs * inv_x + gys * inv_y);
  if (is_depth) g = (raw > 0.00001f) ? -g / (raw * raw) : 0.f;
  dinput[(long long)b * P + p] = g;
}

__global__ void smooth_bwd_kernel(const float* __restrict__ disp, const float* __restrict__ image,
                                  const float* __restrict__ gloss, float* __restrict__ dinput, int h, int w, float gf,
                                  int is_depth, float inv_x, float inv_y) {
  smooth_bwd_body(disp, image, gloss, dinput, h, w, gf, is_depth, inv_x, inv_y);
}

__global__ void smooth_bwd_ms_kernel(SmoothMs a) {
  const int s = blockIdx.z;
  if ((int)blockIdx.x >= a.nblk[s]) return;
  smooth_bwd_body(a.disp[s], a.image[s], a.gloss[s], a.part[s], a.h[s], a.w[s], a.gf, a.is_depth,
                  1.0f / ((float)a.h[s] * (float)(a.w[s] - 1)), 1.0f / ((float)(a.h[s] - 1) * (float)a.w[s]));
}

// =================================================================== total-loss merge (TotalLoss.__call__, losses.py:44-55)
// total = sum_t c[t] * rowsum_t,  by_type[k] = sum_t a[k][t] * rowsum_t,  rowsum_t = sum_b term_t[b]: one workgroup, fixed
// order (what torch does with stack + sum + dot + mv: four launches); backward: grad_t[b] = c[t] * g_total.
struct MergeTerms {
  const float* term[64];
  int n, batch, types;
};

__global__ void merge_total_fwd_kernel(MergeTerms a, const float* __restrict__ c, const float* __restrict__ amat,
                                       float* __restrict__ total, float* __restrict__ by_type) {
  __shared__ float rs[64];
  const int t = threadIdx.x;
  if (t < a.n) {
    float s = 0.f;
    for (int b = 0; b < a.batch; ++b) s += a.term[t][b];
    rs[t] = s;
  }
  __syncthreads();
  if (t == 0) {
    float s = 0.f;
    for (int j = 0; j < a.n; ++j) s += c[j] * rs[j];
    total[0] = s;
  }
  if (t >= 64 && t - 64 < a.types) {
    const int k = t - 64;
    float s = 0.f;
    for (int j = 0; j < a.n; ++j) s += amat[k * a.n + j] * rs[j];
    by_type[k] = s;
  }
}

__global__ void merge_total_bwd_kernel(const float* __restrict__ c, const float* __restrict__ g_total,
                                       float* __restrict__ grads, int n, int batch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * batch) grads[i] = c[i / batch] * g_total[0];
}

// =================================================================== C ABI
extern "C" {

int xpt_merge_total_fwd(int n, const float* const* terms, const float* c, const float* amat, float* total,
                        float* by_type, int batch, int types, void* stream) {
  XPT_CHECK_PTR(terms); XPT_CHECK_PTR(c); XPT_CHECK_PTR(amat); XPT_CHECK_PTR(total); XPT_CHECK_PTR(by_type);
  if (n < 1 || n > 64 || batch < 1 || types < 1 || types > 64) return XPT_ERR_SHAPE;
  MergeTerms a = {};
  for (int t = 0; t < n; ++t) {
    if (!terms[t]) return XPT_ERR_NULL;
    a.term[t] = terms[t];
  }
  a.n = n; a.batch = batch; a.types = types;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(merge_total_fwd_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, a, c, amat, total, by_type);
  return xpt_launch_status();
}

int xpt_merge_total_bwd(int n, const float* c, const float* g_total, float* grads, int batch, void* stream) {
  XPT_CHECK_PTR(c); XPT_CHECK_PTR(g_total); XPT_CHECK_PTR(grads);
  if (n < 1 || n > 64 || batch < 1) return XPT_ERR_SHAPE;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(merge_total_bwd_kernel, dim3((n * batch + 255) / 256), dim3(256), 0, (hipStream_t)stream, c, g_total,
                     grads, n, batch);
  return xpt_launch_status();
}

size_t xpt_photo_workspace_floats(int B, int N, int h, int w) {
  if (B <= 0 || N <= 0 || h <= 0 || w <= 0) return 0;
  return (size_t)B * N * h * w * 9;  // SSIM backward coefficients; also covers the forward partials
}

int xpt_photo_fwd(int method, const float* synth, const float* target, float* map, float* loss, float* workspace,
                  size_t workspace_floats, int B, int N, int h, int w, void* stream) {
  XPT_CHECK_PTR(synth); XPT_CHECK_PTR(target);
  if (!map && !loss) return XPT_ERR_NULL;
  if (B <= 0 || N <= 0 || h <= 0 || w <= 0 || (long long)B * N > 65535) return XPT_ERR_SHAPE;
  if (method < 0 || method > 2) return XPT_ERR_ARG;
  const int P = h * w, nblk = (P + 255) / 256;
  float* part = nullptr;
  if (loss) {
    XPT_CHECK_PTR(workspace);
    if (workspace_floats < (size_t)B * N * nblk) return XPT_ERR_WORKSPACE;
    part = workspace;
  }
  const dim3 grid(nblk, B * N), block(256);
  XPT_BEGIN_LAUNCH();
  hipStream_t s = (hipStream_t)stream;
  if (method == XPT_PHOTO_L1) hipLaunchKernelGGL(photo_fwd_kernel<XPT_PHOTO_L1>, grid, block, 0, s, synth, target, map, part, N, h, w);
  else if (method == XPT_PHOTO_L2) hipLaunchKernelGGL(photo_fwd_kernel<XPT_PHOTO_L2>, grid, block, 0, s, synth, target, map, part, N, h, w);
  else hipLaunchKernelGGL(photo_fwd_kernel<XPT_PHOTO_SSIM>, grid, block, 0, s, synth, target, map, part, N, h, w);
  if (loss)
    hipLaunchKernelGGL(photo_reduce_kernel, dim3((B + 63) / 64), dim3(64), 0, s, part, loss, B, N * nblk,
                       1.0f / ((float)N * (float)P * 3.0f));
  return xpt_launch_status();
}

int xpt_photo_bwd(int method, const float* synth, const float* target, const float* gloss, const float* gmap,
                  float* dsynth, float* workspace, size_t workspace_floats, int B, int N, int h, int w,
                  void* stream) {
  XPT_CHECK_PTR(synth); XPT_CHECK_PTR(target); XPT_CHECK_PTR(dsynth);
  if ((gloss == nullptr) == (gmap == nullptr)) return XPT_ERR_ARG;
  if (B <= 0 || N <= 0 || h <= 0 || w <= 0 || (long long)B * N > 65535) return XPT_ERR_SHAPE;
  if (method < 0 || method > 2) return XPT_ERR_ARG;
  const int P = h * w, nblk = (P + 255) / 256;
  const float inv_count = 1.0f / ((float)N * (float)P * 3.0f);
  const dim3 grid(nblk, B * N), block(256);
  XPT_BEGIN_LAUNCH();
  hipStream_t s = (hipStream_t)stream;
  if (method == XPT_PHOTO_L1) {
    hipLaunchKernelGGL(photo_bwd_pointwise_kernel<XPT_PHOTO_L1>, grid, block, 0, s, synth, target, gloss, gmap, dsynth, N, h, w, inv_count);
  } else if (method == XPT_PHOTO_L2) {
    hipLaunchKernelGGL(photo_bwd_pointwise_kernel<XPT_PHOTO_L2>, grid, block, 0, s, synth, target, gloss, gmap, dsynth, N, h, w, inv_count);
  } else {
    XPT_CHECK_PTR(workspace);
    if (workspace_floats < xpt_photo_workspace_floats(B, N, h, w)) return XPT_ERR_WORKSPACE;
    hipLaunchKernelGGL(ssim_coef_kernel, grid, block, 0, s, synth, target, gloss, gmap, workspace, N, h, w, inv_count);
    hipLaunchKernelGGL(ssim_gather_kernel, grid, block, 0, s, synth, target, workspace, dsynth, N, h, w);
  }
  return xpt_launch_status();
}

size_t xpt_smooth_workspace_floats(int B, int h, int w) {
  if (B <= 0 || h <= 0 || w <= 0) return 0;
  return (size_t)B * (((size_t)h * w + 255) / 256) * 2;
}

int xpt_smooth_fwd(const float* disp, const float* image, float* loss, float* workspace, size_t workspace_floats,
                   int B, int h, int w, float grad_factor, int input_is_depth, void* stream) {
  XPT_CHECK_PTR(disp); XPT_CHECK_PTR(image); XPT_CHECK_PTR(loss); XPT_CHECK_PTR(workspace);
  if (B <= 0 || h < 2 || w < 2 || B > 65535) return XPT_ERR_SHAPE;
  if (workspace_floats < xpt_smooth_workspace_floats(B, h, w)) return XPT_ERR_WORKSPACE;
  const int P = h * w, nblk = (P + 255) / 256;
  XPT_BEGIN_LAUNCH();
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(smooth_fwd_kernel, dim3(nblk, B), dim3(256), 0, s, disp, image, workspace, h, w, grad_factor,
                     input_is_depth);
  hipLaunchKernelGGL(smooth_reduce_kernel, dim3(B), dim3(64), 0, s, workspace, loss, B, nblk,
                     1.0f / ((float)h * (float)(w - 1)), 1.0f / ((float)(h - 1) * (float)w));
  return xpt_launch_status();
}

int xpt_smooth_bwd(const float* disp, const float* image, const float* gloss, float* dinput, int B, int h, int w,
                   float grad_factor, int input_is_depth, void* stream) {
  XPT_CHECK_PTR(disp); XPT_CHECK_PTR(image); XPT_CHECK_PTR(gloss); XPT_CHECK_PTR(dinput);
  if (B <= 0 || h < 2 || w < 2 || B > 65535) return XPT_ERR_SHAPE;
  const int P = h * w, nblk = (P + 255) / 256;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(smooth_bwd_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, disp, image, gloss, dinput,
                     h, w, grad_factor, input_is_depth, 1.0f / ((float)h * (float)(w - 1)),
                     1.0f / ((float)(h - 1) * (float)w));
  return xpt_launch_status();
}

static int smooth_ms_fill(SmoothMs& a, int nscales, const float* const* disp, const float* const* image, int B,
                          const int* h, const int* w, float grad_factor, int input_is_depth, int* max_blk) {
  if (!disp || !image || !h || !w) return XPT_ERR_NULL;
  if (nscales < 1 || nscales > 4 || B <= 0 || B > 65535) return XPT_ERR_SHAPE;
  *max_blk = 0;
  for (int s = 0; s < nscales; ++s) {
    if (!disp[s] || !image[s]) return XPT_ERR_NULL;
    if (h[s] < 2 || w[s] < 2) return XPT_ERR_SHAPE;
    a.disp[s] = disp[s]; a.image[s] = image[s]; a.h[s] = h[s]; a.w[s] = w[s];
    a.nblk[s] = (h[s] * w[s] + 255) / 256;
    if (a.nblk[s] > *max_blk) *max_blk = a.nblk[s];
  }
  a.gf = grad_factor; a.is_depth = input_is_depth; a.B = B;
  return XPT_OK;
}

int xpt_smooth_ms_fwd(int nscales, const float* const* disp, const float* const* image, float* losses,
                      float* workspace, size_t workspace_floats, int B, const int* h, const int* w, float grad_factor,
                      int input_is_depth, void* stream) {
  XPT_CHECK_PTR(losses); XPT_CHECK_PTR(workspace);
  SmoothMs a = {};
  int max_blk = 0;
  const int rc = smooth_ms_fill(a, nscales, disp, image, B, h, w, grad_factor, input_is_depth, &max_blk);
  if (rc != XPT_OK) return rc;
  size_t need = 0;
  for (int s = 0; s < nscales; ++s) {
    a.part[s] = workspace + need;
    a.loss[s] = losses + (size_t)s * B;
    need += xpt_smooth_workspace_floats(B, h[s], w[s]);
  }
  if (workspace_floats < need) return XPT_ERR_WORKSPACE;
  XPT_BEGIN_LAUNCH();
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(smooth_fwd_ms_kernel, dim3(max_blk, B, nscales), dim3(256), 0, st, a);
  hipLaunchKernelGGL(smooth_reduce_ms_kernel, dim3(B, nscales), dim3(64), 0, st, a);
  return xpt_launch_status();
}

int xpt_smooth_ms_bwd(int nscales, const float* const* disp, const float* const* image, const float* const* gloss,
                      float* const* dinput, int B, const int* h, const int* w, float grad_factor, int input_is_depth,
                      void* stream) {
  if (!gloss || !dinput) return XPT_ERR_NULL;
  SmoothMs a = {};
  int max_blk = 0;
  const int rc = smooth_ms_fill(a, nscales, disp, image, B, h, w, grad_factor, input_is_depth, &max_blk);
  if (rc != XPT_OK) return rc;
  for (int s = 0; s < nscales; ++s) {
    if (!gloss[s] || !dinput[s]) return XPT_ERR_NULL;
    a.gloss[s] = gloss[s]; a.part[s] = dinput[s];
  }
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(smooth_bwd_ms_kernel, dim3(max_blk, B, nscales), dim3(256), 0, (hipStream_t)stream, a);
  return xpt_launch_status();
}

}  // extern "C"
