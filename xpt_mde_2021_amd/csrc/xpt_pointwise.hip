// xpt_pointwise.hip -- per-channel epilogues of the convolution layers on NHWC activations for gfx950:
//   bias + LeakyReLU / linear      (CustomConv2D layers of DepthNet decoder and PoseNet: model/model_util/layer_ops.py:31-35,
//                                   activation LeakyReLU(0.1) per config-example.py:56-63)
//   inference-mode BatchNorm affine (+ ReLU on the way in) of the NASNet cells (keras BatchNormalization called without
//                                   training=True, model/train_val.py:82; reference call site pretrained_nets.py:36-44)
// Both are one streaming pass forward and one streaming pass backward (dx plus the per-channel parameter gradients
// reduced deterministically: per-workgroup partials, then a fixed-order sum) instead of the 2-5 library launches per
// layer (bias add, activation, their backward kernels and a separate reduction) of the generic framework path.
#include <hip/hip_bf16.h>

#include "xpt_common.h"

namespace {

template <typename T> __device__ inline float ldf(const T* p);
template <> __device__ inline float ldf<float>(const float* p) { return *p; }
template <> __device__ inline float ldf<__hip_bfloat16>(const __hip_bfloat16* p) { return __bfloat162float(*p); }
template <typename T> __device__ inline void stf(T* p, float v);
template <> __device__ inline void stf<float>(float* p, float v) { *p = v; }
template <> __device__ inline void stf<__hip_bfloat16>(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }

// ---------------------------------------------------------------- y = act(f(x) * scale[c] + shift[c])
// scale = gamma * rsqrt(var + eps), shift = beta - mean * scale (gamma == nullptr: scale = 1, shift = beta).
// slope: LeakyReLU negative slope (1 = linear, 0 = ReLU).  relu_in: f(x) = max(x, 0), else f(x) = x.
struct Affine {
  const float* gamma;
  const float* beta;
  const float* mean;
  const float* var;
  float eps;
};

__device__ inline void affine_coeffs(const Affine& a, int c, float& sc, float& sh, float& rstd, float& mu) {
  if (a.gamma) {
    rstd = rsqrtf(a.var[c] + a.eps);
    mu = a.mean[c];
    sc = a.gamma[c] * rstd;
    sh = a.beta[c] - mu * sc;
  } else {
    rstd = 1.f; mu = 0.f; sc = 1.f; sh = a.beta[c];
  }
}

template <typename T>
__global__ void affine_act_fwd_kernel(const T* __restrict__ x, Affine a, T* __restrict__ y, long long n, int C,
                                      float slope, int relu_in) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    float sc, sh, rstd, mu;
    affine_coeffs(a, c, sc, sh, rstd, mu);
    float v = ldf<T>(x + i);
    if (relu_in) v = fmaxf(v, 0.f);
    v = v * sc + sh;
    v = v > 0.f ? v : v * slope;
    stf<T>(y + i, v);
  }
}

// Backward.  Rows = pixels (n / C), lanes = channels: block (64 channels) x (ROWS pixel rows per block).
//   g  = dy * act'(y)            (act' from the OUTPUT sign: valid for slope > 0, and for slope == 0 where y == 0 => 0)
//   dx = g * scale [* (x > 0) if relu_in]
//   dbeta[c] = sum g ; dgamma[c] = rstd * (sum g f(x) - mean * sum g)
// part[blk][2][C] partial sums of (g, g f(x)).
// When C <= 32 the spare lanes of a wave take further rows (RG = 64 / C row groups, folded by fixed-order shuffles).
#define PW_ROWS 32
template <typename T>
__global__ void affine_act_bwd_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ dy,
                                      Affine a, T* __restrict__ dx, float* __restrict__ part,
                                      long long rows, int C, float slope, int relu_in, int need_dscale, int RG,
                                      int final_partials) {
  __shared__ float red[2][3][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rg = (RG > 1) ? lane / C : 0;
  const int cl = (RG > 1) ? lane - rg * C : lane;
  const int c = blockIdx.x * 64 + cl;
  const long long r0 = (long long)blockIdx.y * PW_ROWS;
  float s_shift = 0.f, s_scale = 0.f;
  if (c < C && rg < RG) {
    float sc, sh, rstd, mu;
    affine_coeffs(a, c, sc, sh, rstd, mu);
    for (int i = wid * RG + rg; i < PW_ROWS; i += 4 * RG) {
      const long long r = r0 + i;
      if (r >= rows) break;
      const long long o = r * C + c;
      float g = ldf<T>(dy + o);
      if (slope != 1.f) g = (ldf<T>(y + o) > 0.f) ? g : g * slope;      // y is not read for a linear epilogue
      float xv = 0.f;
      if (need_dscale || relu_in) xv = ldf<T>(x + o);
      float gx = g * sc;
      if (relu_in) {
        if (!(xv > 0.f)) gx = 0.f;
        xv = fmaxf(xv, 0.f);
      }
      if (dx) stf<T>(dx + o, gx);
      s_shift += g;
      s_scale += g * xv;
    }
  }
  if (RG > 1) {
    float v0 = s_shift, v1 = s_scale;
    for (int r = 1; r < RG; ++r) {
      v0 += __shfl(s_shift, (cl + r * C) & 63, 64);
      v1 += __shfl(s_scale, (cl + r * C) & 63, 64);
    }
    s_shift = v0;
    s_scale = v1;
  }
  if (wid > 0) {
    red[0][wid - 1][lane] = s_shift;
    red[1][wid - 1][lane] = s_scale;
  }
  __syncthreads();
  if (wid == 0 && c < C && rg == 0) {
    float* p = part + (long long)blockIdx.y * 2 * C;
    const float s0 = ((s_shift + red[0][0][lane]) + red[0][1][lane]) + red[0][2][lane];
    float s1 = ((s_scale + red[1][0][lane]) + red[1][1][lane]) + red[1][2][lane];
    if (final_partials && need_dscale) {   // this block's share of dgamma itself (the finishing pass only adds)
      float sc, sh, rstd, mu;
      affine_coeffs(a, c, sc, sh, rstd, mu);
      s1 = rstd * (s1 - mu * s0);
    }
    p[c] = s0;
    p[C + c] = s1;
  }
}

// Few channels (C <= 4, e.g. the 1-channel depth heads): one thread per pixel row keeps the C running sums in
// registers; the block then reduces them.  Same outputs / partial layout as the kernel above (gridDim.x == 1).
template <typename T>
__global__ void affine_act_bwd_smallc_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                             const T* __restrict__ dy, Affine a, T* __restrict__ dx,
                                             float* __restrict__ part, long long rows, int C, float slope,
                                             int relu_in, int need_dscale, int final_partials) {
  __shared__ float red[4 * 8];
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const long long r0 = (long long)blockIdx.y * (PW_ROWS * 8);
  for (int i = threadIdx.x; i < PW_ROWS * 8; i += blockDim.x) {
    const long long r = r0 + i;
    if (r >= rows) break;
    for (int c = 0; c < C; ++c) {
      float sc, sh, rstd, mu;
      affine_coeffs(a, c, sc, sh, rstd, mu);
      const long long o = r * C + c;
      float g = ldf<T>(dy + o);
      if (slope != 1.f) g = (ldf<T>(y + o) > 0.f) ? g : g * slope;
      float xv = 0.f;
      if (need_dscale || relu_in) xv = ldf<T>(x + o);
      float gx = g * sc;
      if (relu_in) {
        if (!(xv > 0.f)) gx = 0.f;
        xv = fmaxf(xv, 0.f);
      }
      if (dx) stf<T>(dx + o, gx);
      acc[c] += g;
      acc[4 + c] += g * xv;
    }
  }
  xpt::block_sum_n<8>(acc, red);
  if (threadIdx.x == 0) {
    float* p = part + (long long)blockIdx.y * 2 * C;
    for (int c = 0; c < C; ++c) {
      float s1 = acc[4 + c];
      if (final_partials && need_dscale) {
        float sc, sh, rstd, mu;
        affine_coeffs(a, c, sc, sh, rstd, mu);
        s1 = rstd * (s1 - mu * acc[c]);
      }
      p[c] = acc[c];
      p[C + c] = s1;
    }
  }
}

// per channel c: S0 = sum_k part[k][0][c], S1 = sum_k part[k][1][c]; one wave (64 threads) per channel so that the
// partial rows are fetched with few dependent round trips; fixed tree -> deterministic.
__global__ void affine_finish_kernel(const float* __restrict__ part, Affine a, float* __restrict__ dbeta,
                                     float* __restrict__ dgamma, int C, int nblk) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int t = threadIdx.x & 63;
  float s0 = 0.f, s1 = 0.f;
  if (c < C)
    for (int k = t; k < nblk; k += 64) {
      s0 += part[(long long)k * 2 * C + c];
      s1 += part[(long long)k * 2 * C + C + c];
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s0 += __shfl_down(s0, off, 64);
    s1 += __shfl_down(s1, off, 64);
  }
  if (t == 0 && c < C) {
    dbeta[c] = s0;
    if (dgamma) {
      float sc, sh, rstd, mu;
      affine_coeffs(a, c, sc, sh, rstd, mu);
      dgamma[c] = rstd * (s1 - mu * s0);
    }
  }
}

inline unsigned grid_for(long long total) {
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

}  // namespace

extern "C" {

int xpt_affine_act_fwd(const void* x, const float* gamma, const float* beta, const float* mean, const float* var,
                       float eps, void* y, long long rows, int C, float slope, int relu_in, int dtype, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(beta); XPT_CHECK_PTR(y);
  if (gamma && (!mean || !var)) return XPT_ERR_NULL;
  if (rows <= 0 || C <= 0) return XPT_ERR_SHAPE;
  if (dtype != 0 && dtype != 1) return XPT_ERR_ARG;
  const long long n = rows * C;
  const Affine a{gamma, beta, mean, var, eps};
  XPT_BEGIN_LAUNCH();
  if (dtype == 0)
    hipLaunchKernelGGL(affine_act_fwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)x, a, (float*)y, n, C, slope, relu_in);
  else
    hipLaunchKernelGGL(affine_act_fwd_kernel<__hip_bfloat16>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                       (const __hip_bfloat16*)x, a, (__hip_bfloat16*)y, n, C, slope, relu_in);
  return xpt_launch_status();
}

size_t xpt_affine_act_bwd_workspace_floats(long long rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  return (size_t)((rows + PW_ROWS - 1) / PW_ROWS) * 2 * (size_t)C;   // upper bound of blocks * 2 * C
}

static int affine_bwd_blocks(long long rows, int C) {
  return (int)(C <= 4 ? (rows + PW_ROWS * 8 - 1) / (PW_ROWS * 8) : (rows + PW_ROWS - 1) / PW_ROWS);
}

// dx + per-block partial sums part[blk][2][C]; final_partials: the second row holds the block's share of dgamma
static void affine_bwd_launch(const void* x, const void* y, const void* dy, const Affine& a, void* dx, float* part,
                              long long rows, int C, float slope, int relu_in, int need_dscale, int final_partials,
                              int dtype, hipStream_t s) {
  const int nblk = affine_bwd_blocks(rows, C);
  if (C <= 4) {
    const dim3 g1(1, nblk);
    if (dtype == 0)
      hipLaunchKernelGGL(affine_act_bwd_smallc_kernel<float>, g1, dim3(256), 0, s, (const float*)x, (const float*)y,
                         (const float*)dy, a, (float*)dx, part, rows, C, slope, relu_in, need_dscale, final_partials);
    else
      hipLaunchKernelGGL(affine_act_bwd_smallc_kernel<__hip_bfloat16>, g1, dim3(256), 0, s, (const __hip_bfloat16*)x,
                         (const __hip_bfloat16*)y, (const __hip_bfloat16*)dy, a, (__hip_bfloat16*)dx, part, rows, C,
                         slope, relu_in, need_dscale, final_partials);
    return;
  }
  const dim3 grid((C + 63) / 64, nblk);
  const int RG = (C <= 32) ? (64 / C > PW_ROWS / 4 ? PW_ROWS / 4 : 64 / C) : 1;
  if (dtype == 0)
    hipLaunchKernelGGL(affine_act_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (const float*)y,
                       (const float*)dy, a, (float*)dx, part, rows, C, slope, relu_in, need_dscale, RG, final_partials);
  else
    hipLaunchKernelGGL(affine_act_bwd_kernel<__hip_bfloat16>, grid, dim3(256), 0, s, (const __hip_bfloat16*)x,
                       (const __hip_bfloat16*)y, (const __hip_bfloat16*)dy, a, (__hip_bfloat16*)dx, part, rows, C,
                       slope, relu_in, need_dscale, RG, final_partials);
}

static int affine_bwd_check(const void* x, const void* y, const void* dy, const float* gamma, const float* beta,
                            const float* mean, const float* var, long long rows, int C, float slope, int relu_in,
                            int dtype) {
  XPT_CHECK_PTR(dy); XPT_CHECK_PTR(beta);
  if (slope != 1.f) XPT_CHECK_PTR(y);
  if (rows <= 0 || C <= 0) return XPT_ERR_SHAPE;
  if (dtype != 0 && dtype != 1) return XPT_ERR_ARG;
  if (gamma && (!mean || !var)) return XPT_ERR_NULL;
  if ((gamma || relu_in) && !x) return XPT_ERR_NULL;
  return XPT_OK;
}

/* dx may be NULL (no data gradient wanted); dgamma must be NULL iff gamma is NULL (bias-only epilogue). */
int xpt_affine_act_bwd(const void* x, const void* y, const void* dy, const float* gamma, const float* beta,
                       const float* mean, const float* var, float eps, void* dx, float* dbeta, float* dgamma,
                       float* workspace, size_t workspace_floats, long long rows, int C, float slope, int relu_in,
                       int dtype, void* stream) {
  XPT_CHECK_PTR(dbeta); XPT_CHECK_PTR(workspace);
  const int rc = affine_bwd_check(x, y, dy, gamma, beta, mean, var, rows, C, slope, relu_in, dtype);
  if (rc != XPT_OK) return rc;
  if ((gamma == nullptr) != (dgamma == nullptr)) return XPT_ERR_ARG;
  if (workspace_floats < xpt_affine_act_bwd_workspace_floats(rows, C)) return XPT_ERR_WORKSPACE;
  const Affine a{gamma, beta, mean, var, eps};
  hipStream_t s = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
  affine_bwd_launch(x, y, dy, a, dx, workspace, rows, C, slope, relu_in, dgamma != nullptr, 0, dtype, s);
  hipLaunchKernelGGL(affine_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, s, workspace, a, dbeta, dgamma, C,
                     affine_bwd_blocks(rows, C));
  return xpt_launch_status();
}

int xpt_affine_act_bwd_blocks(long long rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  return affine_bwd_blocks(rows, C);
}

/* Deferred parameter gradients: dx as above; partials[blk][0][c] = this block's share of dbeta[c] and, when gamma is
 * given, partials[blk][1][c] = its share of dgamma[c] -- to be added up later by xpt_reduce_partials. */
int xpt_affine_act_bwd_partials(const void* x, const void* y, const void* dy, const float* gamma, const float* beta,
                                const float* mean, const float* var, float eps, void* dx, float* partials,
                                size_t partial_floats, long long rows, int C, float slope, int relu_in, int dtype,
                                void* stream) {
  XPT_CHECK_PTR(partials);
  const int rc = affine_bwd_check(x, y, dy, gamma, beta, mean, var, rows, C, slope, relu_in, dtype);
  if (rc != XPT_OK) return rc;
  if (partial_floats < (size_t)affine_bwd_blocks(rows, C) * 2 * (size_t)C) return XPT_ERR_WORKSPACE;
  const Affine a{gamma, beta, mean, var, eps};
  XPT_BEGIN_LAUNCH();
  affine_bwd_launch(x, y, dy, a, dx, partials, rows, C, slope, relu_in, gamma != nullptr, 1, dtype, (hipStream_t)stream);
  return xpt_launch_status();
}

}  // extern "C"
