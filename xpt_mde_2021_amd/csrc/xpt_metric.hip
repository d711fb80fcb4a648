// xpt_metric.hip -- the per-step depth metric of the training loop in ONE launch (gfx950).
//
// Replaces get_depth_metric (model/train_val.py:180-200) = valid_depth_filter + median scaling + abs-rel of
// evaluate/eval_utils.py:109-131, which the reference runs per sample on the host with numpy after every step
// (run_an_epoch -> merge_results, train_val.py:43-64, 157-177):
//     mask   = (gt > min_depth) & (gt < max_depth) & crop            (Garg crop rows [r0, r1), columns [c0, c1))
//     ratio  = median(gt[mask]) / median(pred[mask])                 (np.median: mean of the two middle values)
//     scaled = clip(pred * ratio, min_depth, max_depth)
//     absrel = mean(|gt - scaled| / gt over mask)                    (0 when the mask is empty)
// The library formulation needs two full sorts of [B, h w] (72 merge kernels, ~0.5 ms per step).  Here one workgroup of
// 1024 threads owns a sample: the k-th smallest of the masked values is found by RADIX SELECTION on the float bit
// patterns (4 passes of 8 bits; per-wave histograms in LDS, no global atomics), the upper median by one more pass
// (count of values <= v, smallest value above v), then the error sum.  Exact: the selected values are elements of the
// arrays, and every sum is reduced in a fixed order (deterministic).
#include "xpt_common.h"

namespace {

constexpr int NT = 1024, NWAVE = NT / 64;

struct MetricDims {
  int B, h, w, r0, r1, c0, c1;
  float dmin, dmax;
};

__device__ inline unsigned float_key(float f) {       // monotonic map float -> unsigned (negative values first)
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float key_float(unsigned k) {
  const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

__device__ inline bool in_mask(const MetricDims& d, const float* gt, int p) {
  const int y = p / d.w, x = p - y * d.w;
  const float g = gt[p];
  return g > d.dmin && g < d.dmax && y >= d.r0 && y < d.r1 && x >= d.c0 && x < d.c1;
}

// block-wide sums in a fixed order: lanes by shuffle, waves through LDS in wave order
__device__ inline float block_sum_f(float v, float* sred) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < NWAVE; ++w) t += sred[w];
  return t;
}
__device__ inline unsigned block_sum_u(unsigned v, unsigned* sred) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned t = 0;
  for (int w = 0; w < NWAVE; ++w) t += sred[w];
  return t;
}
__device__ inline unsigned block_min_u(unsigned v, unsigned* sred) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, (unsigned)__shfl_down((int)v, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned t = 0xffffffffu;
  for (int w = 0; w < NWAVE; ++w) t = min(t, sred[w]);
  return t;
}

// key of the k-th smallest (0-based) masked value of `val`; every thread returns the same key
__device__ unsigned select_kth(const MetricDims& d, const float* gt, const float* val, int P, unsigned k,
                               unsigned (*hist)[256], unsigned* sscan) {
  unsigned prefix = 0, known = 0;                      // bits decided so far
  const int wave = threadIdx.x >> 6;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = threadIdx.x; i < NWAVE * 256; i += NT) (&hist[0][0])[i] = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += NT) {
      if (!in_mask(d, gt, p)) continue;
      const unsigned key = float_key(val[p]);
      if ((key & known) == prefix) atomicAdd(&hist[wave][(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256) {                           // totals per digit
      unsigned t = 0;
      for (int w = 0; w < NWAVE; ++w) t += hist[w][threadIdx.x];
      sscan[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                            // the digit whose cumulative count passes k
      unsigned acc = 0;
      int digit = 255;
      for (int b = 0; b < 256; ++b) {
        if (acc + sscan[b] > k) { digit = b; break; }
        acc += sscan[b];
      }
      sscan[256] = (unsigned)digit;
      sscan[257] = acc;
    }
    __syncthreads();
    prefix |= sscan[256] << shift;
    known |= 255u << shift;
    k -= sscan[257];
    __syncthreads();
  }
  return prefix;
}

// median of the masked values (np.median semantics) given their count
__device__ float masked_median(const MetricDims& d, const float* gt, const float* val, int P, unsigned cnt,
                               unsigned (*hist)[256], unsigned* sscan, unsigned* ured) {
  const unsigned k_lo = (cnt - 1) / 2, k_hi = cnt / 2;
  const unsigned key_lo = select_kth(d, gt, val, P, k_lo, hist, sscan);
  float lo = key_float(key_lo), hi = lo;
  if (k_hi != k_lo) {                                  // even count: the next order statistic
    unsigned le = 0, above = 0xffffffffu;
    for (int p = threadIdx.x; p < P; p += NT) {
      if (!in_mask(d, gt, p)) continue;
      const unsigned key = float_key(val[p]);
      if (key <= key_lo) ++le; else above = min(above, key);
    }
    le = block_sum_u(le, ured);
    above = block_min_u(above, ured);
    hi = (le >= k_hi + 1) ? lo : key_float(above);
  }
  return 0.5f * (lo + hi);
}

__global__ __launch_bounds__(NT) void depth_metric_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                          float* __restrict__ out, MetricDims d) {
  __shared__ unsigned hist[NWAVE][256];
  __shared__ unsigned sscan[258];
  __shared__ unsigned ured[NWAVE];
  __shared__ float fred[NWAVE];
  const int b = blockIdx.x, P = d.h * d.w;
  const float* g = gt + (long long)b * P;
  const float* q = pred + (long long)b * P;
  unsigned c = 0;
  for (int p = threadIdx.x; p < P; p += NT) c += in_mask(d, g, p) ? 1u : 0u;
  const unsigned cnt = block_sum_u(c, ured);
  if (cnt == 0) {                                      // uniform
    if (threadIdx.x == 0) out[b] = 0.f;
    return;
  }
  const float med_gt = masked_median(d, g, g, P, cnt, hist, sscan, ured);
  const float med_pr = masked_median(d, g, q, P, cnt, hist, sscan, ured);
  const float ratio = med_gt / med_pr;
  float err = 0.f;
  for (int p = threadIdx.x; p < P; p += NT) {
    if (!in_mask(d, g, p)) continue;
    const float s = fminf(fmaxf(q[p] * ratio, d.dmin), d.dmax);
    err += fabsf(g[p] - s) / g[p];
  }
  err = block_sum_f(err, fred);
  if (threadIdx.x == 0) out[b] = err / (float)cnt;
}

}  // namespace

/* per_sample[b] = abs-rel of pred[b] against gt[b] after valid_depth_filter + median scaling (see the file header);
 * pred, gt [B, h, w] fp32 contiguous; crop = rows [r0, r1) x columns [c0, c1). */
extern "C" int xpt_depth_metric(const float* pred, const float* gt, float* per_sample, int B, int h, int w, int r0, int r1,
                                int c0, int c1, float min_depth, float max_depth, void* stream) {
  XPT_CHECK_PTR(pred); XPT_CHECK_PTR(gt); XPT_CHECK_PTR(per_sample);
  if (B <= 0 || h <= 0 || w <= 0 || (long long)h * w >= (1LL << 30)) return XPT_ERR_SHAPE;
  if (r0 < 0 || r1 > h || c0 < 0 || c1 > w || r0 > r1 || c0 > c1 || !(min_depth < max_depth)) return XPT_ERR_ARG;
  const MetricDims d{B, h, w, r0, r1, c0, c1, min_depth, max_depth};
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(depth_metric_kernel, dim3(B), dim3(NT), 0, (hipStream_t)stream, pred, gt, per_sample, d);
  return xpt_launch_status();
}
