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

// ------------------------------------------------------------------------------------------------ pose metrics
// get_pose_metric (model/train_val.py:203-210) = PoseMetricNumpy (evaluate/eval_utils.py:15-87) for the whole batch in one
// launch: the snippet [P0, P1, I (target), P2, ...] re-based on its first frame (inv(M0) M_j in closed form for rigid
// transforms), then per frame j >= 1 the absolute-scale trajectory error |t_true - t_pred|, the scale-aligned one
// |t_true - t_pred (t_true . t_pred) / (t_pred . t_pred)| and the rotation angle acos((tr(R_pred^T R_true) - 1) / 2);
// out = their means over batch x frames.  ~60 tiny launches of matrix slicing / bmm / norm before.
struct Rigid {
  float R[9], t[3];
};

__device__ inline Rigid load_frame(const float* __restrict__ mats, int b, int N, int f) {
  // frame f of [poses[:2], identity, poses[2:]]
  Rigid m;
  const int split = N < 2 ? N : 2;
  if (f == split) {
#pragma unroll
    for (int i = 0; i < 9; ++i) m.R[i] = (i % 4 == 0) ? 1.f : 0.f;
    m.t[0] = m.t[1] = m.t[2] = 0.f;
    return m;
  }
  const float* p = mats + ((long long)b * N + (f < split ? f : f - 1)) * 16;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) m.R[3 * i + j] = p[4 * i + j];
    m.t[i] = p[4 * i + 3];
  }
  return m;
}

// inv(a) * c for rigid transforms: R = Ra^T Rc, t = Ra^T (tc - ta)
__device__ inline Rigid relative_to(const Rigid& a, const Rigid& c) {
  Rigid o;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) o.R[3 * i + j] = a.R[i] * c.R[j] + a.R[3 + i] * c.R[3 + j] + a.R[6 + i] * c.R[6 + j];
    o.t[i] = a.R[i] * (c.t[0] - a.t[0]) + a.R[3 + i] * (c.t[1] - a.t[1]) + a.R[6 + i] * (c.t[2] - a.t[2]);
  }
  return o;
}

__global__ __launch_bounds__(256) void pose_metric_kernel(const float* __restrict__ pred, const float* __restrict__ truth,
                                                          float* __restrict__ out, int B, int N) {
  __shared__ float red[3][256];
  float s_abs = 0.f, s_rel = 0.f, s_rot = 0.f;
  const int items = B * N;                                  // frames 1 .. N of every snippet
  for (int it = threadIdx.x; it < items; it += 256) {
    const int b = it / N, f = it % N + 1;
    const Rigid p = relative_to(load_frame(pred, b, N, 0), load_frame(pred, b, N, f));
    const Rigid q = relative_to(load_frame(truth, b, N, 0), load_frame(truth, b, N, f));
    const float d0 = q.t[0] - p.t[0], d1 = q.t[1] - p.t[1], d2 = q.t[2] - p.t[2];
    s_abs += sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    const float sc = (q.t[0] * p.t[0] + q.t[1] * p.t[1] + q.t[2] * p.t[2]) / (p.t[0] * p.t[0] + p.t[1] * p.t[1] + p.t[2] * p.t[2]);
    const float e0 = q.t[0] - p.t[0] * sc, e1 = q.t[1] - p.t[1] * sc, e2 = q.t[2] - p.t[2] * sc;
    s_rel += sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
    float tr = 0.f;                                         // trace(Rp^T Rq) = sum_ij Rp[i][j] Rq[i][j]
#pragma unroll
    for (int i = 0; i < 9; ++i) tr += p.R[i] * q.R[i];
    s_rot += acosf(fminf(fmaxf((tr - 1.f) * 0.5f, -1.f), 1.f));
  }
  red[0][threadIdx.x] = s_abs;
  red[1][threadIdx.x] = s_rel;
  red[2][threadIdx.x] = s_rot;
  __syncthreads();
  if (threadIdx.x < 3) {                                    // fixed order: deterministic
    float t = 0.f;
    for (int i = 0; i < 256; ++i) t += red[threadIdx.x][i];
    out[threadIdx.x] = t / (float)items;
  }
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

/* out[0..2] = mean absolute-scale trajectory error, mean scale-aligned trajectory error, mean rotation error (radians) of
 * the predicted pose matrices pred [B, N, 4, 4] against truth [B, N, 4, 4] (both target -> source transforms; the snippet is
 * re-based on its first frame with the identity target pose inserted after the second source). */
extern "C" int xpt_pose_metric(const float* pred, const float* truth, float* out, int B, int N, void* stream) {
  XPT_CHECK_PTR(pred); XPT_CHECK_PTR(truth); XPT_CHECK_PTR(out);
  if (B <= 0 || N <= 0) return XPT_ERR_SHAPE;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(pose_metric_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, truth, out, B, N);
  return xpt_launch_status();
}
