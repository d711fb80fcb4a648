// xpt_corr.hip -- correlation cost volume of PWC-Net (flow_net.py:181-196: tfa.layers.CorrelationCost with
// kernel_size 1, stride_1 1, pad = max_displacement, channels_last) and its gradients, NHWC feature maps.
//
//   out[b, y, x, ty * D + tx] = (1 / C) * sum_c left[b, y, x, c] * right[b, y + (ty - rad) * s2, x + (tx - rad) * s2, c]
//   rad = max_displacement / s2,  D = 2 rad + 1,  zero outside the image (the layer pads both maps with zeros)
//
// Forward: one thread per output element.  A workgroup takes PX = floor(256 / D^2) consecutive pixels (3 for the 81
// displacements of levels 2..5, 10 for the 25 of level 6), stages their left vectors in LDS as fp32 (read back as
// broadcasts) and every thread walks the C channels of ITS displaced right pixel with 8/16-byte loads; the D^2 results
// of a pixel are consecutive in memory, so the stores of a workgroup are one contiguous run.
// Backward: one thread per (pixel, 4 channels) of dleft or dright (blockIdx.y), lanes along the channels (coalesced rows,
// 8 / 16-byte loads), walking the D^2 displacements; both are gathers (dright reads the pixels that looked at it), no
// atomics.
#include "xpt_common.h"


namespace {

template <typename T> __device__ inline float cv_ld(const T* p);
template <> __device__ inline float cv_ld<float>(const float* p) { return *p; }
template <> __device__ inline float cv_ld<xpt_half_t>(const xpt_half_t* p) { return xpt_half2float(*p); }
template <typename T> __device__ inline void cv_st(T* p, float v);
template <> __device__ inline void cv_st<float>(float* p, float v) { *p = v; }
template <> __device__ inline void cv_st<xpt_half_t>(xpt_half_t* p, float v) { *p = xpt_float2half(v); }

// 4 consecutive channels as floats (C % 4 == 0 and 4-element aligned rows are checked on the host for V == 4)
template <typename T> __device__ inline void cv_ld4(const T* p, float (&v)[4]);
template <> __device__ inline void cv_ld4<float>(const float* p, float (&v)[4]) {
  const float4 q = *reinterpret_cast<const float4*>(p);
  v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
template <> __device__ inline void cv_ld4<xpt_half_t>(const xpt_half_t* p, float (&v)[4]) {
  const uint2 q = *reinterpret_cast<const uint2*>(p);
  v[0] = xpt_h2f_lo(q.x); v[1] = xpt_h2f_hi(q.x);
  v[2] = xpt_h2f_lo(q.y); v[3] = xpt_h2f_hi(q.y);
}

// 8 consecutive bf16 channels (16-byte load / store)
__device__ inline void cv_ld8(const xpt_half_t* p, float (&v)[8]) {
  const uint4 q = *reinterpret_cast<const uint4*>(p);
  const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = xpt_h2f_lo(w[i]); v[2 * i + 1] = xpt_h2f_hi(w[i]); }
}
__device__ inline void cv_ld8(const float* p, float (&v)[8]) {          // (not used: fp32 rows go 4 wide)
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = p[i];
}

struct CorrDims {
  int B, H, W, C, rad, s2, D, DD, PX;
  long long npix;
};

template <typename T, int V>
__global__ __launch_bounds__(256) void corr_fwd_kernel(const T* __restrict__ left, const T* __restrict__ right,
                                                       T* __restrict__ out, CorrDims d) {
  extern __shared__ float lvec[];                         // [PX][C]
  const long long p0 = (long long)blockIdx.x * d.PX;
  const int npx = (int)min((long long)d.PX, d.npix - p0);
  for (int i = threadIdx.x; i < npx * d.C; i += 256) lvec[i] = cv_ld(left + p0 * d.C + i);
  __syncthreads();
  const int q = threadIdx.x / d.DD, t = threadIdx.x - q * d.DD;
  if (q >= npx) return;
  const long long p = p0 + q;
  const int x = (int)(p % d.W), y = (int)((p / d.W) % d.H);
  const int ty = t / d.D, tx = t - ty * d.D;
  const int y2 = y + (ty - d.rad) * d.s2, x2 = x + (tx - d.rad) * d.s2;
  const bool ok = y2 >= 0 && y2 < d.H && x2 >= 0 && x2 < d.W;
  // out-of-image displacement: read the own pixel (always valid), zero by select -- no guarded loads
  const long long p2 = ok ? p + (long long)(y2 - y) * d.W + (x2 - x) : p;
  const T* r = right + p2 * d.C;
  const float* l = lvec + q * d.C;
  float acc = 0.f;
  if constexpr (V == 8) {
    for (int c = 0; c < d.C; c += 8) {
      float v[8];
      cv_ld8(r + c, v);
      acc += ((l[c] * v[0] + l[c + 1] * v[1]) + (l[c + 2] * v[2] + l[c + 3] * v[3])) +
             ((l[c + 4] * v[4] + l[c + 5] * v[5]) + (l[c + 6] * v[6] + l[c + 7] * v[7]));
    }
  } else if constexpr (V == 4) {
    for (int c = 0; c < d.C; c += 4) {
      float v[4];
      cv_ld4(r + c, v);
      acc += (l[c] * v[0] + l[c + 1] * v[1]) + (l[c + 2] * v[2] + l[c + 3] * v[3]);
    }
  } else {
    for (int c = 0; c < d.C; ++c) acc += l[c] * cv_ld(r + c);
  }
  cv_st(out + p * d.DD + t, ok ? acc / (float)d.C : 0.f);
}

template <typename T> __device__ inline void cv_st4(T* p, const float (&v)[4]);
template <> __device__ inline void cv_st4<float>(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ inline void cv_st4<xpt_half_t>(xpt_half_t* p, const float (&v)[4]) {
  xpt_half_t h[4] = {xpt_float2half(v[0]), xpt_float2half(v[1]), xpt_float2half(v[2]), xpt_float2half(v[3])};
  *reinterpret_cast<uint2*>(p) = *reinterpret_cast<const uint2*>(h);
}

// side 0: dleft[p, c]  = (1/C) sum_t g[p, t] * right[p + disp(t), c]
// side 1: dright[p, c] = (1/C) sum_t g[p - disp(t), t] * left[p - disp(t), c]
// One thread per (pixel, V consecutive channels): V = 4 reads the other map with 8 / 16-byte loads (2-byte loads are
// address-path-bound on gfx950: one texture-addresser slot per wave load whatever its width).
template <typename T, int V>
__global__ __launch_bounds__(256) void corr_bwd_kernel(const T* __restrict__ left, const T* __restrict__ right,
                                                       const T* __restrict__ gout, T* __restrict__ dleft,
                                                       T* __restrict__ dright, CorrDims d, int pix_per_block) {
  const int side = blockIdx.y;
  const int cpt = d.C / V;                                               // threads per pixel (<= 256)
  const int q = threadIdx.x / cpt, c = (threadIdx.x - q * cpt) * V;
  const long long p = (long long)blockIdx.x * pix_per_block + q;
  if (q >= pix_per_block || p >= d.npix) return;
  const int x = (int)(p % d.W), y = (int)((p / d.W) % d.H);
  const T* other = side == 0 ? right : left;
  const int sign = side == 0 ? 1 : -1;
  float acc[V];
#pragma unroll
  for (int i = 0; i < V; ++i) acc[i] = 0.f;
  for (int ty = 0; ty < d.D; ++ty) {
    const int y2 = y + sign * (ty - d.rad) * d.s2;
    const bool oky = y2 >= 0 && y2 < d.H;
    for (int tx = 0; tx < d.D; ++tx) {
      const int x2 = x + sign * (tx - d.rad) * d.s2;
      const bool ok = oky && x2 >= 0 && x2 < d.W;
      const long long p2 = ok ? p + (long long)(y2 - y) * d.W + (x2 - x) : p;
      const long long pg = side == 0 ? p : p2;                          // the pixel whose cost volume holds the term
      const float g = ok ? cv_ld(gout + pg * d.DD + ty * d.D + tx) : 0.f;   // address always valid: select, no branch
      if constexpr (V == 8) {
        float v[8];
        cv_ld8(other + p2 * d.C + c, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += g * v[i];
      } else if constexpr (V == 4) {
        float v[4];
        cv_ld4(other + p2 * d.C + c, v);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += g * v[i];
      } else {
        acc[0] += g * cv_ld(other + p2 * d.C + c);
      }
    }
  }
  const float inv = 1.0f / (float)d.C;
  T* dst = (side == 0 ? dleft : dright) + p * d.C + c;
  if constexpr (V == 8) {
    float lo[4] = {acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
    float hi[4] = {acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv};
    cv_st4(dst, lo);
    cv_st4(dst + 4, hi);
  } else if constexpr (V == 4) {
    float o[4] = {acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
    cv_st4(dst, o);
  } else {
    cv_st(dst, acc[0] * inv);
  }
}

int make_dims(int B, int H, int W, int C, int max_disp, int stride2, CorrDims* d) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || max_disp < 0 || stride2 <= 0) return XPT_ERR_SHAPE;
  d->B = B; d->H = H; d->W = W; d->C = C;
  d->s2 = stride2;
  d->rad = max_disp / stride2;
  d->D = 2 * d->rad + 1;
  d->DD = d->D * d->D;
  if (d->DD > 256) return XPT_ERR_SHAPE;                  // PWC-Net: 81 (levels 2..5) or 25 (level 6)
  d->PX = 256 / d->DD;
  d->npix = (long long)B * H * W;
  return XPT_OK;
}

}  // namespace

extern "C" {

int xpt_corr_cost_channels(int max_disp, int stride2) {
  if (max_disp < 0 || stride2 <= 0) return XPT_ERR_ARG;
  const int D = 2 * (max_disp / stride2) + 1;
  return D * D;
}

int xpt_corr_cost_fwd(const void* left, const void* right, void* out, int B, int H, int W, int C, int max_disp,
                      int stride2, int dtype, void* stream) {
  XPT_CHECK_PTR(left); XPT_CHECK_PTR(right); XPT_CHECK_PTR(out);
  if (dtype != 0 && dtype != 1) return XPT_ERR_ARG;
  CorrDims d;
  const int rc = make_dims(B, H, W, C, max_disp, stride2, &d);
  if (rc != XPT_OK) return rc;
  const size_t lds = (size_t)d.PX * C * sizeof(float);
  if (lds > 64 * 1024) return XPT_ERR_SHAPE;
  const unsigned blocks = (unsigned)((d.npix + d.PX - 1) / d.PX);
  const int esz = dtype == 0 ? 4 : 2;
  const bool vec = C % 4 == 0 && ((uintptr_t)right) % (size_t)(4 * esz) == 0;
  const bool vec8 = dtype == 1 && C % 8 == 0 && ((uintptr_t)right) % 16 == 0;
  hipStream_t s = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
#define XPT_CORR(T, V) \
  hipLaunchKernelGGL((corr_fwd_kernel<T, V>), dim3(blocks), dim3(256), lds, s, (const T*)left, (const T*)right, (T*)out, d)
  if (dtype == 0) {
    if (vec) XPT_CORR(float, 4); else XPT_CORR(float, 1);
  } else {
    if (vec8) XPT_CORR(xpt_half_t, 8); else if (vec) XPT_CORR(xpt_half_t, 4); else XPT_CORR(xpt_half_t, 1);
  }
#undef XPT_CORR
  return xpt_launch_status();
}

int xpt_corr_cost_bwd(const void* left, const void* right, const void* gout, void* dleft, void* dright, int B, int H,
                      int W, int C, int max_disp, int stride2, int dtype, void* stream) {
  XPT_CHECK_PTR(left); XPT_CHECK_PTR(right); XPT_CHECK_PTR(gout); XPT_CHECK_PTR(dleft); XPT_CHECK_PTR(dright);
  if (dtype != 0 && dtype != 1) return XPT_ERR_ARG;
  CorrDims d;
  const int rc = make_dims(B, H, W, C, max_disp, stride2, &d);
  if (rc != XPT_OK) return rc;
  const int esz = dtype == 0 ? 4 : 2;
  const bool vec = C % 4 == 0 && ((uintptr_t)left) % (size_t)(4 * esz) == 0 && ((uintptr_t)right) % (size_t)(4 * esz) == 0 &&
                   ((uintptr_t)dleft) % (size_t)(4 * esz) == 0 && ((uintptr_t)dright) % (size_t)(4 * esz) == 0;
  const bool vec8 = vec && dtype == 1 && C % 8 == 0 && ((uintptr_t)left) % 16 == 0 && ((uintptr_t)right) % 16 == 0 &&
                    ((uintptr_t)dleft) % 8 == 0 && ((uintptr_t)dright) % 8 == 0;
  const int cpt = vec8 ? C / 8 : vec ? C / 4 : C;
  if (cpt > 256) return XPT_ERR_SHAPE;                    // PWC-Net's widest pyramid level has 196 channels
  const int ppb = 256 / cpt;
  const unsigned blocks = (unsigned)((d.npix + ppb - 1) / ppb);
  hipStream_t s = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
#define XPT_CORR_BWD(T, V)                                                                                              \
  hipLaunchKernelGGL((corr_bwd_kernel<T, V>), dim3(blocks, 2), dim3(256), 0, s, (const T*)left, (const T*)right,        \
                     (const T*)gout, (T*)dleft, (T*)dright, d, ppb)
  if (dtype == 0) {
    if (vec) XPT_CORR_BWD(float, 4); else XPT_CORR_BWD(float, 1);
  } else {
    if (vec8) XPT_CORR_BWD(xpt_half_t, 8); else if (vec) XPT_CORR_BWD(xpt_half_t, 4); else XPT_CORR_BWD(xpt_half_t, 1);
  }
#undef XPT_CORR_BWD
  return xpt_launch_status();
}

}  // extern "C"
