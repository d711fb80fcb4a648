// xpt_common.h -- device helpers shared by the gfx950 kernels of libxpt_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/xpt_hip.h"

#define XPT_WAVE 64

// ---- the 16-bit activation format.  The library is built twice from these sources (csrc/build.py): libxpt_hip.so --
// bfloat16, the benchmarked configuration -- and, with -DXPT_HALF_F16, libxpt_hip_f16.so -- IEEE half, BASELINE.json
// configs[4] "fp16 convs + fp32 loss accumulation".  Everything that depends on the format is here: the element types, the
// conversions (fp32 arithmetic everywhere else) and the matrix-core instruction; `dtype == 1` of the C ABI means "the 16-bit
// format of this build".
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#ifdef XPT_HALF_F16
typedef _Float16 xpt_h16;                 // operand element of the matrix-core builtins
typedef __half xpt_half_t;                // element type of the templated element-wise / stencil kernels
#define XPT_MFMA_32X32X16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
typedef __bf16 xpt_h16;
typedef __hip_bfloat16 xpt_half_t;
#define XPT_MFMA_32X32X16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
typedef xpt_h16 xpt_h16x8 __attribute__((ext_vector_type(8)));
#ifdef __HIPCC__
__device__ __forceinline__ float xpt_h2f(unsigned short u) {             // bits -> fp32 (exact)
#ifdef XPT_HALF_F16
  return (float)__builtin_bit_cast(_Float16, u);
#else
  return __uint_as_float(((unsigned)u) << 16);
#endif
}
__device__ __forceinline__ float xpt_h2f_lo(unsigned w) {                // low / high element of a packed pair
#ifdef XPT_HALF_F16
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu));
#else
  return __uint_as_float(w << 16);
#endif
}
__device__ __forceinline__ float xpt_h2f_hi(unsigned w) {
#ifdef XPT_HALF_F16
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
#else
  return __uint_as_float(w & 0xffff0000u);
#endif
}
__device__ __forceinline__ unsigned short xpt_f2h(float f) {             // fp32 -> bits, round to nearest even (the hardware's conversion)
  return __builtin_bit_cast(unsigned short, (xpt_h16)f);
}
__device__ __forceinline__ unsigned short xpt_f2h_sw(float f) {          // the pointwise kernels' integer round-to-nearest-even (bf16)
#ifdef XPT_HALF_F16
  return xpt_f2h(f);
#else
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
#endif
}
__device__ __forceinline__ float xpt_half2float(xpt_half_t h) {
#ifdef XPT_HALF_F16
  return __half2float(h);
#else
  return __bfloat162float(h);
#endif
}
__device__ __forceinline__ xpt_half_t xpt_float2half(float f) {
#ifdef XPT_HALF_F16
  return __float2half(f);
#else
  return __float2bfloat16(f);
#endif
}
#endif

#define XPT_CHECK_PTR(p) \
  do {                   \
    if ((p) == nullptr) return XPT_ERR_NULL; \
  } while (0)

static inline int xpt_launch_status() { return hipGetLastError() == hipSuccess ? XPT_OK : XPT_ERR_LAUNCH; }
// hipGetLastError() is per-thread and shared with the host framework: drop any stale (already handled)
// error of an earlier runtime call before launching so that the status we return is our own.
#define XPT_BEGIN_LAUNCH() (void)hipGetLastError()

// ---- index arithmetic without the integer divider this ISA does not have.
// A 64-bit `%` or `/` by a run-time value expands to ~120 instructions, a 32-bit unsigned one to ~35; the element-wise and
// stencil kernels split a flat index three times per thread, which at batch 8 was a third to a half of their instructions
// (measured on the k x k weight gradient's staging: 5.81 -> 5.74 ms per step from that function alone).
// xpt_divmod: q = n / d, rem = n % d for n < 2^31, 0 < d < 2^24, branch-free: a float estimate (v_rcp_f32, 1 ulp; off by up
// to ~n 2^-22 / d + 2), one refinement on the small remainder it leaves (exact in float), and a +-1 fix-up.  (A variant
// that fell back to the 32-bit division above 2^24 was if-converted by the compiler: both arms executed, slower than
// before.)
__device__ __forceinline__ unsigned xpt_divmod(unsigned n, unsigned d, unsigned& rem) {
  const float inv = __builtin_amdgcn_rcpf((float)d);
  unsigned q = (unsigned)((float)n * inv);
  int r = (int)(n - q * d);
  q += (unsigned)(int)((float)r * inv);
  r = (int)(n - q * d);
  if (r < 0) { --q; r += (int)d; }
  if (r < 0) { --q; r += (int)d; }
  if (r >= (int)d) { ++q; r -= (int)d; }
  if (r >= (int)d) { ++q; r -= (int)d; }
  rem = (unsigned)r;
  return q;
}
// idx = ((i3 * n2 + i2) * n1 + i1) * n0 + i0  ->  (i0, i1, i2, i3)
__device__ __forceinline__ void xpt_split4(unsigned idx, int n0, int n1, int n2, int& i0, int& i1, int& i2, int& i3) {
  unsigned r0, r1, r2;
  unsigned q = xpt_divmod(idx, (unsigned)n0, r0);
  q = xpt_divmod(q, (unsigned)n1, r1);
  q = xpt_divmod(q, (unsigned)n2, r2);
  i0 = (int)r0; i1 = (int)r1; i2 = (int)r2; i3 = (int)q;
}

namespace xpt {

// ---------------------------------------------------------------- camera (scaled intrinsic + inverse)
// scale_intrinsic (synthesize_base.py:66-71): rows 0,1 of K divided by `scale`, row 2 = (0,0,1);
// pixel2cam (synthesize_base.py:137) inverts the full 3x3 (no zero-skew assumption).
struct Cam {
  float k[9];
  float ki[9];
};

__device__ inline Cam load_cam(const float* __restrict__ K, float scale) {
  Cam c;
#pragma unroll
  for (int i = 0; i < 6; ++i) c.k[i] = K[i] / scale;
  c.k[6] = 0.f;
  c.k[7] = 0.f;
  c.k[8] = 1.f;
  const float a = c.k[0], b = c.k[1], cc = c.k[2], d = c.k[3], e = c.k[4], f = c.k[5], g = c.k[6], h = c.k[7],
              i = c.k[8];
  const float A = e * i - f * h, Bc = -(d * i - f * g), Cc = d * h - e * g;
  const float det = a * A + b * Bc + cc * Cc;
  const float r = 1.0f / det;
  c.ki[0] = A * r;
  c.ki[1] = -(b * i - cc * h) * r;
  c.ki[2] = (b * f - cc * e) * r;
  c.ki[3] = Bc * r;
  c.ki[4] = (a * i - cc * g) * r;
  c.ki[5] = -(a * f - cc * d) * r;
  c.ki[6] = Cc * r;
  c.ki[7] = -(a * h - b * g) * r;
  c.ki[8] = (a * e - b * d) * r;
  return c;
}

// rows 0..2 of a 4x4 target->source matrix
struct Pose {
  float r[9];
  float t[3];
};

__device__ inline Pose load_pose(const float* __restrict__ T) {
  Pose p;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    p.r[3 * i + 0] = T[4 * i + 0];
    p.r[3 * i + 1] = T[4 * i + 1];
    p.r[3 * i + 2] = T[4 * i + 2];
    p.t[i] = T[4 * i + 3];
  }
  return p;
}

// ---------------------------------------------------------------- projective warp of one target pixel
// warp_pixel_coords (synthesize_base.py:106-178):
//   X = d * Kinv (u,v,1)^T ; X' = R X + t ; p = K X' ; (u',v') = p_xy / (p_z + 1e-10)
struct Warp {
  float X[3];    // target-frame point
  float ray[3];  // Kinv (u,v,1)
  float up, vp;  // projected source pixel
  float zinv;    // 1 / (p_z + 1e-10)
};

__device__ inline void backproject(const Cam& c, float u, float v, float d, Warp& w) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    w.ray[i] = c.ki[3 * i + 0] * u + c.ki[3 * i + 1] * v + c.ki[3 * i + 2];
    w.X[i] = w.ray[i] * d;
  }
}

__device__ inline void project(const Cam& c, const Pose& p, Warp& w) {
  float Xs[3], q[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) Xs[i] = p.r[3 * i + 0] * w.X[0] + p.r[3 * i + 1] * w.X[1] + p.r[3 * i + 2] * w.X[2] + p.t[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) q[i] = c.k[3 * i + 0] * Xs[0] + c.k[3 * i + 1] * Xs[1] + c.k[3 * i + 2] * Xs[2];
  const float z = q[2] + 1e-10f;
  w.zinv = 1.0f / z;
  w.up = q[0] / z;
  w.vp = q[1] / z;
}

// ---------------------------------------------------------------- bilinear neighbourhood
// BilinearInterpolation (bilinear_interp.py:34-102): clipped floor / floor+1, validity =
// (uf+1 == uc) & (vf+1 == vc) [& depth != 0], weights (uc-u)(vc-v) ... times the mask.
struct Taps {
  int uf, uc, vf, vc;  // clipped integer neighbours
  float wuf, wuc, wvf, wvc;
  float mask;  // 1 / 0
};

__device__ inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

__device__ inline Taps make_taps(float u, float v, int h, int w, bool extra_valid) {
  Taps t;
  const float fu = floorf(u), fv = floorf(v);
  const float ufc = clampf(fu, 0.f, (float)(w - 1));
  const float ucc = clampf(fu + 1.f, 0.f, (float)(w - 1));
  const float vfc = clampf(fv, 0.f, (float)(h - 1));
  const float vcc = clampf(fv + 1.f, 0.f, (float)(h - 1));
  // NaN coordinates compare false -> invalid (the reference would emit NaN; documented deviation)
  const bool ok = (ufc + 1.f == ucc) && (vfc + 1.f == vcc) && extra_valid;
  t.mask = ok ? 1.f : 0.f;
  t.uf = ok ? (int)ufc : 0;
  t.uc = ok ? (int)ucc : 0;
  t.vf = ok ? (int)vfc : 0;
  t.vc = ok ? (int)vcc : 0;
  t.wuf = ok ? (ucc - u) : 0.f;
  t.wuc = ok ? (u - ufc) : 0.f;
  t.wvf = ok ? (vcc - v) : 0.f;
  t.wvc = ok ? (v - vfc) : 0.f;
  return t;
}

// ---------------------------------------------------------------- wave / block reductions
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

// sum over a block of up to 1024 threads (blockDim.x multiple of 64); result valid in thread 0.
// `red` = shared float[16].  Deterministic order.
__device__ inline float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();  // protect `red` reuse between consecutive calls
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0) {
    for (int i = 0; i < nw; ++i) s += red[i];
  }
  return s;
}

// sums NV per-thread values over a block of 256 threads (4 waves); results valid in thread 0.
// `red` = shared float[4 * NV].  Deterministic order.  Every thread of the block must call it.
template <int NV>
__device__ inline void block_sum_n(float (&v)[NV], float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[wid * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float s = 0.f;
      for (int k = 0; k < nw; ++k) s += red[k * NV + i];
      v[i] = s;
    }
  }
}

}  // namespace xpt

// ---- image-to-XCD affinity.  Workgroup i of a launch runs on XCD i % 8, and the L2 of an XCD keeps what its CUs wrote across
// kernel boundaries (tools/lab/xcd_handoff.hip: a consumer kernel that reads on the XCD that wrote is up to 2x faster than one
// that reads on another; the per-XCD L2s are not coherent with each other, so a remote reader goes to the Infinity Cache).
// A kernel whose logical units are numbered image-major maps workgroup (bx, by) of a (gx, gy) grid to the unit
//   u = (lin % 8) * (total / 8) + lin / 8,   lin = bx + gx * by,
// so that units [k total / 8, (k + 1) total / 8) -- image k of a batch of 8 -- run on XCD k in EVERY kernel that follows
// the convention.  Only when total % 8 == 0 (the launcher checks; otherwise the identity).  Pure renumbering: same results.
extern int g_xpt_xcd_affinity;            // xpt_set_xcd_affinity(); defined in xpt_host.hip
#ifdef __HIPCC__
__device__ __forceinline__ void xpt_xcd_remap(bool on, unsigned& bx, unsigned& by, unsigned gx, unsigned gy) {
  if (!on) return;
  const unsigned lin = bx + gx * by, total = gx * gy;
  const unsigned u = (lin & 7u) * (total >> 3) + (lin >> 3);
  by = u / gx;
  bx = u - by * gx;
}
#endif
inline bool xpt_xcd_ok(unsigned long long total) { return g_xpt_xcd_affinity != 0 && total % 8 == 0 && total >= 64; }

// The same affinity for the row-sliced kernels of the encoder (pointwise / depthwise / cell kernels: activations are [M = B H W, C]
// matrices).  A launch numbers its n units -- contiguous ranges of pixel rows, in row order -- along the FASTEST grid dimension,
// padded to xpt_xcd_pad(n) workgroups; workgroup x of that dimension runs on XCD x % 8 (every other dimension's stride is a
// multiple of 8) and takes unit xpt_xcd_unit(): XCD k gets the units [k n / 8, (k + 1) n / 8), i.e. the rows [k M / 8, (k + 1) M / 8)
// -- image k of a batch of 8 -- in EVERY kernel of the chain, so what a layer wrote is still in the L2 its reader looks in.
// Workgroups past their XCD's share return at once.  Pure renumbering: same results with the affinity off (unit = x).
inline unsigned xpt_xcd_pad(unsigned long long n) { return (unsigned)((n + 7) & ~7ull); }
// n / d for a divisor the host knows: m = xpt_magic(d), q = xpt_fastdiv(n, m); exact for n * d < 2^32 (workgroup decodes)
inline unsigned xpt_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }
#ifdef __HIPCC__
__device__ __forceinline__ unsigned xpt_fastdiv(unsigned n, unsigned m) { return m ? __umulhi(n, m) : n; }
#endif

// flat sweeps (element-wise kernels over pixel-major indices): workgroup x of the fastest grid dimension takes the indices
// [u per, (u + 1) per) of its unit u instead of a grid-stride walk
struct XcdSweep {
  unsigned grid, units, per;      // workgroups to launch along x, units, consecutive indices per unit (a multiple of 256)
  int on;
#ifdef __HIPCC__
  __device__ __forceinline__ bool range(unsigned x, long long total, long long& begin, long long& end) const;
#endif
};
inline XcdSweep xpt_xcd_sweep(long long total, long long max_units, int threads = 256) {
  XcdSweep s;
  long long units = (total + threads - 1) / threads;
  if (units > max_units) units = max_units;
  if (units < 1) units = 1;
  s.per = (unsigned)(((total + units - 1) / units + threads - 1) / threads * threads);
  s.units = (unsigned)((total + s.per - 1) / s.per);
  if (s.units < 1) s.units = 1;
  s.on = g_xpt_xcd_affinity;
  s.grid = s.on ? xpt_xcd_pad(s.units) : s.units;
  return s;
}
#ifdef __HIPCC__
__device__ __forceinline__ bool xpt_xcd_unit(bool on, unsigned x, unsigned n, unsigned& u) {
  if (!on) {
    u = x;
    return x < n;
  }
  const unsigned k = x & 7u, q = x >> 3;
  const unsigned lo = (k * n) >> 3, hi = ((k + 1u) * n) >> 3;
  u = lo + q;
  return u < hi;
}
// A 1-D grid of two classes of row-ordered segments (the depthwise backward launches: nA data-gradient segments of LA
// workgroups, then nB weight-gradient segments of LB): the host launches nA * pad(LA) + nB * pad(LB) workgroups, workgroup x
// gets the block number vb of the plain layout [segment][block] it stands for (XCD k: the k-th eighth of every segment).
inline unsigned xpt_xcd_two_class_grid(int on, unsigned nA, unsigned LA, unsigned nB, unsigned LB) {
  return on ? nA * xpt_xcd_pad(LA) + nB * xpt_xcd_pad(LB) : nA * LA + nB * LB;
}
__device__ __forceinline__ bool xpt_xcd_two_class(bool on, unsigned x, unsigned nA, unsigned LA, unsigned nB, unsigned LB,
                                                  unsigned& vb) {
  if (!on) {
    vb = x;
    return true;
  }
  const unsigned a8 = nA * ((LA + 7u) & ~7u);
  const bool first = x < a8;
  const unsigned L = first ? LA : LB, lin = first ? x : x - a8;
  const unsigned c8 = (L + 7u) >> 3, k = lin & 7u, q = lin >> 3;
  unsigned local;
  const unsigned seg = xpt_divmod(q, c8, local);
  const unsigned lo = (k * L) >> 3, hi = ((k + 1u) * L) >> 3;
  vb = (first ? 0u : nA * LA) + seg * L + lo + local;
  return lo + local < hi;
}
// the consecutive indices [begin, end) of block blk when `total` indices are dealt to nblocks blocks in multiples of 256
__device__ __forceinline__ void xpt_chunk_range(unsigned blk, unsigned nblocks, long long total, long long& begin, long long& end) {
  unsigned rem_;                                                 // (total < 2^31: the callers split these indices in 32 bits)
  const unsigned per = (xpt_divmod((unsigned)total + nblocks - 1u, nblocks, rem_) + 255u) & ~255u;
  begin = (long long)blk * per;
  end = begin + per < total ? begin + per : total;
}
__device__ __forceinline__ bool XcdSweep::range(unsigned x, long long total, long long& begin, long long& end) const {
  unsigned u;
  if (!xpt_xcd_unit(on != 0, x, units, u)) return false;
  begin = (long long)u * per;
  end = begin + per < total ? begin + per : total;
  return true;
}
#endif
