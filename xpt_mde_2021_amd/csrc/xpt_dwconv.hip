// xpt_dwconv.hip -- depthwise k x k convolution (forward, data gradient, weight gradient) on NHWC
// activations for gfx950.  This is the memory-bound half of every keras SeparableConv2D of the
// NASNet-A-Mobile encoder behind DepthNetPretrained (reference call site:
// model/build_model/pretrained_nets.py:36-44 -> tf.keras.applications.NASNetMobile; 10 separable
// convolutions per cell, ~150 per image).  MIOpen routes the bf16 NHWC depthwise weight gradient to
// generic grouped-conv / naive kernels (0.5 - 4.5 ms each, 68 % of the training step in
// profiles/r01_a_*); these kernels are plain streaming stencils instead:
//   * lanes run along the channel axis (innermost in NHWC) -> every load / store is a contiguous segment;
//   * each thread keeps OXT neighbouring outputs of one channel in registers and slides the k-wide row window,
//     so an input row segment is loaded once per ky instead of once per (ky,kx);
//   * the ReLU that always precedes the convolution in _separable_conv_block is fused into the load
//     (and its mask into the data gradient), the zero padding of the stride-2 blocks into the bounds test;
//   * weights / weight gradients stay fp32; activations are fp32 or bf16 with fp32 accumulation;
//   * the weight gradient is reduced deterministically: per-workgroup partials + a fixed-order sum.

#include <initializer_list>

#include "xpt_common.h"

namespace {

template <typename T> __device__ inline float ldf(const T* p);
template <> __device__ inline float ldf<float>(const float* p) { return *p; }
template <> __device__ inline float ldf<xpt_half_t>(const xpt_half_t* p) { return xpt_half2float(*p); }
template <typename T> __device__ inline void stf(T* p, float v);
template <> __device__ inline void stf<float>(float* p, float v) { *p = v; }
template <> __device__ inline void stf<xpt_half_t>(xpt_half_t* p, float v) { *p = xpt_float2half(v); }

struct DwDims {
  int B, H, W, C, OH, OW, pad_t, pad_l;
};

// ---------------------------------------------------------------- forward
// y[b,oy,ox,c] = sum_{ky,kx} f(x[b, oy*S+ky-pad_t, ox*S+kx-pad_l, c]) * w[c,ky,kx],  f = relu or identity.
template <typename T, int K, int S, int OXT>
__device__ inline void dw_fwd_body(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y,
                                   const DwDims& d, int relu_in, const XcdSweep& sw, unsigned block) {
  const int OXG = (d.OW + OXT - 1) / OXT;
  const long long total = (long long)d.B * d.OH * OXG * d.C;
  constexpr int IN = (OXT - 1) * S + K;
  long long idx, end;
  if (!sw.range(block, total, idx, end)) return;      // (pixel-major indices: image-to-XCD numbering, xpt_common.h)
  for (idx += threadIdx.x; idx < end; idx += 256) {
    int c, oxg, oy, b;
    xpt_split4((unsigned)idx, d.C, OXG, d.OH, c, oxg, oy, b);
    const int ox0 = oxg * OXT;
    const int ix0 = ox0 * S - d.pad_l;
    float acc[OXT];
#pragma unroll
    for (int i = 0; i < OXT; ++i) acc[i] = 0.f;
    const float* wc = w + (long long)c * K * K;
    // Loads are unconditional on clamped coordinates and zeroed by a select: a guarded load compiles to a branch with
    // its own s_waitcnt, which serialises the ~K*IN loads of a thread into as many memory round trips.
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int iy = oy * S + ky - d.pad_t;
      const bool row_ok = iy >= 0 && iy < d.H;
      const T* row = x + (((long long)b * d.H + min(max(iy, 0), d.H - 1)) * d.W) * d.C + c;
      float in[IN];
#pragma unroll
      for (int i = 0; i < IN; ++i) {
        const int ix = ix0 + i;
        const float v = ldf<T>(row + (long long)min(max(ix, 0), d.W - 1) * d.C);
        in[i] = (row_ok && ix >= 0 && ix < d.W) ? (relu_in ? fmaxf(v, 0.f) : v) : 0.f;
      }
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const float wv = wc[ky * K + kx];
#pragma unroll
        for (int i = 0; i < OXT; ++i) acc[i] += in[i * S + kx] * wv;
      }
    }
    T* out = y + (((long long)b * d.OH + oy) * d.OW + ox0) * d.C + c;
#pragma unroll
    for (int i = 0; i < OXT; ++i)
      if (ox0 + i < d.OW) stf<T>(out + (long long)i * d.C, acc[i]);
  }
}

template <typename T, int K, int S, int OXT>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, DwDims d,
                              int relu_in, XcdSweep sw) {
  dw_fwd_body<T, K, S, OXT>(x, w, y, d, relu_in, sw, blockIdx.x);
}

// Several stride-1 depthwise layers of one shape in ONE launch (the five branch convolutions of a NASNet normal cell
// read only two tensors and are mutually independent): job = blockIdx.x / blocks_per_job, kernel size per job.
#define DW_MAX_JOBS 6
struct DwMultiFwd {
  const void* x[DW_MAX_JOBS];
  const float* w[DW_MAX_JOBS];
  void* y[DW_MAX_JOBS];
  int k[DW_MAX_JOBS], pad_t[DW_MAX_JOBS], pad_l[DW_MAX_JOBS];
};

template <typename T, int S>
__global__ __launch_bounds__(256) void dw_multi_fwd_kernel(DwMultiFwd m, DwDims d, int relu_in, XcdSweep sw) {
  constexpr int OXT = (S == 1) ? 4 : 2;
  const int job = blockIdx.y;
  const unsigned blk = blockIdx.x;
  const int k = m.k[job];
  DwDims dj = d;
  dj.pad_t = m.pad_t[job];
  dj.pad_l = m.pad_l[job];
  const T* x = (const T*)m.x[job];
  const float* w = m.w[job];
  T* y = (T*)m.y[job];
  if (k == 3)
    dw_fwd_body<T, 3, S, OXT>(x, w, y, dj, relu_in, sw, blk);
  else if (k == 5)
    dw_fwd_body<T, 5, S, OXT>(x, w, y, dj, relu_in, sw, blk);
  else
    dw_fwd_body<T, 7, S, OXT>(x, w, y, dj, relu_in, sw, blk);
}

// ---------------------------------------------------------------- vectorised stencil (forward, stride-1 data gradient)
// The texture addresser spends the same ~16 cycles on a wave's 2-byte load as on a 16-byte one, so the scalar kernels
// above are address-bound.  Here a thread owns V consecutive channels of OXT neighbouring outputs: activations move as
// 8 / 16-byte vectors, and the filter taps (w[c][ky][kx], 100-196 bytes apart between neighbouring channels) are
// staged once per workgroup in LDS as sW[tap][c], read back as one vector per tap.
//   FLIP = 0: y = conv(f(x), w)                                 (f = relu when relu_in)
//   FLIP = 1: the stride-1 data gradient  dx = conv(dy, w rotated by 180 deg) with pad' = K - 1 - pad, masked by
//             (x > 0) when the forward applied the ReLU on the way in (`mask` = x).
template <typename T, int V> struct ChanVec;
template <> struct ChanVec<float, 4> { typedef float4 type; };
template <> struct ChanVec<float, 2> { typedef float2 type; };
template <> struct ChanVec<xpt_half_t, 8> { typedef uint4 type; };
template <> struct ChanVec<xpt_half_t, 4> { typedef uint2 type; };
template <> struct ChanVec<xpt_half_t, 2> { typedef unsigned type; };
template <> struct ChanVec<xpt_half_t, 1> { typedef unsigned short type; };

template <typename T, int V>
__device__ inline void load_chan(const T* p, float (&out)[V]) {
  typename ChanVec<T, V>::type raw = *(const typename ChanVec<T, V>::type*)p;
  const T* e = (const T*)&raw;
#pragma unroll
  for (int i = 0; i < V; ++i) out[i] = ldf<T>(e + i);
}

template <typename T, int V>
__device__ inline void store_chan(T* p, const float (&v)[V]) {
  typename ChanVec<T, V>::type raw;
  T* e = (T*)&raw;
#pragma unroll
  for (int i = 0; i < V; ++i) stf<T>(e + i, v[i]);
  *(typename ChanVec<T, V>::type*)p = raw;
}

template <typename T, int K, int S, int V, int OXT, int FLIP>
__global__ __launch_bounds__(256) void dw_stencil_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                          const T* __restrict__ mask, T* __restrict__ y, DwDims d,
                                                          int relu_in) {
  extern __shared__ __attribute__((aligned(16))) float sW[];       // [K*K][C]
  for (int i = threadIdx.x; i < d.C * K * K; i += 256) {
    const int c = i / (K * K), tap = i - c * (K * K);
    sW[(FLIP ? K * K - 1 - tap : tap) * d.C + c] = w[i];
  }
  __syncthreads();
  const int CG = d.C / V;
  const int OXG = (d.OW + OXT - 1) / OXT;
  const long long total = (long long)d.B * d.OH * OXG * CG;
  constexpr int IN = (OXT - 1) * S + K;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    int c0, oxg, oy, b;
    xpt_split4((unsigned)idx, CG, OXG, d.OH, c0, oxg, oy, b);
    c0 *= V;
    const int ox0 = oxg * OXT;
    const int ix0 = ox0 * S - d.pad_l;
    float acc[OXT][V];
#pragma unroll
    for (int i = 0; i < OXT; ++i)
#pragma unroll
      for (int v = 0; v < V; ++v) acc[i][v] = 0.f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int iy = oy * S + ky - d.pad_t;
      if (iy < 0 || iy >= d.H) continue;           // (guarded loads here: with wide vectors all K rows in flight would spill)
      const T* row = x + (((long long)b * d.H + iy) * d.W) * d.C + c0;
      float in[IN][V];
#pragma unroll
      for (int i = 0; i < IN; ++i) {
        const int ix = ix0 + i;
        if (ix >= 0 && ix < d.W) {
          load_chan<T, V>(row + (long long)ix * d.C, in[i]);
          if (relu_in && !FLIP) {
#pragma unroll
            for (int v = 0; v < V; ++v) in[i][v] = fmaxf(in[i][v], 0.f);
          }
        } else {
#pragma unroll
          for (int v = 0; v < V; ++v) in[i][v] = 0.f;
        }
      }
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        float wv[V];
        const float* wp = sW + (ky * K + kx) * d.C + c0;
#pragma unroll
        for (int v = 0; v < V; ++v) wv[v] = wp[v];
#pragma unroll
        for (int i = 0; i < OXT; ++i)
#pragma unroll
          for (int v = 0; v < V; ++v) acc[i][v] += in[i * S + kx][v] * wv[v];
      }
    }
    const long long o = (((long long)b * d.OH + oy) * d.OW + ox0) * d.C + c0;
#pragma unroll
    for (int i = 0; i < OXT; ++i) {
      if (ox0 + i >= d.OW) break;
      if (FLIP && relu_in) {
        float m[V];
        load_chan<T, V>(mask + o + (long long)i * d.C, m);
#pragma unroll
        for (int v = 0; v < V; ++v)
          if (!(m[v] > 0.f)) acc[i][v] = 0.f;
      }
      store_chan<T, V>(y + o + (long long)i * d.C, acc[i]);
    }
  }
}

// ---------------------------------------------------------------- vectorised multi-layer kernels (stride 1)
// The scalar multi-layer kernels below issue one 2-byte load per lane and tap: measured, their time is the texture
// addresser's instruction rate (dw_multi_bwd: 11 us at 0.27 M threads ... 57 us at 1.5 M, linear in the thread count).
// Here a thread owns V consecutive channels of OXT = 2 neighbouring outputs; the taps of the workgroup's job(s) sit in
// LDS as sW[tap][c] (read as one vector per tap), activations move as 4 / 8 / 16-byte vectors.
template <typename T, int K, int V, int OXT>
__device__ __forceinline__ void dw_vec_accumulate(const T* __restrict__ src, const float* __restrict__ sW, int C, int SH,
                                                  int SW, int pad_t, int pad_l, int b, int oy, int ox0, int c0, bool relu,
                                                  float (&acc)[OXT][V]) {
  constexpr int IN = OXT - 1 + K;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int sy = oy + ky - pad_t;
    const bool row_ok = sy >= 0 && sy < SH;
    const T* row = src + (((long long)b * SH + min(max(sy, 0), SH - 1)) * SW) * C + c0;
    float in[IN][V];
#pragma unroll
    for (int i = 0; i < IN; ++i) {                   // unconditional loads on clamped columns, zeroed by a select
      const int sx = ox0 - pad_l + i;
      load_chan<T, V>(row + (long long)min(max(sx, 0), SW - 1) * C, in[i]);
      const bool ok = row_ok && sx >= 0 && sx < SW;
#pragma unroll
      for (int v = 0; v < V; ++v) in[i][v] = ok ? (relu ? fmaxf(in[i][v], 0.f) : in[i][v]) : 0.f;
    }
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      float wv[V];
      const float* wp = sW + (ky * K + kx) * C + c0;
#pragma unroll
      for (int v = 0; v < V; ++v) wv[v] = wp[v];
#pragma unroll
      for (int i = 0; i < OXT; ++i)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[i][v] += in[i + kx][v] * wv[v];
    }
  }
}

template <typename T, int V>
__device__ __forceinline__ void dw_vec_accumulate_k(int k, const T* __restrict__ src, const float* __restrict__ sW, int C,
                                                    int SH, int SW, int pad_t, int pad_l, int b, int oy, int ox0, int c0,
                                                    bool relu, float (&acc)[2][V]) {
  if (k == 3) dw_vec_accumulate<T, 3, V, 2>(src, sW, C, SH, SW, pad_t, pad_l, b, oy, ox0, c0, relu, acc);
  else if (k == 5) dw_vec_accumulate<T, 5, V, 2>(src, sW, C, SH, SW, pad_t, pad_l, b, oy, ox0, c0, relu, acc);
  else dw_vec_accumulate<T, 7, V, 2>(src, sW, C, SH, SW, pad_t, pad_l, b, oy, ox0, c0, relu, acc);
}

// sW[(flip ? kk - 1 - tap : tap) * C + c] = w[c][tap]
__device__ inline void dw_stage_taps(float* sW, const float* __restrict__ w, int C, int kk, bool flip) {
  for (int i = threadIdx.x; i < C * kk; i += 256) {
    unsigned tap_;
    const int c = (int)xpt_divmod((unsigned)i, (unsigned)kk, tap_), tap = (int)tap_;
    sW[(flip ? kk - 1 - tap : tap) * C + c] = w[i];
  }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void dw_multi_fwd_vec_kernel(DwMultiFwd m, DwDims d, int relu_in, int blocks_per_job, int per,
                                                               int xcd) {
  extern __shared__ __attribute__((aligned(16))) float sW[];
  // grid: x = blocks of `per` consecutive outputs (pixel-major: image-to-XCD numbering, xpt_common.h), y = job
  const int job = blockIdx.y;
  unsigned blk;
  if (!xpt_xcd_unit(xcd != 0, blockIdx.x, (unsigned)blocks_per_job, blk)) return;
  const int k = m.k[job];
  dw_stage_taps(sW, m.w[job], d.C, k * k, false);
  __syncthreads();
  const T* x = (const T*)m.x[job];
  T* y = (T*)m.y[job];
  const int CG = d.C / V, OXG = (d.OW + 1) / 2;
  const long long total = (long long)d.B * d.OH * OXG * CG;
  const long long begin = (long long)blk * per, end = begin + per < total ? begin + per : total;
  for (long long idx = begin + threadIdx.x; idx < end; idx += 256) {
    int c0, ox0, oy, b;
    xpt_split4((unsigned)idx, CG, OXG, d.OH, c0, ox0, oy, b);
    c0 *= V;
    ox0 *= 2;
    float acc[2][V];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int v = 0; v < V; ++v) acc[i][v] = 0.f;
    dw_vec_accumulate_k<T, V>(k, x, sW, d.C, d.H, d.W, m.pad_t[job], m.pad_l[job], b, oy, ox0, c0, relu_in != 0, acc);
    const long long o = (((long long)b * d.OH + oy) * d.OW + ox0) * d.C + c0;
    store_chan<T, V>(y + o, acc[0]);
    if (ox0 + 1 < d.OW) store_chan<T, V>(y + o + d.C, acc[1]);
  }
}

// ---------------------------------------------------------------- small maps: the whole map of an image in LDS (stride 1, bf16)
// On the 8 x 26 and 4 x 13 maps of the deeper stacks a launch of the stencil kernels above is a few dozen workgroups whose
// threads walk the filter rows one memory round trip at a time (k rows, and the backward the rows of up to three filters:
// 11 dependent trips on 73 k elements).  Here a workgroup owns (job, image, chunk of CW channels): it stages the image's
// whole map of those channels and the taps in LDS with ONE burst of loads and computes every output of the chunk from LDS,
// in the same (ky, kx) order with the same fused multiply-adds as the stencil kernels: same bits.
//   FLIP = 0: y = conv(f(x), w);   FLIP = 1: the data gradient dx_u = [x_u > 0] * sum over the jobs reading input u of
//   conv(dy_j, w_j rotated by 180 degrees, pad' = k_j - 1 - pad_j).
__device__ __forceinline__ float bf16_to_f32(unsigned short u) { return xpt_h2f(u); }

template <int G> struct SmallVec;
template <> struct SmallVec<8> { typedef uint4 type; };
template <> struct SmallVec<4> { typedef uint2 type; };
template <> struct SmallVec<2> { typedef unsigned type; };
template <> struct SmallVec<1> { typedef unsigned short type; };

// G = channels of a group (8: 16-byte vectors; 4: 8-byte vectors, the 44-channel stack); the LDS map holds cw channels per pixel
template <int K, int G>
__device__ __forceinline__ void dw_small_accumulate(const unsigned short* __restrict__ xs, const float* __restrict__ ws, int H,
                                                    int W, int cw, int oy, int ox, int c8, int pad_t, int pad_l, bool relu,
                                                    float (&acc)[G]) {
  typedef typename SmallVec<G>::type vec_t;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int sy = oy + ky - pad_t;
    const bool row_ok = sy >= 0 && sy < H;
    const int cy = min(max(sy, 0), H - 1);
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int sx = ox + kx - pad_l;
      const bool ok = row_ok && sx >= 0 && sx < W;
      const vec_t raw = *(const vec_t*)(xs + (cy * W + min(max(sx, 0), W - 1)) * cw + c8);
      const unsigned short* e = (const unsigned short*)&raw;
      const float* wp = ws + (ky * K + kx) * cw + c8;
#pragma unroll
      for (int v = 0; v < G; ++v) {
        const float f = bf16_to_f32(e[v]);
        const float in = ok ? (relu ? fmaxf(f, 0.f) : f) : 0.f;
        acc[v] += in * wp[v];
      }
    }
  }
}

template <int G>
__device__ __forceinline__ void dw_small_accumulate_k(int k, const unsigned short* __restrict__ xs, const float* __restrict__ ws,
                                                      int H, int W, int cw, int oy, int ox, int c8, int pad_t, int pad_l,
                                                      bool relu, float (&acc)[G]) {
  if (k == 3) dw_small_accumulate<3, G>(xs, ws, H, W, cw, oy, ox, c8, pad_t, pad_l, relu, acc);
  else if (k == 5) dw_small_accumulate<5, G>(xs, ws, H, W, cw, oy, ox, c8, pad_t, pad_l, relu, acc);
  else dw_small_accumulate<7, G>(xs, ws, H, W, cw, oy, ox, c8, pad_t, pad_l, relu, acc);
}

// stage the map of `src` (image b, channels c_lo .. c_lo + cw) and the taps of `w` (rotated when flip) into LDS
template <int G>
__device__ __forceinline__ void dw_small_stage(unsigned short* xs, float* ws, const unsigned short* __restrict__ src,
                                               const float* __restrict__ w, int b, int HW, int C, int c_lo, int cw, int kk,
                                               bool flip) {
  typedef typename SmallVec<G>::type vec_t;
  const int g8 = cw / G, nvec = HW * g8;
  for (int v = threadIdx.x; v < nvec; v += 256) {
    unsigned g_;
    const unsigned pix = xpt_divmod((unsigned)v, (unsigned)g8, g_);
    *(vec_t*)(xs + pix * cw + g_ * G) = *(const vec_t*)(src + ((long long)b * HW + pix) * C + c_lo + g_ * G);
  }
  for (int i = threadIdx.x; i < cw * kk; i += 256) {
    unsigned tap_;
    const unsigned c = xpt_divmod((unsigned)i, (unsigned)kk, tap_);
    ws[(flip ? kk - 1 - (int)tap_ : (int)tap_) * cw + c] = w[(long long)(c_lo + c) * kk + tap_];
  }
}

template <int G>
__global__ __launch_bounds__(256) void dw_small_fwd_kernel(DwMultiFwd m, DwDims d, int relu_in, int CW) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  // (numbering the images onto the XCDs was measured here, grid (images, jobs, chunks): 7.8 -> 8.2 us per launch -- the
  //  workgroups that share an image's cache lines, one per 8-channel chunk, then queue on one L2 instead of eight)
  const int job = blockIdx.x, b = blockIdx.y, c_lo = blockIdx.z * CW;
  const int cw = d.C - c_lo < CW ? d.C - c_lo : CW;                     // multiple of G
  const int HW = d.H * d.W, k = m.k[job], kk = k * k;
  unsigned short* xs = (unsigned short*)sm;                             // [HW][cw] bf16
  float* ws = (float*)(sm + (((size_t)HW * CW * 2 + 15) & ~(size_t)15));      // [kk][cw]
  dw_small_stage<G>(xs, ws, (const unsigned short*)m.x[job], m.w[job], b, HW, d.C, c_lo, cw, kk, false);
  __syncthreads();
  const int g8 = cw / G, items = HW * g8;
  for (int it = threadIdx.x; it < items; it += 256) {
    unsigned g_, ox_;
    const unsigned pix = xpt_divmod((unsigned)it, (unsigned)g8, g_);
    const int oy = (int)xpt_divmod(pix, (unsigned)d.W, ox_), ox = (int)ox_, c8 = (int)g_ * G;
    float acc[G];
#pragma unroll
    for (int v = 0; v < G; ++v) acc[v] = 0.f;
    dw_small_accumulate_k<G>(k, xs, ws, d.H, d.W, cw, oy, ox, c8, m.pad_t[job], m.pad_l[job], relu_in != 0, acc);
    store_chan<xpt_half_t, G>((xpt_half_t*)m.y[job] + ((long long)b * HW + pix) * d.C + c_lo + c8, acc);
  }
}

// ---------------------------------------------------------------- tiles: the input region of an output tile in LDS (stride 2, bf16)
// The stride-2 layers (the stem's 7x7 / 5x5 on the 64 x 208 map, the first layers of the reduction cells) ran on the stencil /
// scalar kernels above, whose threads walk the filter rows one memory round trip at a time (k dependent trips: 17 us forward
// and 30 us data gradient for 8.5 MB on the stem).  Here a workgroup owns (job, image, output tile, chunk of CW channels):
// it stages the input region of the tile ((TH-1) S + k rows, zero outside the map) and the taps in LDS with ONE burst of
// loads and computes the tile from LDS; (ky, kx) order and fused multiply-adds as everywhere else in this file: same bits
// as the kernels it replaces in the forward.
struct DwTile {
  int TH, TW;           // tile (output pixels forward; input pixels in the data gradient)
  int tiles_x, tiles;   // tiles along x, tiles per image
  int CW, chunks;       // channels per workgroup, chunks per pixel
  int map_bytes;        // LDS bytes of one staged region (largest kernel size of the launch), multiple of 16
  int slot_bytes;       // data gradient: region + taps of one job
  int xcd;              // image-to-XCD numbering of the workgroups (xpt_common.h)
};

// rows [y0, y0 + RH) x columns [x0, x0 + RW) of image b of src (H x W x C), channels [c_lo, c_lo + cw), zero outside the map
template <int G>
__device__ __forceinline__ void dw_tile_stage(unsigned short* xs, const unsigned short* __restrict__ src, int b, int H, int W,
                                              int C, int c_lo, int cw, int y0, int x0, int RH, int RW) {
  typedef typename SmallVec<G>::type vec_t;
  const int cg = cw / G, n = RH * RW * cg;
  for (int i = threadIdx.x; i < n; i += 256) {
    unsigned g_, rx_;
    const unsigned p = xpt_divmod((unsigned)i, (unsigned)cg, g_);
    const int ry = (int)xpt_divmod(p, (unsigned)RW, rx_);
    const int y = y0 + ry, x = x0 + (int)rx_;
    const bool ok = y >= 0 && y < H && x >= 0 && x < W;
    vec_t v = *(const vec_t*)(src + (((long long)b * H + min(max(y, 0), H - 1)) * W + min(max(x, 0), W - 1)) * C + c_lo + g_ * G);
    if (!ok) v = vec_t{};
    *(vec_t*)(xs + p * cw + g_ * G) = v;
  }
}

// taps of channels [c_lo, c_lo + cw) as ws[ky * KP + kx][cw], KP >= k (entries past k are zero: the parity classes of the
// stride-2 data gradient visit them)
__device__ __forceinline__ void dw_tile_stage_taps(float* ws, const float* __restrict__ w, int c_lo, int cw, int k, int KP) {
  for (int i = threadIdx.x; i < cw * KP * KP; i += 256) {
    unsigned c_, kx_;
    const unsigned tap = xpt_divmod((unsigned)i, (unsigned)cw, c_);
    const int ky = (int)xpt_divmod(tap, (unsigned)KP, kx_), kx = (int)kx_;
    ws[i] = (ky < k && kx < k) ? w[(long long)(c_lo + (int)c_) * k * k + ky * k + kx] : 0.f;
  }
}

template <int K, int S, int G>
__device__ __forceinline__ void dw_tile_accumulate(const unsigned short* __restrict__ xs, const float* __restrict__ ws, int RW,
                                                   int cw, int ly, int lx, int c8, bool relu, float (&acc)[G]) {
  typedef typename SmallVec<G>::type vec_t;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const vec_t raw = *(const vec_t*)(xs + ((ly * S + ky) * RW + lx * S + kx) * cw + c8);
      const unsigned short* e = (const unsigned short*)&raw;
      const float* wp = ws + (ky * K + kx) * cw + c8;
#pragma unroll
      for (int v = 0; v < G; ++v) {
        const float f = bf16_to_f32(e[v]);
        const float in = relu ? fmaxf(f, 0.f) : f;
        acc[v] += in * wp[v];
      }
    }
  }
}

template <int S, int G>
__global__ __launch_bounds__(256) void dw_tile_fwd_kernel(DwMultiFwd m, DwDims d, int relu_in, DwTile t) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  unsigned u_;                                     // grid: x = (image, tile) image-major (image-to-XCD numbering), y = job, z = chunk
  if (!xpt_xcd_unit(t.xcd != 0, blockIdx.x, (unsigned)(d.B * t.tiles), u_)) return;
  const int job = blockIdx.y, b = (int)u_ / t.tiles, tile = (int)u_ - b * t.tiles;
  const int ty = tile / t.tiles_x, tx = tile - ty * t.tiles_x;
  const int c_lo = blockIdx.z * t.CW;
  const int cw = d.C - c_lo < t.CW ? d.C - c_lo : t.CW;                   // multiple of G
  const int k = m.k[job];
  const int oy0 = ty * t.TH, ox0 = tx * t.TW;
  const int RH = (t.TH - 1) * S + k, RW = (t.TW - 1) * S + k;
  unsigned short* xs = (unsigned short*)sm;                               // [RH][RW][cw] bf16
  float* ws = (float*)(sm + t.map_bytes);                                 // [k*k][cw]
  dw_tile_stage<G>(xs, (const unsigned short*)m.x[job], b, d.H, d.W, d.C, c_lo, cw, oy0 * S - m.pad_t[job],
                   ox0 * S - m.pad_l[job], RH, RW);
  dw_tile_stage_taps(ws, m.w[job], c_lo, cw, k, k);
  __syncthreads();
  const int cg = cw / G, items = t.TH * t.TW * cg;
  for (int it = threadIdx.x; it < items; it += 256) {
    unsigned g_, lx_;
    const unsigned px = xpt_divmod((unsigned)it, (unsigned)cg, g_);
    const int ly = (int)xpt_divmod(px, (unsigned)t.TW, lx_), lx = (int)lx_, c8 = (int)g_ * G;
    const int oy = oy0 + ly, ox = ox0 + lx;
    if (oy >= d.OH || ox >= d.OW) continue;
    float acc[G];
#pragma unroll
    for (int v = 0; v < G; ++v) acc[v] = 0.f;
    if (k == 3) dw_tile_accumulate<3, S, G>(xs, ws, RW, cw, ly, lx, c8, relu_in != 0, acc);
    else if (k == 5) dw_tile_accumulate<5, S, G>(xs, ws, RW, cw, ly, lx, c8, relu_in != 0, acc);
    else dw_tile_accumulate<7, S, G>(xs, ws, RW, cw, ly, lx, c8, relu_in != 0, acc);
    store_chan<xpt_half_t, G>((xpt_half_t*)m.y[job] + (((long long)b * d.OH + oy) * d.OW + ox) * d.C + c_lo + c8, acc);
  }
}

// ---------------------------------------------------------------- data gradient
// dx[b,iy,ix,c] = [x>0 if relu] * sum_{ky,kx : (iy+pad_t-ky) % S == 0, ...} dy[b,(iy+pad_t-ky)/S,(ix+pad_l-kx)/S,c] * w[c,ky,kx]
// sum over the taps of one input element: dy[b, (iy+pad_t-ky)/S, (ix+pad_l-kx)/S, c] * w[c, ky, kx]
template <typename T, int K, int S>
__device__ inline float dw_bwd_data_value(const float* __restrict__ w, const T* __restrict__ dy, const DwDims& d, int c,
                                          int iy, int ix, int b) {
  const float* wc = w + (long long)c * K * K;
  float acc = 0.f;
  // stride 2: only the taps of matching parity reach an output, ky = py, py + 2, ... (a quarter of the K*K taps)
  const int py = (S == 1) ? 0 : ((iy + d.pad_t) & 1), px = (S == 1) ? 0 : ((ix + d.pad_l) & 1);
  constexpr int KT = (K + S - 1) / S;
#pragma unroll
  for (int jy = 0; jy < KT; ++jy) {
    const int ky = py + jy * S;
    const int ty = iy + d.pad_t - ky;                // a multiple of S by construction
    const int oy = ty / S;
    const bool row_ok = ky < K && ty >= 0 && oy < d.OH;
    const T* row = dy + (((long long)b * d.OH + min(max(oy, 0), d.OH - 1)) * d.OW) * d.C + c;
#pragma unroll
    for (int jx = 0; jx < KT; ++jx) {
      const int kx = px + jx * S;
      const int tx = ix + d.pad_l - kx;
      const int ox = tx / S;
      const bool ok = row_ok && kx < K && tx >= 0 && ox < d.OW;
      const float v = ldf<T>(row + (long long)min(max(ox, 0), d.OW - 1) * d.C);     // unconditional, see dw_fwd_body
      const float wv = wc[min(ky, K - 1) * K + min(kx, K - 1)];
      acc += ok ? v * wv : 0.f;
    }
  }
  return acc;
}

template <typename T, int K, int S>
__device__ inline void dw_bwd_data_body(const T* __restrict__ x, const float* __restrict__ w, const T* __restrict__ dy,
                                        T* __restrict__ dx, const DwDims& d, int relu_in, long long block,
                                        long long nblocks) {
  const long long total = (long long)d.B * d.H * d.W * d.C;
  long long idx, end;
  xpt_chunk_range((unsigned)block, (unsigned)nblocks, total, idx, end);      // consecutive (pixel-major) indices per workgroup
  for (idx += threadIdx.x; idx < end; idx += 256) {
    int c, ix, iy, b;
    xpt_split4((unsigned)idx, d.C, d.W, d.H, c, ix, iy, b);
    float acc = dw_bwd_data_value<T, K, S>(w, dy, d, c, iy, ix, b);
    if (relu_in && !(ldf<T>(x + idx) > 0.f)) acc = 0.f;
    stf<T>(dx + idx, acc);
  }
}

template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(const T* __restrict__ x, const float* __restrict__ w, const T* __restrict__ dy,
                                   T* __restrict__ dx, DwDims d, int relu_in, int nblocks, int xcd) {
  unsigned blk;
  if (!xpt_xcd_unit(xcd != 0, blockIdx.x, (unsigned)nblocks, blk)) return;
  dw_bwd_data_body<T, K, S>(x, w, dy, dx, d, relu_in, blk, nblocks);
}

// Stride-2 data gradient with V channels per thread (the reduction cells' first layers on the 64x208 / 32x104 maps: a
// 2-byte load per lane and tap costs the address path what a 16-byte one costs): the taps sit in LDS as [tap][channel],
// a thread owns V consecutive channels of one input pixel and visits the (K+1)/2 x (K+1)/2 taps of its parity class.
template <typename T, int K, int V>
__global__ __launch_bounds__(256) void dw_bwd_data_s2_vec_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                                  const T* __restrict__ dy, T* __restrict__ dx, DwDims d,
                                                                  int relu_in) {
  extern __shared__ __attribute__((aligned(16))) float sWt[];     // [K*K][C]
  dw_stage_taps(sWt, w, d.C, K * K, false);
  __syncthreads();
  constexpr int KT = (K + 1) / 2;
  const int CG = d.C / V;
  const long long total = (long long)d.B * d.H * d.W * CG;
  for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += 256LL * gridDim.x) {
    int c0, ix, iy, b;
    xpt_split4((unsigned)idx, CG, d.W, d.H, c0, ix, iy, b);
    c0 *= V;
    float acc[V];
#pragma unroll
    for (int u = 0; u < V; ++u) acc[u] = 0.f;
    const int py = (iy + d.pad_t) & 1, px = (ix + d.pad_l) & 1;
#pragma unroll
    for (int jy = 0; jy < KT; ++jy) {
      const int ky = py + 2 * jy;
      const int ty = iy + d.pad_t - ky;                // even by construction
      const int oy = ty >> 1;
      const bool row_ok = ky < K && ty >= 0 && oy < d.OH;
      const T* row = dy + (((long long)b * d.OH + min(max(oy, 0), d.OH - 1)) * d.OW) * d.C + c0;
#pragma unroll
      for (int jx = 0; jx < KT; ++jx) {
        const int kx = px + 2 * jx;
        const int tx = ix + d.pad_l - kx;
        const int ox = tx >> 1;
        const bool ok = row_ok && kx < K && tx >= 0 && ox < d.OW;
        float v[V];
        load_chan<T, V>(row + (long long)min(max(ox, 0), d.OW - 1) * d.C, v);      // unconditional, zeroed by select
        const float* wt = sWt + (min(ky, K - 1) * K + min(kx, K - 1)) * d.C + c0;
#pragma unroll
        for (int u = 0; u < V; ++u) acc[u] += ok ? v[u] * wt[u] : 0.f;
      }
    }
    const long long o = (((long long)b * d.H + iy) * d.W + ix) * d.C + c0;
    if (relu_in) {
      float m[V];
      load_chan<T, V>(x + o, m);
#pragma unroll
      for (int u = 0; u < V; ++u)
        if (!(m[u] > 0.f)) acc[u] = 0.f;
    }
    store_chan<T, V>(dx + o, acc);
  }
}

// ---------------------------------------------------------------- weight gradient
// dw[c,ky,kx] = sum_{b,oy,ox} dy[b,oy,ox,c] * f(x[b,oy*S+ky-pad_t,ox*S+kx-pad_l,c])
// Work item = OXT neighbouring outputs of one row ("group"); grid (channel chunks of 64, chunks of GRP groups).
// Lanes run along channels; when C <= 32 the spare lanes take further row groups (RG = 64 / C per wave) and are
// folded with fixed-order shuffles.  Every thread slides the k-wide input window over its OXT outputs in registers.
int g_dw_wrw_grp = 0;   // groups of OXT outputs per workgroup; 0 = chosen per shape (xpt_dwconv_tune overrides)

// One group per wave keeps the per-wave dependent work shortest (measured: time grows linearly with the groups a wave
// walks); narrow tensors need 4 x (row groups packed into a wave) groups to fill their lanes, very large maps larger
// chunks to bound the number of partial rows.
inline int wrw_groups(long long ngrp, int C) {
  if (g_dw_wrw_grp) return g_dw_wrw_grp;
  int grp = 4;
  if (C <= 32) {
    const int rg = 64 / C;                       // row groups a wave can hold side by side
    grp = rg >= 4 ? 16 : (rg >= 2 ? 8 : 4);
  }
  const long long cchunks = (C + 63) / 64;
  while (grp < 32 && ((ngrp + grp - 1) / grp) * cchunks > 2048) grp <<= 1;
  return grp;
}
// (EXT: the cross-wave fold uses caller-provided LDS -- 3 * 64 * K * K floats -- instead of a static array, for kernels
//  that already carry dynamic LDS)
template <typename T, int K, int S, bool EXT = false>
__device__ inline void dw_bwd_weight_body(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part,
                                          const DwDims& d, int relu_in, int RG, int GRP, int block_c,
                                          long long block_g, float* red_ext = nullptr) {
  constexpr int OXT = (S == 1) ? 4 : 2;
  constexpr int IN = (OXT - 1) * S + K;
  __shared__ float red_static[EXT ? 1 : 3 * 64 * K * K];
  float (*red)[64 * K * K] = (float (*)[64 * K * K])(EXT ? red_ext : red_static);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rg = (RG > 1) ? lane / d.C : 0;
  const int cl = (RG > 1) ? lane - rg * d.C : lane;
  const int c = block_c * 64 + cl;
  const bool active = (c < d.C) && (rg < RG);
  const int OXG = (d.OW + OXT - 1) / OXT;
  const long long ngrp = (long long)d.B * d.OH * OXG;
  const long long g0 = block_g * GRP;
  float acc[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) acc[i] = 0.f;
  if (active) {
    for (int i = wid * RG + rg; i < GRP; i += 4 * RG) {
      const long long g = g0 + i;
      if (g >= ngrp) break;
      // (ngrp < 2^31: the launchers' grids are 32-bit; three 64-bit divisions were ~360 of this loop body's ~560 instructions)
      unsigned oxg_u, oy_u;
      const int b = (int)xpt_divmod(xpt_divmod((unsigned)g, (unsigned)OXG, oxg_u), (unsigned)d.OH, oy_u);
      const int oxg = (int)oxg_u, oy = (int)oy_u;
      const int ox0 = oxg * OXT;
      const int ix0 = ox0 * S - d.pad_l;
      float gy[OXT];
      const T* dyp = dy + (((long long)b * d.OH + oy) * d.OW + ox0) * d.C + c;
#pragma unroll
      for (int j = 0; j < OXT; ++j) {
        const float v = ldf<T>(dyp + (long long)min(j, d.OW - 1 - ox0) * d.C);            // unconditional, see dw_fwd_kernel
        gy[j] = (ox0 + j < d.OW) ? v : 0.f;
      }
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * S + ky - d.pad_t;
        const bool row_ok = iy >= 0 && iy < d.H;
        const T* row = x + (((long long)b * d.H + min(max(iy, 0), d.H - 1)) * d.W) * d.C + c;
        float in[IN];
#pragma unroll
        for (int t = 0; t < IN; ++t) {
          const int ix = ix0 + t;
          const float v = ldf<T>(row + (long long)min(max(ix, 0), d.W - 1) * d.C);
          in[t] = (row_ok && ix >= 0 && ix < d.W) ? (relu_in ? fmaxf(v, 0.f) : v) : 0.f;
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          float a = acc[ky * K + kx];
#pragma unroll
          for (int j = 0; j < OXT; ++j) a += gy[j] * in[j * S + kx];
          acc[ky * K + kx] = a;
        }
      }
    }
  }
  if (RG > 1) {      // fold the row groups of this wave onto lanes [0, C): fixed order -> deterministic
#pragma unroll
    for (int i = 0; i < K * K; ++i) {
      float v = acc[i];
      for (int r = 1; r < RG; ++r) v += __shfl(acc[i], (cl + r * d.C) & 63, 64);
      acc[i] = v;
    }
  }
  if (wid > 0) {
#pragma unroll
    for (int i = 0; i < K * K; ++i) red[wid - 1][i * 64 + lane] = acc[i];
  }
  __syncthreads();
  if (wid == 0 && c < d.C && rg == 0) {
    float* out = part + (block_g * d.C + c) * (K * K);
#pragma unroll
    for (int i = 0; i < K * K; ++i) out[i] = ((acc[i] + red[0][i * 64 + lane]) + red[1][i * 64 + lane]) + red[2][i * 64 + lane];
  }
}

template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part,
                                     DwDims d, int relu_in, int RG, int GRP) {
  dw_bwd_weight_body<T, K, S>(x, dy, part, d, relu_in, RG, GRP, blockIdx.x, blockIdx.y);
}

// Data gradient and weight-gradient partials of one layer in ONE launch: the two only share their inputs (dy, x, w), so
// the first `data_blocks` workgroups run the data-gradient body and the rest the weight-gradient body -- at batch 8
// both are launch-latency-bound and a second dependent launch costs as much as the work itself.
template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dw_bwd_both_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                          const T* __restrict__ dy, T* __restrict__ dx,
                                                          float* __restrict__ part, DwDims d, int relu_in, int RG,
                                                          int GRP, int data_blocks, int cchunks, int wblocks, int xcd) {
  unsigned vb;                                  // (image-to-XCD numbering of both classes of workgroups, xpt_common.h)
  if (!xpt_xcd_two_class(xcd != 0, blockIdx.x, 1u, (unsigned)data_blocks, 1u, (unsigned)wblocks, vb)) return;
  if ((int)vb < data_blocks) {
    dw_bwd_data_body<T, K, S>(x, w, dy, dx, d, relu_in, vb, data_blocks);
  } else {
    const int t = (int)vb - data_blocks;
    dw_bwd_weight_body<T, K, S>(x, dy, part, d, relu_in, RG, GRP, t % cchunks, t / cchunks);
  }
}

// Backward of dw_multi_fwd_kernel in one launch: the first n_inputs * data_blocks workgroups write the data gradient of
// each DISTINCT input (summed over the jobs that read it: no gradient fan-in pass), the rest the weight-gradient
// partials of every job.
struct DwMultiBwd {
  const void* xin[DW_MAX_JOBS];     // distinct inputs
  void* dxin[DW_MAX_JOBS];
  const void* dy[DW_MAX_JOBS];      // per job
  const float* w[DW_MAX_JOBS];
  float* part[DW_MAX_JOBS];
  int k[DW_MAX_JOBS], pad_t[DW_MAX_JOBS], pad_l[DW_MAX_JOBS];
  int input_of[DW_MAX_JOBS];
  int n, n_inputs;
};

template <typename T, int S>
__global__ __launch_bounds__(256) void dw_multi_bwd_kernel(DwMultiBwd m, DwDims d, int relu_in, int RG, int GRP,
                                                           int data_blocks, int cchunks, int wblocks_per_job, int xcd) {
  // the weight-gradient workgroups' fold buffer, sized by the host for the largest kernel among the jobs (one static
  // buffer per kernel size added up to 63.7 KB and held the launch at two workgroups per CU)
  extern __shared__ __attribute__((aligned(16))) float sFold[];
  const int ndata = m.n_inputs * data_blocks;
  unsigned vb;                                  // (image-to-XCD numbering of both classes of workgroups, xpt_common.h)
  if (!xpt_xcd_two_class(xcd != 0, blockIdx.x, (unsigned)m.n_inputs, (unsigned)data_blocks, (unsigned)m.n, (unsigned)wblocks_per_job, vb))
    return;
  if ((int)vb < ndata) {
    const int u = vb / data_blocks, blk = vb - u * data_blocks;
    const T* x = (const T*)m.xin[u];
    T* dx = (T*)m.dxin[u];
    const long long total = (long long)d.B * d.H * d.W * d.C;
    long long idx, end;
    xpt_chunk_range((unsigned)blk, (unsigned)data_blocks, total, idx, end);
    for (idx += threadIdx.x; idx < end; idx += 256) {
      int c, ix, iy, b;
      xpt_split4((unsigned)idx, d.C, d.W, d.H, c, ix, iy, b);
      float acc = 0.f;
      for (int j = 0; j < m.n; ++j) {
        if (m.input_of[j] != u) continue;
        DwDims dj = d;
        dj.pad_t = m.pad_t[j];
        dj.pad_l = m.pad_l[j];
        const T* dy = (const T*)m.dy[j];
        if (m.k[j] == 3)
          acc += dw_bwd_data_value<T, 3, S>(m.w[j], dy, dj, c, iy, ix, b);
        else if (m.k[j] == 5)
          acc += dw_bwd_data_value<T, 5, S>(m.w[j], dy, dj, c, iy, ix, b);
        else
          acc += dw_bwd_data_value<T, 7, S>(m.w[j], dy, dj, c, iy, ix, b);
      }
      if (relu_in && !(ldf<T>(x + idx) > 0.f)) acc = 0.f;
      stf<T>(dx + idx, acc);
    }
    return;
  }
  const int t = (int)vb - ndata;
  const int job = t / wblocks_per_job, tt = t - job * wblocks_per_job;
  DwDims dj = d;
  dj.pad_t = m.pad_t[job];
  dj.pad_l = m.pad_l[job];
  const T* x = (const T*)m.xin[m.input_of[job]];
  const T* dy = (const T*)m.dy[job];
  if (m.k[job] == 3)
    dw_bwd_weight_body<T, 3, S, true>(x, dy, m.part[job], dj, relu_in, RG, GRP, tt % cchunks, tt / cchunks, sFold);
  else if (m.k[job] == 5)
    dw_bwd_weight_body<T, 5, S, true>(x, dy, m.part[job], dj, relu_in, RG, GRP, tt % cchunks, tt / cchunks, sFold);
  else
    dw_bwd_weight_body<T, 7, S, true>(x, dy, m.part[job], dj, relu_in, RG, GRP, tt % cchunks, tt / cchunks, sFold);
}

// Stride-2 backward with the data-gradient part on tiles: a workgroup owns (input u, image, tile of dx, chunk of CW channels);
// for every job reading u it stages the region of dy_j the tile's taps reach (zero outside the map) and the taps (as
// [ky][kx] padded to even extents with zeros: a parity class visits ky = py, py + 2, ...) in LDS in one burst; each dx
// element sums its (k+1)/2 x (k+1)/2 taps per job, jobs added in job order like the kernel above.
template <int K, int S, int G>
__device__ __forceinline__ void dw_tile_bwd_accumulate(const unsigned short* __restrict__ ds, const float* __restrict__ ws,
                                                       int RW, int cw, int ty, int tx, int oy_min, int ox_min, int c8,
                                                       float (&acc)[G]) {
  typedef typename SmallVec<G>::type vec_t;
  constexpr int KT = (S == 2) ? (K + 1) / 2 : K, KP = (S == 2) ? 2 * KT : K;
  const int py = (S == 2) ? (ty & 1) : 0, px = (S == 2) ? (tx & 1) : 0;      // ty = iy + pad_t, tx = ix + pad_l
  float a[G];
#pragma unroll
  for (int v = 0; v < G; ++v) a[v] = 0.f;
#pragma unroll
  for (int jy = 0; jy < KT; ++jy) {
    const int ky = py + S * jy;
    const int ry = (S == 2 ? ((ty - ky) >> 1) : ty - ky) - oy_min;
#pragma unroll
    for (int jx = 0; jx < KT; ++jx) {
      const int kx = px + S * jx;
      const int rx = (S == 2 ? ((tx - kx) >> 1) : tx - kx) - ox_min;
      const vec_t raw = *(const vec_t*)(ds + (ry * RW + rx) * cw + c8);
      const unsigned short* e = (const unsigned short*)&raw;
      const float* wp = ws + (ky * KP + kx) * cw + c8;
#pragma unroll
      for (int v = 0; v < G; ++v) a[v] += bf16_to_f32(e[v]) * wp[v];
    }
  }
#pragma unroll
  for (int v = 0; v < G; ++v) acc[v] += a[v];
}

template <int S, int G>
__device__ __forceinline__ void dw_tile_bwd_block(const DwMultiBwd& m, const DwDims& d, int relu_in, const DwTile& t, int blk,
                                                   unsigned char* sm) {
  // blk = ((u * B + b) * tiles + tile) * chunks + chunk
  int r = blk;
  const int chunk = r % t.chunks; r /= t.chunks;
  const int tile = r % t.tiles; r /= t.tiles;
  const int b = r % d.B, u = r / d.B;
  const int tyi = tile / t.tiles_x, txi = tile - tyi * t.tiles_x;
  const int c_lo = chunk * t.CW;
  const int cw = d.C - c_lo < t.CW ? d.C - c_lo : t.CW;
  const int iy0 = tyi * t.TH, ix0 = txi * t.TW;
  int nj = 0;
  for (int j = 0; j < m.n; ++j) {
    if (m.input_of[j] != u) continue;
    const int k = m.k[j], KP = (S == 2) ? k + 1 : k;                      // stride 2, k odd: 2 * ((k + 1) / 2)
    const int sh = S - 1;                                                 // (v >> 1 floors; stride 1: no shift)
    const int y_lo = (iy0 + m.pad_t[j] - (KP - 1)) >> sh, y_hi = (iy0 + t.TH - 1 + m.pad_t[j]) >> sh;
    const int x_lo = (ix0 + m.pad_l[j] - (KP - 1)) >> sh, x_hi = (ix0 + t.TW - 1 + m.pad_l[j]) >> sh;
    unsigned char* slot = sm + (size_t)nj * t.slot_bytes;
    dw_tile_stage<G>((unsigned short*)slot, (const unsigned short*)m.dy[j], b, d.OH, d.OW, d.C, c_lo, cw, y_lo, x_lo,
                     y_hi - y_lo + 1, x_hi - x_lo + 1);
    dw_tile_stage_taps((float*)(slot + t.map_bytes), m.w[j], c_lo, cw, k, KP);
    ++nj;
  }
  __syncthreads();
  const unsigned short* x = (const unsigned short*)m.xin[u];
  const int cg = cw / G, items = t.TH * t.TW * cg;
  for (int it = threadIdx.x; it < items; it += 256) {
    unsigned g_, lx_;
    const unsigned px = xpt_divmod((unsigned)it, (unsigned)cg, g_);
    const int ly = (int)xpt_divmod(px, (unsigned)t.TW, lx_), c8 = (int)g_ * G;
    const int iy = iy0 + ly, ix = ix0 + (int)lx_;
    if (iy >= d.H || ix >= d.W) continue;
    const long long o = (((long long)b * d.H + iy) * d.W + ix) * d.C + c_lo + c8;
    float mk[G], acc[G];
#pragma unroll
    for (int v = 0; v < G; ++v) {
      mk[v] = 1.f;
      acc[v] = 0.f;
    }
    if (relu_in) load_chan<xpt_half_t, G>((const xpt_half_t*)x + o, mk);
    int q = 0;
    for (int j = 0; j < m.n; ++j) {
      if (m.input_of[j] != u) continue;
      const unsigned char* slot = sm + (size_t)q * t.slot_bytes;
      const unsigned short* ds = (const unsigned short*)slot;
      const float* ws = (const float*)(slot + t.map_bytes);
      const int k = m.k[j], tyy = iy + m.pad_t[j], txx = ix + m.pad_l[j];
      const int sh = S - 1, reach = (S == 2) ? k : k - 1;                 // the region's origin and pitch, as staged above
      const int y_lo = (iy0 + m.pad_t[j] - reach) >> sh;
      const int x_lo = (ix0 + m.pad_l[j] - reach) >> sh, rw = ((ix0 + t.TW - 1 + m.pad_l[j]) >> sh) - x_lo + 1;
      if (k == 3) dw_tile_bwd_accumulate<3, S, G>(ds, ws, rw, cw, tyy, txx, y_lo, x_lo, c8, acc);
      else if (k == 5) dw_tile_bwd_accumulate<5, S, G>(ds, ws, rw, cw, tyy, txx, y_lo, x_lo, c8, acc);
      else dw_tile_bwd_accumulate<7, S, G>(ds, ws, rw, cw, tyy, txx, y_lo, x_lo, c8, acc);
      ++q;
    }
    if (relu_in) {
#pragma unroll
      for (int v = 0; v < G; ++v)
        if (!(mk[v] > 0.f)) acc[v] = 0.f;
    }
    store_chan<xpt_half_t, G>((xpt_half_t*)m.dxin[u] + o, acc);
  }
}

// wbpj == 0: data gradient only (the single-layer entry point)
template <int S, int G>
__global__ __launch_bounds__(256) void dw_multi_bwd_tile_kernel(DwMultiBwd m, DwDims d, int relu_lab, int RG, int GRP,
                                                                int ndata, int cchunks, int wblocks_per_job, DwTile t) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smt[];
  const int relu_in = relu_lab & 255, lab = relu_lab >> 8;
  unsigned vb;                                  // (image-to-XCD numbering of both classes of workgroups, xpt_common.h)
  if (!xpt_xcd_two_class(t.xcd != 0, blockIdx.x, (unsigned)m.n_inputs, (unsigned)(ndata / m.n_inputs), (unsigned)m.n,
                         (unsigned)wblocks_per_job, vb))
    return;
  if (((int)vb < ndata) ? (lab & 1) : (lab & 2)) return;
  if ((int)vb < ndata) {
    dw_tile_bwd_block<S, G>(m, d, relu_in, t, vb, smt);
    return;
  }
  const int tt0 = (int)vb - ndata;
  const int job = tt0 / wblocks_per_job, tt = tt0 - job * wblocks_per_job;
  DwDims dj = d;
  dj.pad_t = m.pad_t[job];
  dj.pad_l = m.pad_l[job];
  typedef xpt_half_t T;
  const T* x = (const T*)m.xin[m.input_of[job]];
  const T* dy = (const T*)m.dy[job];
  float* fold = (float*)smt;
  if (m.k[job] == 3)
    dw_bwd_weight_body<T, 3, S, true>(x, dy, m.part[job], dj, relu_in, RG, GRP, tt % cchunks, tt / cchunks, fold);
  else if (m.k[job] == 5)
    dw_bwd_weight_body<T, 5, S, true>(x, dy, m.part[job], dj, relu_in, RG, GRP, tt % cchunks, tt / cchunks, fold);
  else
    dw_bwd_weight_body<T, 7, S, true>(x, dy, m.part[job], dj, relu_in, RG, GRP, tt % cchunks, tt / cchunks, fold);
}

// The same launch with the data-gradient part vectorised (stride 1): dx_u = [x_u > 0] * sum over the jobs j reading input u
// of conv(dy_j, w_j rotated by 180 degrees, pad' = k_j - 1 - pad_j); the rotated taps of those jobs sit in LDS.
template <typename T, int V>
__global__ __launch_bounds__(256, 4) void dw_multi_bwd_vec_kernel(DwMultiBwd m, DwDims d, int relu_lab, int RG, int GRP,
                                                               int data_blocks, int cchunks, int wblocks_per_job, int xcd) {
  extern __shared__ __attribute__((aligned(16))) float sW[];
  const int relu_in = relu_lab & 255, lab = relu_lab >> 8;      // lab knobs (xpt_dwconv_tune(-21 / -22)): 1 no data part, 2 no weight part
  // data_blocks < 0: the small-map data gradient (whole map of an image in LDS): -data_blocks = B * C / 8 workgroups per input
  const bool small = data_blocks < 0;
  if (small) data_blocks = -data_blocks;
  const int ndata = m.n_inputs * data_blocks;
  unsigned vb;                                  // (image-to-XCD numbering of both classes of workgroups, xpt_common.h)
  if (!xpt_xcd_two_class(xcd != 0, blockIdx.x, (unsigned)m.n_inputs, (unsigned)data_blocks, (unsigned)m.n, (unsigned)wblocks_per_job, vb))
    return;
  if (((int)vb < ndata) ? (lab & 1) : (lab & 2)) return;
  if constexpr (sizeof(T) == 2 && (V == 8 || V == 4)) {
    if (small && (int)vb < ndata) {
      // workgroup = (input u, image b, V-channel group): the maps dy_j of the jobs reading u and their rotated taps go to LDS
      // in ONE burst; outputs are summed over those jobs in job order, tap by tap as the stencil path does: same bits
      const int u = vb / data_blocks, rest = vb - u * data_blocks;
      const int g8 = d.C / V, b = rest / g8, c_lo = (rest - b * g8) * V;
      const int HW = d.OH * d.OW;                                         // stride 1: output map = input map
      unsigned char* sm = (unsigned char*)sW;
      const size_t map_bytes = ((size_t)HW * V * 2 + 15) & ~(size_t)15, taps_at = (size_t)DW_MAX_JOBS * map_bytes;
      int nj = 0;
      for (int j = 0; j < m.n; ++j) {
        if (m.input_of[j] != u) continue;
        const int kk = m.k[j] * m.k[j];
        dw_small_stage<V>((unsigned short*)(sm + nj * map_bytes), (float*)(sm + taps_at) + nj * 49 * V,
                          (const unsigned short*)m.dy[j], m.w[j], b, HW, d.C, c_lo, V, kk, true);
        ++nj;
      }
      __syncthreads();
      const unsigned short* x = (const unsigned short*)m.xin[u];
      for (int pix = threadIdx.x; pix < HW; pix += 256) {
        unsigned ix_;
        const int iy = (int)xpt_divmod((unsigned)pix, (unsigned)d.W, ix_), ix = (int)ix_;
        float acc[V];
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = 0.f;
        int q = 0;
        for (int j = 0; j < m.n; ++j) {
          if (m.input_of[j] != u) continue;
          const int k = m.k[j];
          dw_small_accumulate_k<V>(k, (const unsigned short*)(sm + q * map_bytes), (const float*)(sm + taps_at) + q * 49 * V, d.OH,
                                   d.OW, V, iy, ix, 0, k - 1 - m.pad_t[j], k - 1 - m.pad_l[j], false, acc);
          ++q;
        }
        const long long o = ((long long)b * HW + pix) * d.C + c_lo;
        if (relu_in) {
          float mk[V];
          load_chan<T, V>((const T*)x + o, mk);
#pragma unroll
          for (int v = 0; v < V; ++v)
            if (!(mk[v] > 0.f)) acc[v] = 0.f;
        }
        store_chan<T, V>((T*)m.dxin[u] + o, acc);
      }
      return;
    }
  }
  if ((int)vb < ndata) {
    const int u = vb / data_blocks, blk = vb - u * data_blocks;
    int off = 0;
    for (int j = 0; j < m.n; ++j) {
      if (m.input_of[j] != u) continue;
      dw_stage_taps(sW + off, m.w[j], d.C, m.k[j] * m.k[j], true);
      off += m.k[j] * m.k[j] * d.C;
    }
    __syncthreads();
    const T* x = (const T*)m.xin[u];
    T* dx = (T*)m.dxin[u];
    const int CG = d.C / V, IXG = (d.W + 1) / 2;
    const long long total = (long long)d.B * d.H * IXG * CG;
    long long idx, end;
    xpt_chunk_range((unsigned)blk, (unsigned)data_blocks, total, idx, end);
    for (idx += threadIdx.x; idx < end; idx += 256) {
      int c0, ix0, iy, b;
      xpt_split4((unsigned)idx, CG, IXG, d.H, c0, ix0, iy, b);
      c0 *= V;
      ix0 *= 2;
      float acc[2][V];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[i][v] = 0.f;
      int o2 = 0;
      for (int j = 0; j < m.n; ++j) {
        if (m.input_of[j] != u) continue;
        const int k = m.k[j];
        dw_vec_accumulate_k<T, V>(k, (const T*)m.dy[j], sW + o2, d.C, d.OH, d.OW, k - 1 - m.pad_t[j], k - 1 - m.pad_l[j], b,
                                  iy, ix0, c0, false, acc);
        o2 += k * k * d.C;
      }
      const long long o = (((long long)b * d.H + iy) * d.W + ix0) * d.C + c0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (ix0 + i >= d.W) break;
        if (relu_in) {
          float mk[V];
          load_chan<T, V>(x + o + (long long)i * d.C, mk);
#pragma unroll
          for (int v = 0; v < V; ++v)
            if (!(mk[v] > 0.f)) acc[i][v] = 0.f;
        }
        store_chan<T, V>(dx + o + (long long)i * d.C, acc[i]);
      }
    }
    return;
  }
  const int t = (int)vb - ndata;
  const int job = t / wblocks_per_job, tt = t - job * wblocks_per_job;
  DwDims dj = d;
  dj.pad_t = m.pad_t[job];
  dj.pad_l = m.pad_l[job];
  const T* x = (const T*)m.xin[m.input_of[job]];
  const T* dy = (const T*)m.dy[job];
  if (m.k[job] == 3)
    dw_bwd_weight_body<T, 3, 1, true>(x, dy, m.part[job], dj, relu_in, RG, GRP, tt % cchunks, tt / cchunks, sW);
  else if (m.k[job] == 5)
    dw_bwd_weight_body<T, 5, 1, true>(x, dy, m.part[job], dj, relu_in, RG, GRP, tt % cchunks, tt / cchunks, sW);
  else
    dw_bwd_weight_body<T, 7, 1, true>(x, dy, m.part[job], dj, relu_in, RG, GRP, tt % cchunks, tt / cchunks, sW);
}

// dw[i] = sum_k part[k][i]: 64 threads per output (few dependent round trips), fixed tree -> deterministic.
__global__ void dw_wrw_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int n, int nchunk) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int t = threadIdx.x & 63;
  float s = 0.f;
  if (i < n)
    for (int k = t; k < nchunk; k += 64) s += part[(long long)k * n + i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (t == 0 && i < n) dw[i] = s;
}

inline unsigned grid_for(long long total) {
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

int g_dw_small_cw = 8;      // channels per workgroup of the small-map forward (0: small-map kernels off; xpt_dwconv_tune(-100 - cw); in-step: 8 / 16 / 24 / 32 / 64 channels -> 5.46 / 5.50 / 5.52 / 5.55 / 5.72 ms, off 5.52)
int g_dw_small_max_px = 256;    // largest map (pixels) the small-map kernels take (xpt_dwconv_tune(-10000 - px)); 16x52 maps through them: 5.60 vs 5.41 ms/step
int g_dw_lab = 0;           // lab knobs of the vectorised multi-layer backward (xpt_dwconv_tune(-20 - bits))
int g_dw_multi_vec = 1;     // 0: the scalar multi-layer kernels (A/B, xpt_dwconv_tune(-1) / (-2))
int g_dw_s2_vec = 1;        // 0: the scalar stride-2 data gradient (A/B, xpt_dwconv_tune(-3) / (-4))

// channel vector width of the vectorised multi-layer kernels: 8 / 4 / 2 bf16 or 4 / 2 floats dividing C; 1 = none
inline int multi_vec_width(int dtype, int C, std::initializer_list<const void*>) {
  int v = dtype == 0 ? 4 : 8;
  while (v > 1 && C % v != 0) v >>= 1;
  return v;
}

// widest channel vector (elements) the tensors allow; 1 = use the scalar kernels
template <typename T>
inline int chan_vec(const DwDims& d, int K, std::initializer_list<const void*> ptrs) {
  if ((size_t)d.C * K * K * sizeof(float) > 48 * 1024) return 1;          // taps must fit the LDS staging buffer
  // Small maps (everything below the stem at batch 8) are launch/latency-bound: there the scalar kernels' 4-8x more
  // threads win over fewer, fatter ones (measured: 8x176x4x13 k5 7.9 vs 12.9 us; 8x32x64x208 k7 s2 29.4 vs 17.7 us).
  if ((long long)d.B * d.H * d.W * d.C < (1LL << 21)) return 1;
  int v = sizeof(T) == 2 ? 8 : 4;
  while (v > 1) {
    bool ok = d.C % v == 0;
    for (const void* p : ptrs) ok = ok && (p == nullptr || ((uintptr_t)p) % (v * sizeof(T)) == 0);
    if (ok) break;
    v >>= 1;
  }
  return v;
}

template <typename T, int K, int S, int OXT, int FLIP>
void launch_stencil(int v, const void* x, const float* w, const void* mask, void* y, const DwDims& d, int relu_in,
                    hipStream_t s) {
  const long long total = (long long)d.B * d.OH * ((d.OW + OXT - 1) / OXT) * (d.C / v);
  const size_t lds = (size_t)d.C * K * K * sizeof(float);
  const dim3 grid(grid_for(total));
#define XPT_STENCIL(V)                                                                                                \
  hipLaunchKernelGGL((dw_stencil_kernel<T, K, S, V, OXT, FLIP>), grid, dim3(256), lds, s, (const T*)x, w, (const T*)mask, \
                     (T*)y, d, relu_in)
  if (sizeof(T) == 2 && v == 8) {
    if constexpr (sizeof(T) == 2) XPT_STENCIL(8);
  } else if (v == 4) {
    XPT_STENCIL(4);
  } else {
    XPT_STENCIL(2);
  }
#undef XPT_STENCIL
}

bool tile_fwd_launch(const DwMultiFwd& m, int n, const DwDims& d, int relu_in, hipStream_t s, int stride = 2);
bool tile_bwd_launch(const DwMultiBwd& m, const DwDims& d, int relu_in, int RG, int GRP, int cchunks, int wbpj, hipStream_t s,
                     int stride = 2);

template <typename T, int K, int S>
int launch_fwd(const void* x, const float* w, void* y, const DwDims& d, int relu_in, hipStream_t s) {
  constexpr int OXT = (S == 1) ? 4 : 2;
  if (S == 2 && sizeof(T) == 2) {             // tiles in LDS
    DwMultiFwd m{};
    m.x[0] = x; m.w[0] = w; m.y[0] = y; m.k[0] = K; m.pad_t[0] = d.pad_t; m.pad_l[0] = d.pad_l;
    if (tile_fwd_launch(m, 1, d, relu_in, s)) return xpt_launch_status();
  }
  const int v = chan_vec<T>(d, K, {x, y});
  if (v > 1) {
    launch_stencil<T, K, S, OXT, 0>(v, x, w, nullptr, y, d, relu_in, s);
    return xpt_launch_status();
  }
  constexpr int OXS = OXT;
  const long long total = (long long)d.B * d.OH * ((d.OW + OXS - 1) / OXS) * d.C;
  const XcdSweep sw = xpt_xcd_sweep(total, 4096);
  hipLaunchKernelGGL((dw_fwd_kernel<T, K, S, OXS>), dim3(sw.grid), dim3(256), 0, s, (const T*)x, w, (T*)y, d, relu_in, sw);
  return xpt_launch_status();
}

template <typename T, int K, int S>
int launch_bwd_data(const void* x, const float* w, const void* dy, void* dx, const DwDims& d, int relu_in,
                    hipStream_t s) {
  if (S == 1) {
    // dx = conv(dy, rot180(w)) with pad' = K - 1 - pad: the forward stencil on (dy -> dx) with swapped extents
    const DwDims t{d.B, d.OH, d.OW, d.C, d.H, d.W, K - 1 - d.pad_t, K - 1 - d.pad_l};
    const int v = chan_vec<T>(t, K, {dy, dx, relu_in ? x : nullptr});
    if (v > 1 && t.pad_t >= 0 && t.pad_l >= 0) {
      launch_stencil<T, K, 1, 4, 1>(v, dy, w, x, dx, t, relu_in, s);
      return xpt_launch_status();
    }
  }
  if (S == 2 && sizeof(T) == 2) {             // tiles in LDS
    DwMultiBwd m{};
    m.n = 1; m.n_inputs = 1;
    m.xin[0] = x; m.dxin[0] = dx; m.dy[0] = dy; m.w[0] = w; m.k[0] = K; m.pad_t[0] = d.pad_t; m.pad_l[0] = d.pad_l;
    if (tile_bwd_launch(m, d, relu_in, 1, 1, 1, 0, s)) return xpt_launch_status();
  }
  if (S == 2 && g_dw_s2_vec) {
    const int v = chan_vec<T>(d, K, {dy, dx, relu_in ? x : nullptr});
    if (v > 1) {
      const long long groups = (long long)d.B * d.H * d.W * (d.C / v);
      const size_t lds = (size_t)K * K * d.C * sizeof(float);
      const dim3 grid(grid_for(groups));
#define XPT_S2V(V) \
  hipLaunchKernelGGL((dw_bwd_data_s2_vec_kernel<T, K, V>), grid, dim3(256), lds, s, (const T*)x, w, (const T*)dy, (T*)dx, d, relu_in)
      if constexpr (sizeof(T) == 2) {
        if (v == 8) XPT_S2V(8); else if (v == 4) XPT_S2V(4); else XPT_S2V(2);
      } else {
        if (v == 4) XPT_S2V(4); else XPT_S2V(2);
      }
#undef XPT_S2V
      return xpt_launch_status();
    }
  }
  const long long total = (long long)d.B * d.H * d.W * d.C;
  const int nblocks = (int)grid_for(total), xcd = g_xpt_xcd_affinity;
  hipLaunchKernelGGL((dw_bwd_data_kernel<T, K, S>), dim3(xcd ? xpt_xcd_pad(nblocks) : nblocks), dim3(256), 0, s, (const T*)x, w,
                     (const T*)dy, (T*)dx, d, relu_in, nblocks, xcd);
  return xpt_launch_status();
}

template <typename T, int K, int S>
int launch_bwd_weight(const void* x, const void* dy, float* dw, float* ws, const DwDims& d, int relu_in,
                      hipStream_t s) {
  constexpr int OXT = (S == 1) ? 4 : 2;
  const long long ngrp = (long long)d.B * d.OH * ((d.OW + OXT - 1) / OXT);
  const int GRP = wrw_groups(ngrp, d.C);
  const int nchunk = (int)((ngrp + GRP - 1) / GRP);
  const int RG = (d.C <= 32) ? (64 / d.C > GRP / 4 ? (GRP / 4 > 0 ? GRP / 4 : 1) : 64 / d.C) : 1;
  hipLaunchKernelGGL((dw_bwd_weight_kernel<T, K, S>), dim3((d.C + 63) / 64, nchunk), dim3(256), 0, s, (const T*)x,
                     (const T*)dy, ws, d, relu_in, RG, GRP);
  const int n = d.C * K * K;
  if (dw) hipLaunchKernelGGL(dw_wrw_reduce_kernel, dim3((n + 3) / 4), dim3(256), 0, s, ws, dw, n, nchunk);
  return xpt_launch_status();
}

template <typename T, int K, int S>
int launch_bwd_both(const void* x, const float* w, const void* dy, void* dx, float* ws, const DwDims& d, int relu_in,
                    hipStream_t s) {
  constexpr int OXT = (S == 1) ? 4 : 2;
  const long long ngrp = (long long)d.B * d.OH * ((d.OW + OXT - 1) / OXT);
  const int GRP = wrw_groups(ngrp, d.C);
  const int nchunk = (int)((ngrp + GRP - 1) / GRP);
  const int RG = (d.C <= 32) ? (64 / d.C > GRP / 4 ? (GRP / 4 > 0 ? GRP / 4 : 1) : 64 / d.C) : 1;
  const int cchunks = (d.C + 63) / 64;
  if (sizeof(T) == 2) {                       // data gradient on tiles in LDS, weight-gradient workgroups behind them
    DwMultiBwd m{};
    m.n = 1; m.n_inputs = 1;
    m.xin[0] = x; m.dxin[0] = dx; m.dy[0] = dy; m.w[0] = w; m.part[0] = ws; m.k[0] = K; m.pad_t[0] = d.pad_t; m.pad_l[0] = d.pad_l;
    if (tile_bwd_launch(m, d, relu_in, RG, GRP, cchunks, cchunks * nchunk, s, S)) return xpt_launch_status();
  }
  const int data_blocks = (int)grid_for((long long)d.B * d.H * d.W * d.C);
  const int xcd = g_xpt_xcd_affinity;
  hipLaunchKernelGGL((dw_bwd_both_kernel<T, K, S>), dim3(xpt_xcd_two_class_grid(xcd, 1, data_blocks, 1, cchunks * nchunk)), dim3(256), 0,
                     s, (const T*)x, w, (const T*)dy, (T*)dx, ws, d, relu_in, RG, GRP, data_blocks, cchunks, cchunks * nchunk, xcd);
  return xpt_launch_status();
}

// ---- tile kernels (stride 2, bf16): plan and launch; false = not served here (the caller falls through to the kernels above)
// Measured hot at batch 8 (tools/lab/dw_tile_probe.py), kernels above -> tiles.  Backward (dx + weight-gradient partials in one
// launch): stem 7x7 on 64x208x32 77 -> 29 us, 5x5 53 -> 21; the reduction cells' three layers 16x52x88 36 -> 15, 8x26x176
// 20 -> 9.5, 32x104x22 (2-channel groups, 8 x 16 tiles) 36 -> 20.  Forward: only the stem gains (18.2 -> 13.5, 11.2 -> 9.1 us);
// on the smaller maps the region of a stride-2 tile is four times its outputs and the old kernels win (7.9 vs 12.4 us).
int g_dw_tile_fwd = 1616;   // TH * 100 + TW of the forward's output tiles, 0 = off (xpt_dwconv_tune(-20000 - code))
int g_dw_tile_bwd = 1;      // TH * 100 + TW of the data gradient's dx tiles, 0 = off, 1 = 16 x 32 for 8-channel groups and
                            // 8 x 16 for narrower ones (xpt_dwconv_tune(-30000 - code))
long long g_dw_tile_fwd_min = 1LL << 21;   // forward: input elements from which the tiles are used
int g_dw_tile_min_group = 2; // narrowest channel group served (odd channel counts = 1: measured no gain, 23.2 -> 21.5 us on the
                            // stem's 11-channel 5x5; xpt_dwconv_tune(-80000 - g))
// Stride 1 on the 16 x 52 maps (above the small-map limit), measured: forward 7.1 -> 7.0 us, backward 12.0 -> 24.5: off.
int g_dw_tile_bwd1 = 0;     // stride-1 backward on maps above the small-map limit: tile code (1 = as g_dw_tile_bwd), 0 = off
                            // (xpt_dwconv_tune(-70000 - code))
int g_dw_tile_fwd1 = 0;     // stride-1 layers of one launch on maps above the small-map limit: tile code, 0 = off
                            // (xpt_dwconv_tune(-60000 - code))
int g_dw_tile_cw = 0;       // channels per workgroup; 0 = 8 for 8-channel groups, all channels for narrower ones (xpt_dwconv_tune(-40000 - cw))

inline int tile_group(int C, std::initializer_list<const void*> ptrs) {
  int g = C % 8 == 0 ? 8 : (C % 4 == 0 ? 4 : (C % 2 == 0 ? 2 : 1));
  for (const void* p : ptrs)
    while (g > 1 && ((uintptr_t)p) % (2 * g)) g >>= 1;
  return g;
}

inline void tile_geometry(int code, int rows, int cols, int C, int G, DwTile& t) {
  t.TH = code / 100 < rows ? code / 100 : rows;
  t.TW = code % 100 < cols ? code % 100 : cols;
  t.tiles_x = (cols + t.TW - 1) / t.TW;
  t.tiles = t.tiles_x * ((rows + t.TH - 1) / t.TH);
  int cw = g_dw_tile_cw > 0 ? g_dw_tile_cw / G * G : (G == 8 ? 8 : C);      // 0: 8 channels, all of them for narrower groups
  if (cw < G) cw = G;
  if (cw > C) cw = C;
  t.CW = cw;
  t.chunks = (C + cw - 1) / cw;
}

bool tile_fwd_launch(const DwMultiFwd& m, int n, const DwDims& d, int relu_in, hipStream_t s, int stride) {
  const int code = stride == 2 ? g_dw_tile_fwd : g_dw_tile_fwd1;
  if (code <= 0) return false;
  const long long elems = (long long)d.B * d.H * d.W * d.C;
  if (stride == 2 ? elems < g_dw_tile_fwd_min : (d.H * d.W <= g_dw_small_max_px)) return false;
  int G = tile_group(d.C, {});
  int kmax = 0;
  for (int j = 0; j < n && G; ++j) {
    const int gj = tile_group(d.C, {m.x[j], m.y[j]});
    G = gj < G ? gj : G;
    kmax = m.k[j] > kmax ? m.k[j] : kmax;
  }
  if (G < g_dw_tile_min_group) return false;
  DwTile t{};
  tile_geometry(code, d.OH, d.OW, d.C, G, t);
  const size_t region = (size_t)((t.TH - 1) * stride + kmax) * ((t.TW - 1) * stride + kmax) * t.CW * 2;
  t.map_bytes = (int)((region + 15) & ~(size_t)15);
  const size_t lds = (size_t)t.map_bytes + (size_t)kmax * kmax * t.CW * sizeof(float);
  if (lds > 64 * 1024 || (long long)n * t.tiles > 65535 * 32LL || d.B > 65535 || t.chunks > 65535) return false;
  t.xcd = g_xpt_xcd_affinity;
  const dim3 grid(t.xcd ? xpt_xcd_pad((unsigned long long)d.B * t.tiles) : d.B * t.tiles, n, t.chunks);
  XPT_BEGIN_LAUNCH();
#define XPT_TF(S_, G_) hipLaunchKernelGGL((dw_tile_fwd_kernel<S_, G_>), grid, dim3(256), lds, s, m, d, relu_in, t)
  if (stride == 2) {
    if (G == 8) XPT_TF(2, 8); else if (G == 4) XPT_TF(2, 4); else if (G == 2) XPT_TF(2, 2); else XPT_TF(2, 1);
  } else {
    if (G == 8) XPT_TF(1, 8); else if (G == 4) XPT_TF(1, 4); else if (G == 2) XPT_TF(1, 2); else XPT_TF(1, 1);
  }
#undef XPT_TF
  return true;
}

// wbpj = 0: the data gradient alone
bool tile_bwd_launch(const DwMultiBwd& m, const DwDims& d, int relu_in, int RG, int GRP, int cchunks, int wbpj, hipStream_t s,
                     int stride) {
  const int code = stride == 2 ? g_dw_tile_bwd : g_dw_tile_bwd1;
  if (code <= 0 || (stride == 1 && d.H * d.W <= g_dw_small_max_px)) return false;
  int G = tile_group(d.C, {});
  for (int u = 0; u < m.n_inputs && G; ++u) {
    const int gu = tile_group(d.C, {m.xin[u], m.dxin[u]});
    G = gu < G ? gu : G;
  }
  int kmax = 0, slots = 0;
  size_t fold = 0;
  for (int j = 0; j < m.n && G; ++j) {
    const int gj = tile_group(d.C, {m.dy[j]});
    G = gj < G ? gj : G;
    kmax = m.k[j] > kmax ? m.k[j] : kmax;
    const size_t f = (size_t)3 * 64 * m.k[j] * m.k[j] * sizeof(float);
    fold = f > fold ? f : fold;
    int same = 0;
    for (int i = 0; i < m.n; ++i) same += m.input_of[i] == m.input_of[j];
    slots = same > slots ? same : slots;
  }
  if (G < g_dw_tile_min_group) return false;
  DwTile t{};
  tile_geometry(code == 1 ? (G == 8 ? 1632 : 816) : code, d.H, d.W, d.C, G, t);
  const int KP = stride == 2 ? kmax + 1 : kmax;
  const size_t region = stride == 2 ? (size_t)(t.TH / 2 + KP / 2 + 2) * (t.TW / 2 + KP / 2 + 2) * t.CW * 2
                                    : (size_t)(t.TH + KP - 1) * (t.TW + KP - 1) * t.CW * 2;
  t.map_bytes = (int)((region + 15) & ~(size_t)15);
  t.slot_bytes = t.map_bytes + (int)(((size_t)KP * KP * t.CW * sizeof(float) + 15) & ~(size_t)15);
  size_t lds = (size_t)slots * t.slot_bytes;
  if (wbpj > 0 && fold > lds) lds = fold;
  const long long ndata = (long long)m.n_inputs * d.B * t.tiles * t.chunks;
  if (lds > 64 * 1024 || ndata + (long long)m.n * wbpj > 0x7fffffffLL) return false;
  t.xcd = g_xpt_xcd_affinity;
  const dim3 grid(xpt_xcd_two_class_grid(t.xcd, m.n_inputs, (unsigned)(ndata / m.n_inputs), wbpj > 0 ? m.n : 0, wbpj > 0 ? wbpj : 0));
  XPT_BEGIN_LAUNCH();
#define XPT_TB(S_, G_) \
  hipLaunchKernelGGL((dw_multi_bwd_tile_kernel<S_, G_>), grid, dim3(256), lds, s, m, d, relu_in | (g_dw_lab << 8), RG, GRP, (int)ndata, \
                     cchunks, wbpj > 0 ? wbpj : 1, t)
  if (stride == 2) {
    if (G == 8) XPT_TB(2, 8); else if (G == 4) XPT_TB(2, 4); else if (G == 2) XPT_TB(2, 2); else XPT_TB(2, 1);
  } else {
    if (G == 8) XPT_TB(1, 8); else if (G == 4) XPT_TB(1, 4); else if (G == 2) XPT_TB(1, 2); else XPT_TB(1, 1);
  }
#undef XPT_TB
  return true;
}

#define DW_DISPATCH(FN, ...)                                                   \
  do {                                                                         \
    if (dtype == 0) {                                                          \
      if (k == 3 && stride == 1) return FN<float, 3, 1>(__VA_ARGS__);          \
      if (k == 5 && stride == 1) return FN<float, 5, 1>(__VA_ARGS__);          \
      if (k == 7 && stride == 1) return FN<float, 7, 1>(__VA_ARGS__);          \
      if (k == 3 && stride == 2) return FN<float, 3, 2>(__VA_ARGS__);          \
      if (k == 5 && stride == 2) return FN<float, 5, 2>(__VA_ARGS__);          \
      if (k == 7 && stride == 2) return FN<float, 7, 2>(__VA_ARGS__);          \
    } else {                                                                   \
      if (k == 3 && stride == 1) return FN<xpt_half_t, 3, 1>(__VA_ARGS__); \
      if (k == 5 && stride == 1) return FN<xpt_half_t, 5, 1>(__VA_ARGS__); \
      if (k == 7 && stride == 1) return FN<xpt_half_t, 7, 1>(__VA_ARGS__); \
      if (k == 3 && stride == 2) return FN<xpt_half_t, 3, 2>(__VA_ARGS__); \
      if (k == 5 && stride == 2) return FN<xpt_half_t, 5, 2>(__VA_ARGS__); \
      if (k == 7 && stride == 2) return FN<xpt_half_t, 7, 2>(__VA_ARGS__); \
    }                                                                          \
    return XPT_ERR_ARG;                                                        \
  } while (0)

int check_dims(int B, int H, int W, int C, int k, int stride, int pad_t, int pad_l, int OH, int OW, int dtype) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || pad_t < 0 || pad_l < 0) return XPT_ERR_SHAPE;
  if ((k != 3 && k != 5 && k != 7) || (stride != 1 && stride != 2) || (dtype != 0 && dtype != 1)) return XPT_ERR_ARG;
  // (the kernels split flat element indices in 32 bits)
  if ((long long)B * H * W * C >= (1LL << 31) || (long long)B * OH * OW * C >= (1LL << 31)) return XPT_ERR_SHAPE;
  // every output window must start inside the padded input
  if ((OH - 1) * stride - pad_t >= H || (OW - 1) * stride - pad_l >= W) return XPT_ERR_SHAPE;
  return XPT_OK;
}

}  // namespace

extern "C" {

int xpt_dwconv_fwd(const void* x, const float* w, void* y, int B, int H, int W, int C, int k, int stride, int pad_t,
                   int pad_l, int OH, int OW, int relu_in, int dtype, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(w); XPT_CHECK_PTR(y);
  const int rc = check_dims(B, H, W, C, k, stride, pad_t, pad_l, OH, OW, dtype);
  if (rc != XPT_OK) return rc;
  const DwDims d{B, H, W, C, OH, OW, pad_t, pad_l};
  XPT_BEGIN_LAUNCH();
  DW_DISPATCH(launch_fwd, x, w, y, d, relu_in, (hipStream_t)stream);
}

int xpt_dwconv_bwd_data(const void* x, const float* w, const void* dy, void* dx, int B, int H, int W, int C, int k,
                        int stride, int pad_t, int pad_l, int OH, int OW, int relu_in, int dtype, void* stream) {
  XPT_CHECK_PTR(w); XPT_CHECK_PTR(dy); XPT_CHECK_PTR(dx);
  if (relu_in) XPT_CHECK_PTR(x);
  const int rc = check_dims(B, H, W, C, k, stride, pad_t, pad_l, OH, OW, dtype);
  if (rc != XPT_OK) return rc;
  const DwDims d{B, H, W, C, OH, OW, pad_t, pad_l};
  XPT_BEGIN_LAUNCH();
  DW_DISPATCH(launch_bwd_data, x, w, dy, dx, d, relu_in, (hipStream_t)stream);
}

size_t xpt_dwconv_bwd_weight_workspace_floats(int B, int OH, int OW, int C, int k) {
  if (B <= 0 || OH <= 0 || OW <= 0 || C <= 0 || k <= 0) return 0;
  // groups of OXT (>= 2) neighbouring outputs per row, g_dw_wrw_grp groups per workgroup: upper bound with OXT = 2
  const size_t ngrp = (size_t)B * OH * (((size_t)OW + 1) / 2);
  return ((ngrp + 3) / 4) * (size_t)C * k * k;
}

int xpt_dwconv_bwd_weight(const void* x, const void* dy, float* dw, float* workspace, size_t workspace_floats, int B,
                          int H, int W, int C, int k, int stride, int pad_t, int pad_l, int OH, int OW, int relu_in,
                          int dtype, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(dy); XPT_CHECK_PTR(dw); XPT_CHECK_PTR(workspace);
  const int rc = check_dims(B, H, W, C, k, stride, pad_t, pad_l, OH, OW, dtype);
  if (rc != XPT_OK) return rc;
  if (workspace_floats < xpt_dwconv_bwd_weight_workspace_floats(B, OH, OW, C, k)) return XPT_ERR_WORKSPACE;
  const DwDims d{B, H, W, C, OH, OW, pad_t, pad_l};
  XPT_BEGIN_LAUNCH();
  DW_DISPATCH(launch_bwd_weight, x, dy, dw, workspace, d, relu_in, (hipStream_t)stream);
}

/* launch-plan knob (process-wide, for benchmarking): groups of 4 (stride 1) / 2 (stride 2) outputs per workgroup of the
 * weight-gradient kernel; 4, 8, 16 or 32, 0 = automatic */
int xpt_dwconv_tune(int wrw_groups) {
  if (wrw_groups == -1 || wrw_groups == -2) {        // -1: scalar multi-layer kernels, -2: vectorised (default)
    g_dw_multi_vec = wrw_groups == -2;
    return XPT_OK;
  }
  if (wrw_groups <= -80000) {                        // tile kernels: narrowest channel group served (1, 2, 4, 8)
    g_dw_tile_min_group = -80000 - wrw_groups;
    return XPT_OK;
  }
  if (wrw_groups <= -70000) {                        // tile kernels: tile code of the stride-1 backward (0 = off)
    g_dw_tile_bwd1 = -70000 - wrw_groups;
    return XPT_OK;
  }
  if (wrw_groups <= -60000) {                        // tile kernels: tile code of the stride-1 multi-layer forward (0 = off)
    g_dw_tile_fwd1 = -60000 - wrw_groups;
    return XPT_OK;
  }
  if (wrw_groups <= -50000) {                        // tile kernels: forward from 2^n input elements on
    g_dw_tile_fwd_min = 1LL << (-50000 - wrw_groups);
    return XPT_OK;
  }
  if (wrw_groups <= -40000) {                        // tile kernels: channels per workgroup
    g_dw_tile_cw = -40000 - wrw_groups;
    return XPT_OK;
  }
  if (wrw_groups <= -30000) {                        // tile kernels: TH * 100 + TW of the data gradient's tiles (0 = off)
    g_dw_tile_bwd = -30000 - wrw_groups;
    return XPT_OK;
  }
  if (wrw_groups <= -20000) {                        // tile kernels: TH * 100 + TW of the forward's tiles (0 = off)
    g_dw_tile_fwd = -20000 - wrw_groups;
    return XPT_OK;
  }
  if (wrw_groups <= -10000) {                        // largest map (pixels) of the small-map kernels
    g_dw_small_max_px = -10000 - wrw_groups;
    return XPT_OK;
  }
  if (wrw_groups <= -100 && wrw_groups >= -356) {    // channels per workgroup of the small-map kernels: -100 off, -132 / -164 ...
    g_dw_small_cw = (-100 - wrw_groups) / 8 * 8;
    return XPT_OK;
  }
  if (wrw_groups <= -20 && wrw_groups >= -23) {      // lab: -21 no data-gradient part, -22 no weight-gradient part, -20 off
    g_dw_lab = -20 - wrw_groups;
    return XPT_OK;
  }
  if (wrw_groups == -3 || wrw_groups == -4) {        // -3: scalar stride-2 data gradient, -4: vectorised (default)
    g_dw_s2_vec = wrw_groups == -4;
    return XPT_OK;
  }
  if (wrw_groups != 0 && wrw_groups != 4 && wrw_groups != 8 && wrw_groups != 16 && wrw_groups != 32) return XPT_ERR_ARG;
  g_dw_wrw_grp = wrw_groups;
  return XPT_OK;
}

int xpt_dwconv_bwd_weight_chunks(int B, int OH, int OW, int C, int k, int stride) {
  if (B <= 0 || OH <= 0 || OW <= 0 || C <= 0 || k <= 0 || (stride != 1 && stride != 2)) return 0;
  const int oxt = stride == 1 ? 4 : 2;
  const long long ngrp = (long long)B * OH * ((OW + oxt - 1) / oxt);
  const int grp = wrw_groups(ngrp, C);
  return (int)((ngrp + grp - 1) / grp);
}

/* Deferred weight gradient: partials[chunk][C][k][k], to be added up later by xpt_reduce_partials. */
int xpt_dwconv_bwd_weight_partials(const void* x, const void* dy, float* partials, size_t partial_floats, int B, int H,
                                   int W, int C, int k, int stride, int pad_t, int pad_l, int OH, int OW, int relu_in,
                                   int dtype, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(dy); XPT_CHECK_PTR(partials);
  const int rc = check_dims(B, H, W, C, k, stride, pad_t, pad_l, OH, OW, dtype);
  if (rc != XPT_OK) return rc;
  if (partial_floats < (size_t)xpt_dwconv_bwd_weight_chunks(B, OH, OW, C, k, stride) * C * k * k) return XPT_ERR_WORKSPACE;
  const DwDims d{B, H, W, C, OH, OW, pad_t, pad_l};
  XPT_BEGIN_LAUNCH();
  DW_DISPATCH(launch_bwd_weight, x, dy, (float*)nullptr, partials, d, relu_in, (hipStream_t)stream);
}

/* dx (as xpt_dwconv_bwd_data) and the weight-gradient partials (as xpt_dwconv_bwd_weight_partials) in one launch. */
int xpt_dwconv_bwd_both(const void* x, const float* w, const void* dy, void* dx, float* partials, size_t partial_floats,
                        int B, int H, int W, int C, int k, int stride, int pad_t, int pad_l, int OH, int OW, int relu_in,
                        int dtype, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(w); XPT_CHECK_PTR(dy); XPT_CHECK_PTR(dx); XPT_CHECK_PTR(partials);
  const int rc = check_dims(B, H, W, C, k, stride, pad_t, pad_l, OH, OW, dtype);
  if (rc != XPT_OK) return rc;
  if (partial_floats < (size_t)xpt_dwconv_bwd_weight_chunks(B, OH, OW, C, k, stride) * C * k * k) return XPT_ERR_WORKSPACE;
  const DwDims d{B, H, W, C, OH, OW, pad_t, pad_l};
  XPT_BEGIN_LAUNCH();
  DW_DISPATCH(launch_bwd_both, x, w, dy, dx, partials, d, relu_in, (hipStream_t)stream);
}

/* n (<= 6) depthwise layers of one activation shape and stride in one launch: y[j] = dwconv(f(x[j]), w[j]), kernel
 * size k[j] in {3, 5, 7}, leading padding (pad_t[j], pad_l[j]); inputs may repeat. */
int xpt_dwconv_multi_fwd(const void* const* x, const float* const* w, void* const* y, const int* k, const int* pad_t,
                         const int* pad_l, int n, int B, int H, int W, int C, int stride, int OH, int OW, int relu_in,
                         int dtype, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(w); XPT_CHECK_PTR(y); XPT_CHECK_PTR(k); XPT_CHECK_PTR(pad_t); XPT_CHECK_PTR(pad_l);
  if (n < 1 || n > DW_MAX_JOBS) return XPT_ERR_ARG;
  DwMultiFwd m{};
  for (int j = 0; j < n; ++j) {
    if (!x[j] || !w[j] || !y[j]) return XPT_ERR_NULL;
    const int rc = check_dims(B, H, W, C, k[j], stride, pad_t[j], pad_l[j], OH, OW, dtype);
    if (rc != XPT_OK) return rc;
    m.x[j] = x[j]; m.w[j] = w[j]; m.y[j] = y[j]; m.k[j] = k[j]; m.pad_t[j] = pad_t[j]; m.pad_l[j] = pad_l[j];
  }
  const DwDims d{B, H, W, C, OH, OW, 0, 0};
  hipStream_t s = (hipStream_t)stream;
  if (stride == 1 && dtype == 1 && g_dw_small_cw > 0 && C % 4 == 0 && H * W <= g_dw_small_max_px && OH == H && OW == W) {
    // small maps: the whole map of an image in LDS, one round trip per workgroup, G = 8 (or 4: the 44-channel stack)
    // channels per workgroup
    const int G = C % 8 == 0 ? 8 : 4;
    bool ok = true;
    int kmax = 0;
    for (int j = 0; j < n; ++j) {
      ok = ok && ((uintptr_t)x[j]) % (2 * G) == 0 && ((uintptr_t)y[j]) % (2 * G) == 0;
      kmax = k[j] > kmax ? k[j] : kmax;
    }
    int cw = g_dw_small_cw < C ? g_dw_small_cw : C;
    cw = cw / G * G;
    if (cw < G) cw = G;
    const size_t lds = (((size_t)H * W * cw * 2 + 15) & ~(size_t)15) + (size_t)kmax * kmax * cw * 4;
    if (ok && lds <= 64 * 1024) {
      XPT_BEGIN_LAUNCH();
      if (G == 8)
        hipLaunchKernelGGL(dw_small_fwd_kernel<8>, dim3(n, B, (C + cw - 1) / cw), dim3(256), lds, s, m, d, relu_in, cw);
      else
        hipLaunchKernelGGL(dw_small_fwd_kernel<4>, dim3(n, B, (C + cw - 1) / cw), dim3(256), lds, s, m, d, relu_in, cw);
      return xpt_launch_status();
    }
  }
  if (dtype == 1 && tile_fwd_launch(m, n, d, relu_in, s, stride)) return xpt_launch_status();      // tiles in LDS
  if (stride == 1 && g_dw_multi_vec) {      // vectorised path: V channels per thread, taps in LDS
    int kmax = 0;
    std::initializer_list<const void*> none{};
    int v = multi_vec_width(dtype, C, none);
    for (int j = 0; j < n && v > 1; ++j) {
      kmax = k[j] > kmax ? k[j] : kmax;
      const size_t a = (size_t)v * (dtype == 0 ? 4 : 2);
      if (((uintptr_t)x[j]) % a || ((uintptr_t)y[j]) % a) v = 1;
    }
    const size_t lds = (size_t)kmax * kmax * C * sizeof(float);
    if (v > 1 && lds <= 64 * 1024) {
      const long long totalv = (long long)B * OH * ((OW + 1) / 2) * (C / v);
      int bpjv = (int)grid_for(totalv);
      const int xcd = g_xpt_xcd_affinity;
      const int per = (int)(((totalv + bpjv - 1) / bpjv + 255) / 256 * 256);     // consecutive outputs per workgroup
      bpjv = (int)((totalv + per - 1) / per);
      XPT_BEGIN_LAUNCH();
#define XPT_MV(T, V) \
  hipLaunchKernelGGL((dw_multi_fwd_vec_kernel<T, V>), dim3(xcd ? xpt_xcd_pad(bpjv) : bpjv, n), dim3(256), lds, s, m, d, relu_in, bpjv, per, xcd)
      if (dtype == 0) { if (v == 4) XPT_MV(float, 4); else XPT_MV(float, 2); }
      else { if (v == 8) XPT_MV(xpt_half_t, 8); else if (v == 4) XPT_MV(xpt_half_t, 4); else XPT_MV(xpt_half_t, 2); }
#undef XPT_MV
      return xpt_launch_status();
    }
  }
  const int oxt = stride == 1 ? 4 : 2;
  const XcdSweep sw = xpt_xcd_sweep((long long)B * OH * ((OW + oxt - 1) / oxt) * C, 4096);
  XPT_BEGIN_LAUNCH();
#define XPT_MULTI(T, S) hipLaunchKernelGGL((dw_multi_fwd_kernel<T, S>), dim3(sw.grid, n), dim3(256), 0, s, m, d, relu_in, sw)
  if (dtype == 0) {
    if (stride == 1) XPT_MULTI(float, 1); else XPT_MULTI(float, 2);
  } else {
    if (stride == 1) XPT_MULTI(xpt_half_t, 1); else XPT_MULTI(xpt_half_t, 2);
  }
#undef XPT_MULTI
  return xpt_launch_status();
}

/* Backward of xpt_dwconv_multi_fwd in one launch.  xin / dxin: the n_inputs distinct inputs and their gradients (each
 * the sum over the jobs reading it); per job j: dy[j], w[j], k[j], pads, input_of[j] (index into xin) and partials[j]
 * (xpt_dwconv_bwd_weight_chunks(B, OH, OW, C, k[j], stride) * C * k[j]^2 floats, finished by xpt_reduce_partials). */
int xpt_dwconv_multi_bwd(const void* const* xin, void* const* dxin, int n_inputs, const void* const* dy,
                         const float* const* w, float* const* partials, const int* k, const int* pad_t,
                         const int* pad_l, const int* input_of, int n, int B, int H, int W, int C, int stride, int OH,
                         int OW, int relu_in, int dtype, void* stream) {
  XPT_CHECK_PTR(xin); XPT_CHECK_PTR(dxin); XPT_CHECK_PTR(dy); XPT_CHECK_PTR(w); XPT_CHECK_PTR(partials);
  XPT_CHECK_PTR(k); XPT_CHECK_PTR(pad_t); XPT_CHECK_PTR(pad_l); XPT_CHECK_PTR(input_of);
  if (n < 1 || n > DW_MAX_JOBS || n_inputs < 1 || n_inputs > n) return XPT_ERR_ARG;
  DwMultiBwd m{};
  m.n = n; m.n_inputs = n_inputs;
  for (int u = 0; u < n_inputs; ++u) {
    if (!xin[u] || !dxin[u]) return XPT_ERR_NULL;
    m.xin[u] = xin[u]; m.dxin[u] = dxin[u];
  }
  for (int j = 0; j < n; ++j) {
    if (!dy[j] || !w[j] || !partials[j]) return XPT_ERR_NULL;
    const int rc = check_dims(B, H, W, C, k[j], stride, pad_t[j], pad_l[j], OH, OW, dtype);
    if (rc != XPT_OK) return rc;
    if (input_of[j] < 0 || input_of[j] >= n_inputs) return XPT_ERR_ARG;
    m.dy[j] = dy[j]; m.w[j] = w[j]; m.part[j] = partials[j]; m.k[j] = k[j]; m.pad_t[j] = pad_t[j]; m.pad_l[j] = pad_l[j];
    m.input_of[j] = input_of[j];
  }
  const DwDims d{B, H, W, C, OH, OW, 0, 0};
  const int oxt = stride == 1 ? 4 : 2;
  const long long ngrp = (long long)B * OH * ((OW + oxt - 1) / oxt);
  const int GRP = wrw_groups(ngrp, C);
  const int nchunk = (int)((ngrp + GRP - 1) / GRP);
  const int RG = (C <= 32) ? (64 / C > GRP / 4 ? (GRP / 4 > 0 ? GRP / 4 : 1) : 64 / C) : 1;
  const int cchunks = (C + 63) / 64;
  const int wbpj = cchunks * nchunk;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == 1 && tile_bwd_launch(m, d, relu_in, RG, GRP, cchunks, wbpj, s, stride)) return xpt_launch_status();
  if (stride == 1 && g_dw_multi_vec) {
    std::initializer_list<const void*> none{};
    int v = multi_vec_width(dtype, C, none);
    const size_t a = (size_t)v * (dtype == 0 ? 4 : 2);
    for (int u = 0; u < n_inputs && v > 1; ++u)
      if (((uintptr_t)xin[u]) % a || ((uintptr_t)dxin[u]) % a) v = 1;
    size_t lds = 0;
    for (int u = 0; u < n_inputs; ++u) {
      size_t l = 0;
      for (int j = 0; j < n; ++j)
        if (input_of[j] == u) l += (size_t)k[j] * k[j] * C * sizeof(float);
      lds = l > lds ? l : lds;
    }
    for (int j = 0; j < n; ++j) {
      if (((uintptr_t)dy[j]) % a) v = 1;
      const size_t fold = (size_t)3 * 64 * k[j] * k[j] * sizeof(float);      // the weight-gradient workgroups' fold buffer
      lds = fold > lds ? fold : lds;
    }
    if (v > 1 && lds <= 64 * 1024) {
      int dbv = (int)grid_for((long long)B * H * ((W + 1) / 2) * (C / v));
      // small maps (the 8 x 26 / 4 x 13 stacks), 16-byte channel groups: the data-gradient part stages whole maps in LDS
      // (one round trip per workgroup instead of one per filter row of every job); dbv < 0 tells the kernel
      if (g_dw_small_cw > 0 && dtype == 1 && (v == 8 || v == 4) && H * W <= g_dw_small_max_px && OH == H && OW == W) {
        const size_t map_bytes = ((size_t)H * W * v * 2 + 15) & ~(size_t)15;
        const size_t small_lds = (size_t)DW_MAX_JOBS * map_bytes + (size_t)DW_MAX_JOBS * 49 * v * sizeof(float);
        if (small_lds <= 64 * 1024) {
          lds = small_lds > lds ? small_lds : lds;
          dbv = -(B * (C / v));
        }
      }
      const int xcd = g_xpt_xcd_affinity;
      const dim3 gridv(xpt_xcd_two_class_grid(xcd, n_inputs, dbv < 0 ? -dbv : dbv, n, wbpj));
      XPT_BEGIN_LAUNCH();
#define XPT_MV(T, V) \
  hipLaunchKernelGGL((dw_multi_bwd_vec_kernel<T, V>), gridv, dim3(256), lds, s, m, d, relu_in | (g_dw_lab << 8), RG, GRP, dbv, cchunks, wbpj, xcd)
      if (dtype == 0) { if (v == 4) XPT_MV(float, 4); else XPT_MV(float, 2); }
      else { if (v == 8) XPT_MV(xpt_half_t, 8); else if (v == 4) XPT_MV(xpt_half_t, 4); else XPT_MV(xpt_half_t, 2); }
#undef XPT_MV
      return xpt_launch_status();
    }
  }
  const int data_blocks = (int)grid_for((long long)B * H * W * C);
  const int xcd = g_xpt_xcd_affinity;
  const dim3 grid(xpt_xcd_two_class_grid(xcd, n_inputs, data_blocks, n, wbpj));
  size_t fold_bytes = 0;
  for (int j = 0; j < n; ++j) {
    const size_t f = (size_t)3 * 64 * k[j] * k[j] * sizeof(float);
    fold_bytes = f > fold_bytes ? f : fold_bytes;
  }
  XPT_BEGIN_LAUNCH();
#define XPT_MULTI(T, S) \
  hipLaunchKernelGGL((dw_multi_bwd_kernel<T, S>), grid, dim3(256), fold_bytes, s, m, d, relu_in, RG, GRP, data_blocks, cchunks, wbpj, xcd)
  if (dtype == 0) {
    if (stride == 1) XPT_MULTI(float, 1); else XPT_MULTI(float, 2);
  } else {
    if (stride == 1) XPT_MULTI(xpt_half_t, 1); else XPT_MULTI(xpt_half_t, 2);
  }
#undef XPT_MULTI
  return xpt_launch_status();
}

}  // extern "C"
