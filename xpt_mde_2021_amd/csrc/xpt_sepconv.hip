// xpt_sepconv.hip -- one branch stage of a NASNet-A cell as ONE launch: ReLU -> depthwise k x k (stride 1, SAME) ->
// pointwise 1x1 -> BatchNorm (inference statistics) [+ the sibling branch of the cell's `add`, + a residual].
//
// keras nasnet._separable_conv_block (tensorflow.keras.applications, instantiated by the reference at
// model/build_model/pretrained_nets.py:36-44) is, per half: Activation('relu') -> SeparableConv2D -> BatchNormalization;
// _normal_a_cell adds the two branches of a block.  The unfused path runs the depthwise halves of a cell stage as one
// launch (xpt_dwconv_multi_fwd) and the pointwise + BatchNorm halves as one or two more (xpt_pwconv_bn_multi_fwd: the
// left branches wait for the right ones, whose outputs they add) -- at batch 8 each of those launches is 5-10 us of
// mostly launch latency on 0.3-1.2 MB tensors.  Here a workgroup owns 32 output pixels of one branch (pair):
//   phase 1: the 256 threads compute the depthwise outputs of those pixels for all channels (V channels per item, the
//            taps staged in LDS as [tap][channel], exactly the arithmetic of dw_multi_fwd_vec_kernel), round them to
//            bf16, keep them in LDS as the MFMA A operand AND store them (the pointwise weight gradient needs them);
//   phase 2: every wave takes 32-column output tiles: A fragments from LDS, filter rows straight from global memory,
//            v_mfma_f32_32x32x16_bf16, then the BatchNorm epilogue of xpt_pwconv.hip (ypre kept for the backward).
// A job may carry a second branch (b): y = BN_a(conv_a) + bf16(BN_b(conv_b)) [+ residual] -- bit for bit what the two
// unfused launches produce (the right branch's output is rounded to bf16 before it is added).
#include "xpt_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef xpt_h16x8 bf16x8;      // (8 operands of the build's 16-bit format, xpt_common.h)

__device__ inline float bf16_bits_to_f32(unsigned short u) { return xpt_h2f(u); }
__device__ inline unsigned short f32_to_bf16_bits(float f) {   // round to nearest even
  return xpt_f2h_sw(f);
}

template <int V> struct BfVec;
template <> struct BfVec<8> { typedef uint4 type; };
template <> struct BfVec<4> { typedef uint2 type; };
template <> struct BfVec<2> { typedef unsigned type; };
template <> struct BfVec<1> { typedef unsigned short type; };

template <int V>
__device__ inline void load_bf(const unsigned short* p, float (&out)[V]) {
  typename BfVec<V>::type raw = *(const typename BfVec<V>::type*)p;
  const unsigned short* e = (const unsigned short*)&raw;
#pragma unroll
  for (int i = 0; i < V; ++i) out[i] = bf16_bits_to_f32(e[i]);
}

// bf16 pairs of a loaded vector -> floats, by shifts on the 32-bit components (no pointer into the register array: an
// address-taken local array lands in scratch memory)
__device__ inline void unpack2(unsigned w, float& lo, float& hi) { lo = xpt_h2f_lo(w); hi = xpt_h2f_hi(w); }
__device__ inline void unpack(const uint4& r, float (&o)[8]) { unpack2(r.x, o[0], o[1]); unpack2(r.y, o[2], o[3]); unpack2(r.z, o[4], o[5]); unpack2(r.w, o[6], o[7]); }
__device__ inline void unpack(const uint2& r, float (&o)[4]) { unpack2(r.x, o[0], o[1]); unpack2(r.y, o[2], o[3]); }
__device__ inline void unpack(const unsigned& r, float (&o)[2]) { unpack2(r, o[0], o[1]); }

// 8 consecutive bf16 of one filter row starting at element k; elements at or past K read as zero (xpt_pwconv.hip load_frag)
template <int V>
__device__ inline uint4 load_frag(const unsigned short* __restrict__ row, int k, int K) {
  if constexpr (V == 8) {
    return *(const uint4*)(row + k);
  } else {
    typedef typename BfVec<V>::type vec_t;
    unsigned short e[8];
#pragma unroll
    for (int g = 0; g < 8 / V; ++g) {
      const int kg = k + g * V;
      const bool ok = kg < K;
      const vec_t raw = *(const vec_t*)(row + (ok ? kg : k));
#pragma unroll
      for (int u = 0; u < V; ++u) e[g * V + u] = ok ? ((const unsigned short*)&raw)[u] : (unsigned short)0;
    }
    uint4 f;
    f.x = e[0] | ((unsigned)e[1] << 16); f.y = e[2] | ((unsigned)e[3] << 16);
    f.z = e[4] | ((unsigned)e[5] << 16); f.w = e[6] | ((unsigned)e[7] << 16);
    return f;
  }
}

struct SepBn {
  const float* gamma;
  const float* beta;
  const float* mean;
  const float* var;
};

struct SepBranch {
  const unsigned short* x;       // [B,H,W,C] bf16 channels_last
  const float* wdw;              // [C][k][k] fp32
  const unsigned short* wpw;     // [cout][C] bf16
  unsigned short* ydw;           // [M][C] depthwise output (saved for the backward)
  unsigned short* ypre;          // [M][cout] pointwise output before the BatchNorm (saved for the backward)
  SepBn bn;
  int k, pad_t, pad_l, present;
};

#ifndef SEP_RB5
#define SEP_RB5 3        // tap rows of a 5 x 5 kernel whose loads are in flight together (5: spills at the 128-register budget)
#endif
#define SEP_MAX_JOBS 6
struct SepMulti {
  SepBranch a[SEP_MAX_JOBS], b[SEP_MAX_JOBS];
  const unsigned short* residual[SEP_MAX_JOBS];
  unsigned short* y[SEP_MAX_JOBS];
  unsigned short* yb[SEP_MAX_JOBS];   // optional: the b branch's own output (nullptr: not stored)
};

struct SepDims {
  int B, H, W, C, cout, CP;      // CP = LDS row pitch of the depthwise tile (elements): C rounded up to 16, + 8
  long long M;
  float eps;
};

// one output pixel x V channels of the depthwise convolution: the operation order of dw_vec_accumulate (xpt_dwconv.hip,
// OXT = 1: ky outer, kx inner, one fma per tap and channel), so that the fused and the unfused path agree bit for bit --
// but ALL K x K input vectors are requested before the first one is used (one memory round trip per item; the unfused
// kernel's K round trips are hidden by its many more threads, this kernel runs one workgroup per 32 pixels)
template <int K, int V>
__device__ __forceinline__ void dw_pixel(const unsigned short* __restrict__ src, const float* __restrict__ sW, int C, int SH,
                                         int SW, int pad_t, int pad_l, int b, int oy, int ox, int c0, float (&acc)[V]) {
  // rows of taps whose loads are in flight together: all of them up to 5 x 5 (100 registers of raw bf16 vectors at V = 8),
  // one row at a time for 7 x 7 (the register budget of a 1024-thread workgroup is 128)
  constexpr int RB = K <= 3 ? K : (K == 5 ? SEP_RB5 : 1);
#pragma unroll
  for (int ky0 = 0; ky0 < K; ky0 += RB) {
    typename BfVec<V>::type raw[RB][K];
    bool ok[RB][K];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      if (ky0 + j >= K) break;                       // (compile time: the last batch of a 5 x 5 kernel has two rows)
      const int sy = oy + ky0 + j - pad_t;
      const bool row_ok = sy >= 0 && sy < SH;
      const unsigned short* row = src + (((long long)b * SH + min(max(sy, 0), SH - 1)) * SW) * C + c0;
#pragma unroll
      for (int i = 0; i < K; ++i) {                  // unconditional loads on clamped columns, zeroed by a select
        const int sx = ox - pad_l + i;
        raw[j][i] = *(const typename BfVec<V>::type*)(row + (long long)min(max(sx, 0), SW - 1) * C);
        ok[j][i] = row_ok && sx >= 0 && sx < SW;
      }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j)
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        if (ky0 + j >= K) break;
        float e[V];
        unpack(raw[j][kx], e);
        const float* wp = sW + ((ky0 + j) * K + kx) * C + c0;
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const float in = ok[j][kx] ? fmaxf(e[v], 0.f) : 0.f;      // the block's Activation('relu')
          acc[v] += in * wp[v];
        }
      }
  }
}

template <int V>
__device__ inline void depthwise_tile(const SepBranch& br, const SepDims& d, long long m0, float* sW, unsigned short* D) {
  // taps as sW[tap][c]
  const int kk = br.k * br.k;
  for (int i = threadIdx.x; i < d.C * kk; i += blockDim.x) {
    const int c = i / kk, tap = i - c * kk;
    sW[tap * d.C + c] = br.wdw[i];
  }
  __syncthreads();
  const int CG = d.C / V;
  const int hw = d.H * d.W;
  for (int idx = threadIdx.x; idx < 32 * CG; idx += blockDim.x) {
    const int px = idx / CG, c0 = (idx - px * CG) * V;
    const long long m = m0 + px;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    if (m < d.M) {
      const int b = (int)(m / hw), rem = (int)(m - (long long)b * hw);
      const int oy = rem / d.W, ox = rem - oy * d.W;
      if (br.k == 3) dw_pixel<3, V>(br.x, sW, d.C, d.H, d.W, br.pad_t, br.pad_l, b, oy, ox, c0, acc);
      else if (br.k == 5) dw_pixel<5, V>(br.x, sW, d.C, d.H, d.W, br.pad_t, br.pad_l, b, oy, ox, c0, acc);
      else dw_pixel<7, V>(br.x, sW, d.C, d.H, d.W, br.pad_t, br.pad_l, b, oy, ox, c0, acc);
    }
    unsigned w2[V / 2];
#pragma unroll
    for (int v = 0; v < V / 2; ++v) w2[v] = (unsigned)f32_to_bf16_bits(acc[2 * v]) | ((unsigned)f32_to_bf16_bits(acc[2 * v + 1]) << 16);
    typename BfVec<V>::type raw;
    if constexpr (V == 8) raw = make_uint4(w2[0], w2[1], w2[2], w2[3]);
    else if constexpr (V == 4) raw = make_uint2(w2[0], w2[1]);
    else raw = w2[0];
    *(typename BfVec<V>::type*)(D + px * d.CP + c0) = raw;
    if (m < d.M) *(typename BfVec<V>::type*)(br.ydw + m * d.C + c0) = raw;
  }
}

template <int V>
__device__ inline f32x16 pointwise_tile(const unsigned short* D, const SepBranch& br, const SepDims& d, int n0) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int col = n0 + r;
  const bool col_ok = col < d.cout;
  const unsigned short* brow = br.wpw + (long long)(col_ok ? col : d.cout - 1) * d.C;
  const unsigned short* arow = D + r * d.CP;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int ksteps = (d.C + 15) / 16;
  for (int s = 0; s < ksteps; ++s) {
    const int k = s * 16 + 8 * h;
    const bool ok = k < d.C;
    const uint4 a = *(const uint4*)(arow + k);                 // (columns C .. CP of the tile are zero)
    const uint4 bq = load_frag<V>(brow, ok ? k : 0, d.C);
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    acc = XPT_MFMA_32X32X16(__builtin_bit_cast(bf16x8, a),
                                                  __builtin_bit_cast(bf16x8, (ok && col_ok) ? bq : zero), acc);
  }
  return acc;
}

template <int V>
__global__ __launch_bounds__(1024) void sepconv_bn_multi_fwd_kernel(SepMulti m, SepDims d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int j = blockIdx.y;
  const long long m0 = (long long)blockIdx.x * 32;
  const SepBranch A = m.a[j];          // (by value: wave-uniform scalars; a reference into the indexed argument array
  const SepBranch Bb = m.b[j];         //  makes the compiler copy the struct to scratch memory)
  const bool dual = Bb.present != 0;
  unsigned short* DA = (unsigned short*)lds;
  unsigned short* DB = DA + 32 * d.CP;
  float* sW = (float*)(DB + 32 * d.CP);
  // zero the padded columns of both tiles once (the k loop reads whole 16-element steps)
  for (int i = threadIdx.x; i < 2 * 32 * d.CP; i += blockDim.x) DA[i] = 0;
  __syncthreads();
  depthwise_tile<V>(A, d, m0, sW, DA);
  if (dual) {
    __syncthreads();                                   // the taps of branch a are no longer needed
    depthwise_tile<V>(Bb, d, m0, sW, DB);
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int nwaves = blockDim.x >> 6;
  for (int n0 = wave * 32; n0 < d.cout; n0 += nwaves * 32) {
    const int col = n0 + r;
    const bool col_ok = col < d.cout;
    const int bc = col_ok ? col : d.cout - 1;
    // epilogue operands first (their round trip overlaps the MFMAs)
    const float ga = A.bn.gamma[bc], va = A.bn.var[bc], ba = A.bn.beta[bc], ma = A.bn.mean[bc];
    float gb = 0.f, vb = 1.f, bb = 0.f, mb = 0.f;
    if (dual) { gb = Bb.bn.gamma[bc]; vb = Bb.bn.var[bc]; bb = Bb.bn.beta[bc]; mb = Bb.bn.mean[bc]; }
    unsigned short res_raw[16];
    const unsigned short* residual = m.residual[j];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long long mm = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
      const long long mc = mm < d.M ? mm : d.M - 1;
      res_raw[i] = residual ? residual[mc * d.cout + bc] : (unsigned short)0;
    }
    const f32x16 acc_a = pointwise_tile<V>(DA, A, d, n0);
    f32x16 acc_b;
    if (dual) acc_b = pointwise_tile<V>(DB, Bb, d, n0);
    if (!col_ok) continue;
    const float sca = ga * rsqrtf(va + d.eps), sha = ba - ma * sca;
    const float scb = gb * rsqrtf(vb + d.eps), shb = bb - mb * scb;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long long mm = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (mm < d.M) {
        const long long o = mm * d.cout + col;
        float add = bf16_bits_to_f32(res_raw[i]);
        if (dual) {
          // the sibling branch exactly as its own launch would have produced it: BatchNorm of the ROUNDED convolution
          // output, rounded to bf16, then added as the other branch's residual
          const unsigned short pre_b = f32_to_bf16_bits(acc_b[i]);
          Bb.ypre[o] = pre_b;
          const unsigned short yb = f32_to_bf16_bits(bf16_bits_to_f32(pre_b) * scb + shb);
          if (m.yb[j]) m.yb[j][o] = yb;
          add = bf16_bits_to_f32(yb);                   // (a job has a sibling branch or a residual, not both)
        }
        const unsigned short pre_a = f32_to_bf16_bits(acc_a[i]);
        A.ypre[o] = pre_a;
        m.y[j][o] = f32_to_bf16_bits(bf16_bits_to_f32(pre_a) * sca + sha + add);
      }
    }
  }
}

}  // namespace

/* n <= 6 jobs of one activation shape [B,H,W,C] -> [B,H,W,cout] (stride 1, SAME padding, ReLU on the way in).  Arrays of
 * n entries; the *_b arrays describe the optional sibling branch of job j (x_b[j] == NULL: none) whose BatchNorm output is
 * added to the job's result; residual[j] (or NULL) is added when there is no sibling.  ydw / ypre (and ypre_b / ydw_b)
 * receive the depthwise and the pre-BatchNorm pointwise outputs (inputs of the backward kernels), y the results; yb[j]
 * (or NULL) the sibling's own BatchNorm output.  k in {3, 5, 7}; C * 2 bytes and all bases aligned to the vector width
 * the kernel picks (16 / 8 / 4 bytes). */
extern "C" int xpt_sepconv_bn_multi_fwd(int n, const void* const* x, const float* const* wdw, const void* const* wpw,
                                        const float* const* gamma, const float* const* beta, const float* const* mean,
                                        const float* const* var, const int* k, void* const* ydw, void* const* ypre,
                                        const void* const* x_b, const float* const* wdw_b, const void* const* wpw_b,
                                        const float* const* gamma_b, const float* const* beta_b, const float* const* mean_b,
                                        const float* const* var_b, const int* k_b, void* const* ydw_b, void* const* ypre_b,
                                        void* const* yb, const void* const* residual, void* const* y, float eps, int B,
                                        int H, int W, int C, int cout, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(wdw); XPT_CHECK_PTR(wpw); XPT_CHECK_PTR(gamma); XPT_CHECK_PTR(beta); XPT_CHECK_PTR(mean);
  XPT_CHECK_PTR(var); XPT_CHECK_PTR(k); XPT_CHECK_PTR(ydw); XPT_CHECK_PTR(ypre); XPT_CHECK_PTR(x_b); XPT_CHECK_PTR(residual);
  XPT_CHECK_PTR(y); XPT_CHECK_PTR(yb);
  if (n < 1 || n > SEP_MAX_JOBS) return XPT_ERR_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || cout <= 0 || C % 2 != 0) return XPT_ERR_SHAPE;
  SepMulti m{};
  int v = 8, kmax = 0;
  auto narrow = [&](const void* p) { while (v > 1 && ((uintptr_t)p) % (2 * v) != 0) v >>= 1; };
  while (v > 1 && C % v != 0) v >>= 1;
  for (int j = 0; j < n; ++j) {
    if (!x[j] || !wdw[j] || !wpw[j] || !gamma[j] || !beta[j] || !mean[j] || !var[j] || !ydw[j] || !ypre[j] || !y[j])
      return XPT_ERR_NULL;
    if (k[j] != 3 && k[j] != 5 && k[j] != 7) return XPT_ERR_ARG;
    m.a[j] = SepBranch{(const unsigned short*)x[j], wdw[j], (const unsigned short*)wpw[j], (unsigned short*)ydw[j],
                       (unsigned short*)ypre[j], SepBn{gamma[j], beta[j], mean[j], var[j]}, k[j], k[j] / 2, k[j] / 2, 1};
    narrow(x[j]); narrow(wpw[j]); narrow(ydw[j]);
    kmax = k[j] > kmax ? k[j] : kmax;
    if (x_b[j]) {
      if (!wdw_b || !wpw_b || !gamma_b || !beta_b || !mean_b || !var_b || !k_b || !ydw_b || !ypre_b) return XPT_ERR_NULL;
      if (!wdw_b[j] || !wpw_b[j] || !gamma_b[j] || !beta_b[j] || !mean_b[j] || !var_b[j] || !ydw_b[j] || !ypre_b[j])
        return XPT_ERR_NULL;
      if (k_b[j] != 3 && k_b[j] != 5 && k_b[j] != 7) return XPT_ERR_ARG;
      if (residual[j]) return XPT_ERR_ARG;                  // a sibling branch or a residual, not both
      m.b[j] = SepBranch{(const unsigned short*)x_b[j], wdw_b[j], (const unsigned short*)wpw_b[j], (unsigned short*)ydw_b[j],
                         (unsigned short*)ypre_b[j], SepBn{gamma_b[j], beta_b[j], mean_b[j], var_b[j]}, k_b[j], k_b[j] / 2,
                         k_b[j] / 2, 1};
      narrow(x_b[j]); narrow(wpw_b[j]); narrow(ydw_b[j]);
      kmax = k_b[j] > kmax ? k_b[j] : kmax;
    }
    m.residual[j] = (const unsigned short*)residual[j];
    m.y[j] = (unsigned short*)y[j];
    m.yb[j] = (unsigned short*)yb[j];
  }
  if (v < 2) return XPT_ERR_SHAPE;
  SepDims d{B, H, W, C, cout, ((C + 15) / 16) * 16 + 8, (long long)B * H * W, eps};
  const long long tiles = (d.M + 31) / 32;
  if (tiles > 0x7fffffffLL) return XPT_ERR_SHAPE;
  const size_t lds = (size_t)2 * 32 * d.CP * 2 + (size_t)kmax * kmax * C * sizeof(float);
  if (lds > 160 * 1024) return XPT_ERR_SHAPE;
  const dim3 grid((unsigned)tiles, (unsigned)n);
  // one (pixel, channel group) item per thread where the workgroup size allows it: the depthwise phase is one memory round
  // trip long; at least enough waves for the 32-column tiles of the matrix-core phase
  int threads = ((32 * (C / v) + 63) / 64) * 64;
  const int tile_waves = (cout + 31) / 32;
  if (threads < 64 * tile_waves) threads = 64 * tile_waves;
  if (threads < 256) threads = 256;
  if (threads > 1024) threads = 1024;
  hipStream_t s = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
  if (v == 8) hipLaunchKernelGGL(sepconv_bn_multi_fwd_kernel<8>, grid, dim3(threads), lds, s, m, d);
  else if (v == 4) hipLaunchKernelGGL(sepconv_bn_multi_fwd_kernel<4>, grid, dim3(threads), lds, s, m, d);
  else hipLaunchKernelGGL(sepconv_bn_multi_fwd_kernel<2>, grid, dim3(threads), lds, s, m, d);
  return xpt_launch_status();
}
