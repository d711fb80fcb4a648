// xpt_pwconv.hip -- forward of a pointwise (1x1) convolution with its BatchNorm (+ branch add) as ONE launch.
//
//   ypre[m, co] = sum_ci x[m, ci] * W[co, ci]                (bf16 operands, fp32 accumulation, stored as bf16)
//   y[m, co]    = ypre * s[co] + shift[co] (+ residual[m, co]),   s = gamma * rsqrt(var + eps), shift = beta - mean * s
//
// Every pointwise convolution of NASNet-A-Mobile (the second half of each SeparableConv2D and the cell heads) is followed
// by an inference-mode BatchNormalization (keras nasnet._separable_conv_block / _adjust_block / _normal_a_cell); with a
// library GEMM that is a second launch re-reading the GEMM output.  Both operands are k-contiguous (activation rows
// and filter rows), which is exactly the operand layout of v_mfma_f32_32x32x16_bf16: lane (r = lane & 31, h = lane >> 5)
// holds A[row r][k = 8h .. 8h+7] and B[k = 8h .. 8h+7][col r], i.e. one 16-byte global load per operand and k step with
// no LDS staging and no transposition.  A wave owns a 32x32 output tile; the 4 waves of a workgroup stack along the
// pixel axis and share the filter rows through the L1.  The k loop issues the loads of up to 8 k steps (16 per lane)
// before the MFMAs that consume them.  ypre is kept because the fused conv+BN backward (xpt_gemm.hip, BnFuse) needs it
// for dgamma.  Rows that are only 4- or 2-byte aligned (22 / 11 channels: the stem cells) are read with
// narrower loads (load_frag).
#include "xpt_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline float bf16_bits_to_f32(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ inline unsigned short f32_to_bf16_bits(float f) {   // round to nearest even
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// 8 consecutive bf16 of one row starting at element k (k < K); elements at or past K read as zero.  V = elements per
// load: rows whose pitch / base only allow 8-, 4- or 2-byte loads (44, 22 or 11 channels) take 2, 4 or 8 loads per fragment.
template <int V> struct FragVec;
template <> struct FragVec<4> { typedef uint2 type; };
template <> struct FragVec<2> { typedef unsigned type; };
template <> struct FragVec<1> { typedef unsigned short type; };

template <int V>
__device__ inline uint4 load_frag(const unsigned short* __restrict__ row, int k, int K) {
  if constexpr (V == 8) {
    return *(const uint4*)(row + k);
  } else {
    typedef typename FragVec<V>::type vec_t;
    unsigned short e[8];
#pragma unroll
    for (int g = 0; g < 8 / V; ++g) {               // unconditional loads from a clamped k, zeroed by select
      const int kg = k + g * V;
      const bool ok = kg < K;                        // K % V == 0: a vector is inside or outside as a whole
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

struct PwBn {
  const float* gamma;
  const float* beta;
  const float* mean;
  const float* var;
  float eps;
};

constexpr int PW_G = 8;   // k steps (of 16) whose loads are issued together

// KW = 1: the 4 waves of a workgroup own 4 tiles stacked along the pixel axis.  KW = 4 (deep reductions on small maps:
// too few tiles to fill the chip, and one wave walking 66 k steps is 9 dependent batches of loads): the 4 waves share ONE
// tile and split its k steps into 4 contiguous ranges, summed through LDS in wave order before the epilogue.
template <int V, int KW = 1>
__device__ inline void pwconv_bn_fwd_body(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                          const PwBn& bn, const unsigned short* __restrict__ residual,
                                          unsigned short* __restrict__ ypre, unsigned short* __restrict__ y,
                                          long long M, int cin, int cout, long long pitch_x) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const long long m0 = (KW == 1 ? (long long)blockIdx.y * 4 + wave : (long long)blockIdx.y) * 32;
  const int n0 = blockIdx.x * 32;
  if (m0 >= M) return;                                   // wave-uniform (KW = 4: workgroup-uniform, barriers below)
  const long long am = m0 + r < M ? m0 + r : M - 1;      // rows past the end re-read the last row (never stored)
  const int bn_ = n0 + r;
  const bool col_ok = bn_ < cout;
  const unsigned short* arow = x + am * pitch_x;
  const unsigned short* brow = w + (long long)(col_ok ? bn_ : cout - 1) * cin;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  // the epilogue's operands (BatchNorm parameters of this lane's output channel, the residual elements of its 16 output
  // rows) are fetched FIRST, from clamped addresses: their round trip then overlaps the first batch of operand loads
  // instead of following the last MFMA
  const int bc = col_ok ? bn_ : cout - 1;
  const float e_gamma = bn.gamma[bc], e_var = bn.var[bc], e_beta = bn.beta[bc], e_mean = bn.mean[bc];
  unsigned short res_raw[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const long long m = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
    const long long mc = m < M ? m : M - 1;
    res_raw[i] = residual ? residual[mc * cout + bc] : (unsigned short)0;
  }

  const int ksteps_all = (cin + 15) / 16;
  const int per_wave = (ksteps_all + KW - 1) / KW;
  const int s_begin = KW == 1 ? 0 : wave * per_wave;
  const int ksteps = KW == 1 ? ksteps_all : (s_begin + per_wave < ksteps_all ? s_begin + per_wave : ksteps_all);
  for (int s0 = s_begin; s0 < ksteps; s0 += PW_G) {
    uint4 fa[PW_G], fb[PW_G];
#pragma unroll
    for (int g = 0; g < PW_G; ++g) {                     // unconditional loads from clamped k, zeroed by select
      const int k = (s0 + g) * 16 + 8 * h;
      const bool ok = k < cin;
      const int kc = ok ? k : 0;
      const uint4 a = load_frag<V>(arow, kc, cin), b = load_frag<V>(brow, kc, cin);
      const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
      fa[g] = ok ? a : zero;
      fb[g] = (ok && col_ok) ? b : zero;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < PW_G; ++g)
      if (s0 + g < ksteps)                               // wave-uniform
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[g]),
                                                      __builtin_bit_cast(bf16x8, fb[g]), acc, 0, 0, 0);
  }

  if constexpr (KW > 1) {                                // the k ranges of the 4 waves, added in wave order
    __shared__ float red[KW - 1][16][64];
    if (wave > 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) red[wave - 1][i][lane] = acc[i];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int q = 0; q < KW - 1; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] += red[q][i][lane];
  }
  // accumulator (reg i, lane): output row m0 + (i & 3) + 8 (i >> 2) + 4 h, column n0 + r
  if (!col_ok) return;
  const float sc = e_gamma * rsqrtf(e_var + bn.eps);
  const float sh = e_beta - e_mean * sc;
  float res[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) res[i] = bf16_bits_to_f32(res_raw[i]);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const long long m = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (m < M) {
      const unsigned short pre = f32_to_bf16_bits(acc[i]);
      ypre[m * cout + bn_] = pre;
      // the BatchNorm sees the ROUNDED convolution output, as it does after a library GEMM with bf16 output
      y[m * cout + bn_] = f32_to_bf16_bits(bf16_bits_to_f32(pre) * sc + sh + res[i]);
    }
  }
}

template <int V, int KW = 1>
__global__ __launch_bounds__(256) void pwconv_bn_fwd_kernel(const unsigned short* __restrict__ x,
                                                             const unsigned short* __restrict__ w, PwBn bn,
                                                             const unsigned short* __restrict__ residual,
                                                             unsigned short* __restrict__ ypre,
                                                             unsigned short* __restrict__ y, long long M, int cin,
                                                             int cout, long long pitch_x) {
  pwconv_bn_fwd_body<V, KW>(x, w, bn, residual, ypre, y, M, cin, cout, pitch_x);
}

// up to 6 independent layers of one shape (the branch convolutions of a cell stage): job = blockIdx.z
#define PW_MAX_JOBS 6
struct PwMulti {
  const unsigned short* x[PW_MAX_JOBS];
  const unsigned short* w[PW_MAX_JOBS];
  const unsigned short* residual[PW_MAX_JOBS];
  unsigned short* ypre[PW_MAX_JOBS];
  unsigned short* y[PW_MAX_JOBS];
  PwBn bn[PW_MAX_JOBS];
};

template <int V>
__global__ __launch_bounds__(256) void pwconv_bn_multi_fwd_kernel(PwMulti m, long long M, int cin, int cout,
                                                                   long long pitch_x) {
  const int j = blockIdx.z;
  pwconv_bn_fwd_body<V>(m.x[j], m.w[j], m.bn[j], m.residual[j], m.ypre[j], m.y[j], M, cin, cout, pitch_x);
}

int g_pw_ksplit_min_cin = 256, g_pw_ksplit_max_tiles = 512;   // in-step sweep: (384,0) 6.66 ms, (384,512) 6.63, (256,512) 6.61-6.62, (256,2048) 6.62

}  // namespace

extern "C" int xpt_pwconv_tune(int ksplit_min_cin, int ksplit_max_tiles) {
  if (ksplit_min_cin < 16 || ksplit_max_tiles < 0) return XPT_ERR_ARG;
  g_pw_ksplit_min_cin = ksplit_min_cin;
  g_pw_ksplit_max_tiles = ksplit_max_tiles;
  return XPT_OK;
}

extern "C" int xpt_pwconv_bn_multi_fwd(int n, const void* const* x, const void* const* w, const float* const* gamma,
                                       const float* const* beta, const float* const* mean, const float* const* var,
                                       float eps, const void* const* residual, void* const* ypre, void* const* y,
                                       long long M, int cin, int cout, long long pitch_x, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(w); XPT_CHECK_PTR(gamma); XPT_CHECK_PTR(beta); XPT_CHECK_PTR(mean);
  XPT_CHECK_PTR(var); XPT_CHECK_PTR(residual); XPT_CHECK_PTR(ypre); XPT_CHECK_PTR(y);
  if (n < 1 || n > PW_MAX_JOBS) return XPT_ERR_ARG;
  if (M <= 0 || cin <= 0 || cout <= 0 || pitch_x < cin) return XPT_ERR_SHAPE;
  int v = 8;
  PwMulti m{};
  for (int j = 0; j < n; ++j) {
    if (!x[j] || !w[j] || !gamma[j] || !beta[j] || !mean[j] || !var[j] || !ypre[j] || !y[j]) return XPT_ERR_NULL;
    while (v > 1 && (cin % v != 0 || pitch_x % v != 0 || ((uintptr_t)x[j]) % (2 * v) != 0 ||
                     ((uintptr_t)w[j]) % (2 * v) != 0))
      v >>= 1;
    m.x[j] = (const unsigned short*)x[j];
    m.w[j] = (const unsigned short*)w[j];
    m.residual[j] = (const unsigned short*)residual[j];
    m.ypre[j] = (unsigned short*)ypre[j];
    m.y[j] = (unsigned short*)y[j];
    m.bn[j] = PwBn{gamma[j], beta[j], mean[j], var[j], eps};
  }
  const long long mblocks = (M + 127) / 128;
  if (mblocks > 65535) return XPT_ERR_SHAPE;
  const dim3 grid((cout + 31) / 32, (unsigned)mblocks, n);
  hipStream_t s = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
  if (v == 8)
    hipLaunchKernelGGL(pwconv_bn_multi_fwd_kernel<8>, grid, dim3(256), 0, s, m, M, cin, cout, pitch_x);
  else if (v == 4)
    hipLaunchKernelGGL(pwconv_bn_multi_fwd_kernel<4>, grid, dim3(256), 0, s, m, M, cin, cout, pitch_x);
  else if (v == 2)
    hipLaunchKernelGGL(pwconv_bn_multi_fwd_kernel<2>, grid, dim3(256), 0, s, m, M, cin, cout, pitch_x);
  else
    hipLaunchKernelGGL(pwconv_bn_multi_fwd_kernel<1>, grid, dim3(256), 0, s, m, M, cin, cout, pitch_x);
  return xpt_launch_status();
}

extern "C" int xpt_pwconv_bn_fwd(const void* x, const void* w, const float* gamma, const float* beta, const float* mean,
                                 const float* var, float eps, const void* residual, void* ypre, void* y, long long M,
                                 int cin, int cout, long long pitch_x, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(w); XPT_CHECK_PTR(gamma); XPT_CHECK_PTR(beta); XPT_CHECK_PTR(mean);
  XPT_CHECK_PTR(var); XPT_CHECK_PTR(ypre); XPT_CHECK_PTR(y);
  if (M <= 0 || cin <= 0 || cout <= 0 || pitch_x < cin) return XPT_ERR_SHAPE;
  // widest operand load (elements) the row pitch, the channel count and the bases allow
  int v = 8;
  while (v > 1 && (cin % v != 0 || pitch_x % v != 0 || ((uintptr_t)x) % (2 * v) != 0 || ((uintptr_t)w) % (2 * v) != 0))
    v >>= 1;
  const PwBn bn{gamma, beta, mean, var, eps};
  hipStream_t s = (hipStream_t)stream;
  // deep reduction, few tiles: the 4 waves of a workgroup split the k steps of one tile (pwconv_bn_fwd_body, KW = 4)
  const long long tiles = ((M + 31) / 32) * ((cout + 31) / 32);
  if (v == 8 && cin >= g_pw_ksplit_min_cin && tiles <= g_pw_ksplit_max_tiles && (M + 31) / 32 <= 65535) {
    const dim3 gridk((cout + 31) / 32, (unsigned)((M + 31) / 32));
    XPT_BEGIN_LAUNCH();
    hipLaunchKernelGGL((pwconv_bn_fwd_kernel<8, 4>), gridk, dim3(256), 0, s, (const unsigned short*)x,
                       (const unsigned short*)w, bn, (const unsigned short*)residual, (unsigned short*)ypre,
                       (unsigned short*)y, M, cin, cout, pitch_x);
    return xpt_launch_status();
  }
  const long long mblocks = (M + 127) / 128;
  if (mblocks > 65535) return XPT_ERR_SHAPE;
  const dim3 grid((cout + 31) / 32, (unsigned)mblocks);
  XPT_BEGIN_LAUNCH();
  if (v == 8)
    hipLaunchKernelGGL(pwconv_bn_fwd_kernel<8>, grid, dim3(256), 0, s, (const unsigned short*)x,
                       (const unsigned short*)w, bn, (const unsigned short*)residual, (unsigned short*)ypre,
                       (unsigned short*)y, M, cin, cout, pitch_x);
  else if (v == 4)
    hipLaunchKernelGGL(pwconv_bn_fwd_kernel<4>, grid, dim3(256), 0, s, (const unsigned short*)x,
                       (const unsigned short*)w, bn, (const unsigned short*)residual, (unsigned short*)ypre,
                       (unsigned short*)y, M, cin, cout, pitch_x);
  else if (v == 2)
    hipLaunchKernelGGL(pwconv_bn_fwd_kernel<2>, grid, dim3(256), 0, s, (const unsigned short*)x,
                       (const unsigned short*)w, bn, (const unsigned short*)residual, (unsigned short*)ypre,
                       (unsigned short*)y, M, cin, cout, pitch_x);
  else
    hipLaunchKernelGGL(pwconv_bn_fwd_kernel<1>, grid, dim3(256), 0, s, (const unsigned short*)x,
                       (const unsigned short*)w, bn, (const unsigned short*)residual, (unsigned short*)ypre,
                       (unsigned short*)y, M, cin, cout, pitch_x);
  return xpt_launch_status();
}
