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
typedef xpt_h16x8 bf16x8;      // (8 operands of the build's 16-bit format, xpt_common.h)

__device__ inline float bf16_bits_to_f32(unsigned short u) { return xpt_h2f(u); }
__device__ inline unsigned short f32_to_bf16_bits(float f) {   // round to nearest even
  return xpt_f2h_sw(f);
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

// a second layer of the same shape whose BatchNorm output is ADDED to the main layer's in the epilogue (the right branch
// of a NASNet cell's `add`: x1 = bn(pw(left)) + bn(pw(right))) -- computed by the same workgroup, so the sum never makes
// the round trip through memory; its convolution output is kept (ypre) for the BatchNorm backward like the main one
struct PwSibling {
  const unsigned short* x;
  const unsigned short* w;
  unsigned short* ypre;
  PwBn bn;
};

constexpr int PW_G = 8;        // k steps (of 16) whose loads are issued together
constexpr int PW_TP = 36;      // row pitch (floats) of the 32 x 32 accumulator tile in LDS: 16-byte aligned rows

template <int VO> struct OutVec;
template <> struct OutVec<8> { typedef uint4 type; };
template <> struct OutVec<4> { typedef uint2 type; };
template <> struct OutVec<2> { typedef unsigned type; };
template <> struct OutVec<1> { typedef unsigned short type; };

// one 32 x 32 tile: acc += A[rows][k range] B[k range][cols]; the loads of up to PW_G k steps are issued before their MFMAs
template <int V>
__device__ inline void pw_gemm_tile(f32x16& acc, const unsigned short* __restrict__ arow,
                                    const unsigned short* __restrict__ brow, bool col_ok, int cin, int s_begin, int ksteps,
                                    int h) {
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
        acc = XPT_MFMA_32X32X16(__builtin_bit_cast(bf16x8, fa[g]),
                                                      __builtin_bit_cast(bf16x8, fb[g]), acc);
  }
}

// KW = 1: the 4 waves of a workgroup own 4 tiles stacked along the pixel axis.  KW = 4 (deep reductions on small maps:
// too few tiles to fill the chip, and one wave walking 66 k steps is 9 dependent batches of loads): the 4 waves share ONE
// tile and split its k steps into 4 contiguous ranges, summed through LDS in wave order before the epilogue.
//
// Epilogue.  An accumulator register holds ONE element of 16 different rows, i.e. 16 two-byte stores per output tensor
// and 16 two-byte loads of the residual per lane.  The tile goes through LDS instead (row-major, [32][PW_TP] floats) and
// comes back as VO consecutive channels of one row per lane: residual, ypre and y move as 16 / 8 / 4-byte vectors (VO =
// the widest the channel count and the bases allow), the residual vectors are fetched before the k loop.
template <int V, int VO, int KW = 1>
__device__ inline void pwconv_bn_fwd_body(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                          const PwBn& bn, const unsigned short* __restrict__ residual,
                                          unsigned short* __restrict__ ypre, unsigned short* __restrict__ y,
                                          const PwSibling& sib, long long M, int cin, int cout, long long pitch_x,
                                          long long pitch_y, int xcd) {
  constexpr int NTH = KW == 1 ? 64 : 256;                // threads that share the store phase of one tile
  constexpr int CPR = 32 / VO, CHUNKS = 32 * CPR, ITER = (CHUNKS + NTH - 1) / NTH;
  constexpr int TILES = KW == 1 ? 4 : 1;
  __shared__ __attribute__((aligned(16))) float s_tile[TILES][2][32 * PW_TP];
  __shared__ float s_sc[2][32], s_sh[2][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  // grid: x = row blocks (image-to-XCD numbering, xpt_common.h), y = 32-channel tiles of the output
  unsigned mb;
  if (!xpt_xcd_unit(xcd != 0, blockIdx.x, (unsigned)((M + (KW == 1 ? 127 : 31)) / (KW == 1 ? 128 : 32)), mb)) return;
  const long long m0 = (KW == 1 ? (long long)mb * 4 + wave : (long long)mb) * 32;
  const int n0 = blockIdx.y * 32;
  const bool active = m0 < M;                            // wave-uniform (KW = 4: workgroup-uniform); no early exit: barriers below
  const long long am = m0 + r < M ? m0 + r : M - 1;      // rows past the end re-read the last row (never stored)
  const int bn_ = n0 + r;
  const bool col_ok = bn_ < cout;
  const bool has_sib = sib.x != nullptr;                 // uniform
  const int bc = col_ok ? bn_ : cout - 1;

  // the epilogue's operands are fetched FIRST: the BatchNorm scale / shift of the tile's 32 channels (threads 0..31 and,
  // for the sibling, 32..63 of the workgroup) and this thread's residual vectors -- their round trip overlaps the operand loads
  if (threadIdx.x < 64) {
    const PwBn& b = threadIdx.x < 32 ? bn : sib.bn;
    if (threadIdx.x < 32 || has_sib) {
      const float sc = b.gamma[bc] * rsqrtf(b.var[bc] + b.eps);
      s_sc[threadIdx.x >> 5][r] = sc;
      s_sh[threadIdx.x >> 5][r] = b.beta[bc] - b.mean[bc] * sc;
    }
  }
  const int st = KW == 1 ? lane : (int)threadIdx.x;      // this thread's index within the store phase of its tile
  typedef typename OutVec<VO>::type ovec_t;
  ovec_t res_raw[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int c = st + it * NTH;
    const int row = c / CPR, col = (c % CPR) * VO;
    const long long m = m0 + row < M ? m0 + row : M - 1;
    const int cc = n0 + col < cout ? n0 + col : cout - VO;         // cout % VO == 0: a vector is inside or outside as a whole
    res_raw[it] = residual ? *(const ovec_t*)(residual + m * cout + cc) : ovec_t();
  }

  f32x16 acc, acc2;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = acc2[i] = 0.f;
  const int ksteps_all = (cin + 15) / 16;
  const int per_wave = (ksteps_all + KW - 1) / KW;
  const int s_begin = KW == 1 ? 0 : wave * per_wave;
  const int ksteps = KW == 1 ? ksteps_all : (s_begin + per_wave < ksteps_all ? s_begin + per_wave : ksteps_all);
  if (active) {
    const unsigned short* brow = w + (long long)bc * cin;
    pw_gemm_tile<V>(acc, x + am * pitch_x, brow, col_ok, cin, s_begin, ksteps, h);
    if (has_sib) pw_gemm_tile<V>(acc2, sib.x + am * pitch_x, sib.w + (long long)bc * cin, col_ok, cin, s_begin, ksteps, h);
  }

  float* tile = &s_tile[KW == 1 ? wave : 0][0][0];
  float* tile2 = &s_tile[KW == 1 ? wave : 0][1][0];
  if constexpr (KW > 1) {                                // the k ranges of the 4 waves, added in wave order by wave 0
    __shared__ float red[KW - 1][16][64];
    if (wave > 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) red[wave - 1][i][lane] = acc[i];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int q = 0; q < KW - 1; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += red[q][i][lane];
    }
  }
  // accumulator (reg i, lane): output row (i & 3) + 8 (i >> 2) + 4 h, column r of the tile
  if (KW == 1 || wave == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      tile[row * PW_TP + r] = acc[i];
      if (has_sib) tile2[row * PW_TP + r] = acc2[i];
    }
  }
  __syncthreads();
  if (!active) return;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int c = st + it * NTH;
    const int row = c / CPR, col = (c % CPR) * VO;
    if (c >= CHUNKS || m0 + row >= M || n0 + col >= cout) continue;
    ovec_t pre_v, y_v, pre2_v;
    unsigned short* pre_e = (unsigned short*)&pre_v;
    unsigned short* y_e = (unsigned short*)&y_v;
    unsigned short* pre2_e = (unsigned short*)&pre2_v;
    const unsigned short* res_e = (const unsigned short*)&res_raw[it];
#pragma unroll
    for (int e = 0; e < VO; ++e) {
      const unsigned short pre = f32_to_bf16_bits(tile[row * PW_TP + col + e]);
      pre_e[e] = pre;
      float add = bf16_bits_to_f32(res_e[e]);
      if (has_sib) {
        // the sibling's BatchNorm output, rounded to bf16 as it was when it made the trip through memory as a residual
        const unsigned short pre2 = f32_to_bf16_bits(tile2[row * PW_TP + col + e]);
        pre2_e[e] = pre2;
        add = bf16_bits_to_f32(f32_to_bf16_bits(bf16_bits_to_f32(pre2) * s_sc[1][col + e] + s_sh[1][col + e] + add));
      }
      // the BatchNorm sees the ROUNDED convolution output, as it does after a library GEMM with bf16 output
      y_e[e] = f32_to_bf16_bits(bf16_bits_to_f32(pre) * s_sc[0][col + e] + s_sh[0][col + e] + add);
    }
    const long long o = (m0 + row) * cout + n0 + col;
    *(ovec_t*)(ypre + o) = pre_v;
    *(ovec_t*)(y + (m0 + row) * pitch_y + n0 + col) = y_v;       // (y may be a channel slice of a wider tensor)
    if (has_sib) *(ovec_t*)(sib.ypre + o) = pre2_v;
  }
}

template <int V, int VO, int KW = 1>
__global__ __launch_bounds__(256) void pwconv_bn_fwd_kernel(const unsigned short* __restrict__ x,
                                                             const unsigned short* __restrict__ w, PwBn bn,
                                                             const unsigned short* __restrict__ residual,
                                                             unsigned short* __restrict__ ypre,
                                                             unsigned short* __restrict__ y, long long M, int cin,
                                                             int cout, long long pitch_x, int xcd) {
  const PwSibling none{};
  pwconv_bn_fwd_body<V, VO, KW>(x, w, bn, residual, ypre, y, none, M, cin, cout, pitch_x, cout, xcd);
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
  PwSibling sib[PW_MAX_JOBS];
};

template <int V, int VO>
__global__ __launch_bounds__(256) void pwconv_bn_multi_fwd_kernel(PwMulti m, long long M, int cin, int cout,
                                                                   long long pitch_x, long long pitch_y, int xcd) {
  const int j = blockIdx.z;
  pwconv_bn_fwd_body<V, VO>(m.x[j], m.w[j], m.bn[j], m.residual[j], m.ypre[j], m.y[j], m.sib[j], M, cin, cout, pitch_x,
                            pitch_y, xcd);
}

// widest vector (elements) of the output-side tensors: cout and every base must allow it
inline int out_width(int cout, std::initializer_list<const void*> ptrs) {
  int v = 8;
  while (v > 1) {
    bool ok = cout % v == 0;
    for (const void* p : ptrs) ok = ok && (p == nullptr || ((uintptr_t)p) % (2 * v) == 0);
    if (ok) break;
    v >>= 1;
  }
  return v;
}

int g_pw_ksplit_min_cin = 256, g_pw_ksplit_max_tiles = 512;   // in-step sweep: (384,0) 6.66 ms, (384,512) 6.63, (256,512) 6.61-6.62, (256,2048) 6.62

}  // namespace

extern "C" int xpt_pwconv_tune(int ksplit_min_cin, int ksplit_max_tiles) {
  if (ksplit_min_cin < 16 || ksplit_max_tiles < 0) return XPT_ERR_ARG;
  g_pw_ksplit_min_cin = ksplit_min_cin;
  g_pw_ksplit_max_tiles = ksplit_max_tiles;
  return XPT_OK;
}

// (V, VO) pairs that are instantiated: equal widths, and 16-byte operand rows with any narrower output side (the stem's
// 32 -> 11 layer on the 64 x 208 map, the 264 -> 44 heads)
#define XPT_PW_PAIRS(X) X(8, 8) X(8, 4) X(8, 2) X(8, 1) X(4, 4) X(2, 2) X(1, 1)

static void pw_widths(int& v, int& vo) {      // fold an arbitrary pair onto an instantiated one
  if (v == 8) return;
  if (vo < v) v = vo;
  vo = v;
}

static int pwconv_multi_launch(int n, const void* const* x, const void* const* w, const float* const* gamma,
                               const float* const* beta, const float* const* mean, const float* const* var, float eps,
                               const void* const* residual, void* const* ypre, void* const* y, const void* const* sib_x,
                               const void* const* sib_w, const float* const* sib_gamma, const float* const* sib_beta,
                               const float* const* sib_mean, const float* const* sib_var, void* const* sib_ypre,
                               long long M, int cin, int cout, long long pitch_x, long long pitch_y, void* stream) {
  if (n < 1 || n > PW_MAX_JOBS) return XPT_ERR_ARG;
  if (pitch_y == 0) pitch_y = cout;
  if (M <= 0 || cin <= 0 || cout <= 0 || pitch_x < cin || pitch_y < cout) return XPT_ERR_SHAPE;
  int v = 8, vo = 8;
  PwMulti m{};
  for (int j = 0; j < n; ++j) {
    if (!x[j] || !w[j] || !gamma[j] || !beta[j] || !mean[j] || !var[j] || !ypre[j] || !y[j]) return XPT_ERR_NULL;
    const bool sib = sib_x != nullptr && sib_x[j] != nullptr;
    if (sib && (!sib_w || !sib_gamma || !sib_beta || !sib_mean || !sib_var || !sib_ypre || !sib_w[j] || !sib_gamma[j] ||
                !sib_beta[j] || !sib_mean[j] || !sib_var[j] || !sib_ypre[j]))
      return XPT_ERR_NULL;
    while (v > 1 && (cin % v != 0 || pitch_x % v != 0 || ((uintptr_t)x[j]) % (2 * v) != 0 ||
                     ((uintptr_t)w[j]) % (2 * v) != 0 ||
                     (sib && (((uintptr_t)sib_x[j]) % (2 * v) != 0 || ((uintptr_t)sib_w[j]) % (2 * v) != 0))))
      v >>= 1;
    int o = out_width(cout, {residual[j], ypre[j], y[j], sib ? sib_ypre[j] : nullptr});
    while (o > 1 && pitch_y % o != 0) o >>= 1;
    if (o < vo) vo = o;
    m.x[j] = (const unsigned short*)x[j];
    m.w[j] = (const unsigned short*)w[j];
    m.residual[j] = (const unsigned short*)residual[j];
    m.ypre[j] = (unsigned short*)ypre[j];
    m.y[j] = (unsigned short*)y[j];
    m.bn[j] = PwBn{gamma[j], beta[j], mean[j], var[j], eps};
    if (sib)
      m.sib[j] = PwSibling{(const unsigned short*)sib_x[j], (const unsigned short*)sib_w[j], (unsigned short*)sib_ypre[j],
                           PwBn{sib_gamma[j], sib_beta[j], sib_mean[j], sib_var[j], eps}};
  }
  pw_widths(v, vo);
  const long long mblocks = (M + 127) / 128;
  if (mblocks > 65535) return XPT_ERR_SHAPE;
  const int xcd = g_xpt_xcd_affinity && mblocks >= 8;      // (fewer row blocks than XCDs: every XCD takes part, no numbering)
  const dim3 grid(xcd ? xpt_xcd_pad(mblocks) : (unsigned)mblocks, (cout + 31) / 32, n);
  if (pitch_y != cout)                          // residuals are dense [M, cout] tensors; with a pitched y there are none
    for (int j = 0; j < n; ++j)
      if (residual[j]) return XPT_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
#define XPT_PW_CASE(VV, VVO)                                                                                         \
  if (v == VV && vo == VVO)                                                                                          \
    hipLaunchKernelGGL((pwconv_bn_multi_fwd_kernel<VV, VVO>), grid, dim3(256), 0, s, m, M, cin, cout, pitch_x, pitch_y, xcd);
  XPT_PW_PAIRS(XPT_PW_CASE)
#undef XPT_PW_CASE
  return xpt_launch_status();
}

extern "C" int xpt_pwconv_bn_multi_fwd(int n, const void* const* x, const void* const* w, const float* const* gamma,
                                       const float* const* beta, const float* const* mean, const float* const* var,
                                       float eps, const void* const* residual, void* const* ypre, void* const* y,
                                       long long M, int cin, int cout, long long pitch_x, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(w); XPT_CHECK_PTR(gamma); XPT_CHECK_PTR(beta); XPT_CHECK_PTR(mean);
  XPT_CHECK_PTR(var); XPT_CHECK_PTR(residual); XPT_CHECK_PTR(ypre); XPT_CHECK_PTR(y);
  return pwconv_multi_launch(n, x, w, gamma, beta, mean, var, eps, residual, ypre, y, nullptr, nullptr, nullptr, nullptr,
                             nullptr, nullptr, nullptr, M, cin, cout, pitch_x, 0, stream);
}

/* The same launch with SIBLING layers: where sib_x[j] is not NULL, job j computes
 *   y_j = bn_j(x_j w_j^T) + bf16(bn'_j(sib_x_j sib_w_j^T)) (+ residual_j)
 * i.e. the second operand of the cell's `add` is a layer of the same shape evaluated by the same workgroups (its sum with
 * the main layer never goes through memory); sib_ypre_j receives the sibling's convolution output for its BatchNorm
 * backward.  Bit for bit what two launches (siblings first, their outputs as residuals of the second) produce. */
extern "C" int xpt_pwconv_bn_multi_fwd_sib(int n, const void* const* x, const void* const* w, const float* const* gamma,
                                           const float* const* beta, const float* const* mean, const float* const* var,
                                           float eps, const void* const* residual, void* const* ypre, void* const* y,
                                           const void* const* sib_x, const void* const* sib_w,
                                           const float* const* sib_gamma, const float* const* sib_beta,
                                           const float* const* sib_mean, const float* const* sib_var,
                                           void* const* sib_ypre, long long M, int cin, int cout, long long pitch_x,
                                           long long pitch_y, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(w); XPT_CHECK_PTR(gamma); XPT_CHECK_PTR(beta); XPT_CHECK_PTR(mean);
  XPT_CHECK_PTR(var); XPT_CHECK_PTR(residual); XPT_CHECK_PTR(ypre); XPT_CHECK_PTR(y);
  return pwconv_multi_launch(n, x, w, gamma, beta, mean, var, eps, residual, ypre, y, sib_x, sib_w, sib_gamma, sib_beta,
                             sib_mean, sib_var, sib_ypre, M, cin, cout, pitch_x, pitch_y, stream);
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
  int vo = out_width(cout, {residual, ypre, y});
  const PwBn bn{gamma, beta, mean, var, eps};
  hipStream_t s = (hipStream_t)stream;
  int xcd = g_xpt_xcd_affinity && (M + 31) / 32 >= 8;
  // deep reduction, few tiles: the 4 waves of a workgroup split the k steps of one tile (pwconv_bn_fwd_body, KW = 4)
  const long long tiles = ((M + 31) / 32) * ((cout + 31) / 32);
  if (v == 8 && vo == 8 && cin >= g_pw_ksplit_min_cin && tiles <= g_pw_ksplit_max_tiles && (M + 31) / 32 <= 65535) {
    const dim3 gridk(xcd ? xpt_xcd_pad((M + 31) / 32) : (unsigned)((M + 31) / 32), (cout + 31) / 32);
    XPT_BEGIN_LAUNCH();
    hipLaunchKernelGGL((pwconv_bn_fwd_kernel<8, 8, 4>), gridk, dim3(256), 0, s, (const unsigned short*)x,
                       (const unsigned short*)w, bn, (const unsigned short*)residual, (unsigned short*)ypre,
                       (unsigned short*)y, M, cin, cout, pitch_x, xcd);
    return xpt_launch_status();
  }
  pw_widths(v, vo);
  const long long mblocks = (M + 127) / 128;
  if (mblocks > 65535) return XPT_ERR_SHAPE;
  xcd = g_xpt_xcd_affinity && mblocks >= 8;
  const dim3 grid(xcd ? xpt_xcd_pad(mblocks) : (unsigned)mblocks, (cout + 31) / 32);
  XPT_BEGIN_LAUNCH();
#define XPT_PW_CASE(VV, VVO)                                                                                     \
  if (v == VV && vo == VVO)                                                                                      \
    hipLaunchKernelGGL((pwconv_bn_fwd_kernel<VV, VVO>), grid, dim3(256), 0, s, (const unsigned short*)x,         \
                       (const unsigned short*)w, bn, (const unsigned short*)residual, (unsigned short*)ypre,     \
                       (unsigned short*)y, M, cin, cout, pitch_x, xcd);
  XPT_PW_PAIRS(XPT_PW_CASE)
#undef XPT_PW_CASE
  return xpt_launch_status();
}
