// xpt_gemm.hip -- weight gradient of the pointwise (1x1) convolutions of NASNet-Mobile on the gfx950 matrix cores.
//
//   dW[co, ci] = sum_m dy[m, co] * x[m, ci]        m = the B*H*W pixels of an NHWC activation, 416 ... 106496 here
//
// The output is tiny (11x32 ... 1056x1056) and the reduction long, the shape the BLAS library serves worst (one or a
// few workgroups walk the whole K loop: 19-67 us per call, 4.8 ms of a 26 ms step).  This kernel splits the reduction
// over enough workgroups to fill the chip and finishes in the same launch:
//   * one wave owns a 32x32 tile of dW and issues v_mfma_f32_32x32x16_bf16 (bf16 operands, f32 accumulate: the products
//     of two bf16 are exact in f32, only the accumulation rounds).  The reduction index (pixels) is the slow axis of both
//     row-major activations, so the rows are staged in LDS as they lie in memory (coalesced 16-byte loads) and read back
//     transposed (ds_read_b64_tr_b16): lane (r, h) gets pixels 8h .. 8h+7 of channel r for A = dy and for B = x;
//   * a workgroup = 4 waves = up to 4 tile quadrants (64x64, 64x32, 32x64) or, for narrow outputs, several interleaved
//     slices of the reduction that are combined through LDS;
//   * the reduction is split over gridDim.z workgroups per tile; each writes its partial tile to the workspace, and the
//     workgroup that arrives last (one counter per tile, reset for the next call) adds the partials IN SPLIT ORDER and
//     writes dW: deterministic, no second launch, no float atomics.
#include "xpt_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradPlan {
  int tco, tci;        // tile extents (32 or 64)
  int tiles_co, tiles_ci;
  int waves;           // waves per workgroup (4 or 16)
  int nsplit;
  long long rows_per_block;
};

// tuning knobs (xpt_conv1x1_bwd_weight_tune; defaults measured on MI355X)
int g_waves = 8;              // waves per workgroup (8: two workgroups per CU at <= 128 VGPRs; in-step 16/8 pairs -> 7.17 ms, 8/16 -> 7.05 ms)
int g_pairs_per_wave = 16;    // row pairs (MFMAs) each wave should at least get (in-step sweep with 8 waves: 8 / 12 / 16 / 20 / 24 -> 7.20 / 7.09 / 7.05 / 7.21 / 7.19 ms)
int g_max_blocks = 1024;      // workgroups per launch
int g_max_partial_kib = 512;  // partial tiles the finishing workgroup adds, per output tile
int g_defer_cap_mib = 8;      // deferred mode: bytes of split partials one layer may leave for xpt_reduce_partials

inline WgradPlan wgrad_plan(long long M, int cout, int cin, bool defer = false) {
  WgradPlan p;
  p.tco = cout <= 32 ? 32 : 64;
  p.tci = cin <= 32 ? 32 : 64;
  p.tiles_co = (cout + p.tco - 1) / p.tco;
  p.tiles_ci = (cin + p.tci - 1) / p.tci;
  p.waves = g_waves;
  const long long ntiles = (long long)p.tiles_co * p.tiles_ci;
  const int quadrants = (p.tco / 32) * (p.tci / 32);
  const int kslices = p.waves / quadrants;
  long long rows = 2LL * g_pairs_per_wave * kslices;
  // larger weights: more rows per workgroup (fewer splits, so fewer partial tiles to write and to reduce; in-step sweep of
  // the thresholds, round 3: 5.58 -> 5.53 ms).  The 44 x 44 layers of the first stack keep the short workgroups.
  if ((long long)cout * cin >= 2048) rows *= 2;
  if ((long long)cout * cin >= 30000) rows *= 2;
  // splits: bounded by what the finishing workgroup can add (in-kernel finish) or by 8 MiB of partials (deferred)
  long long smax = defer ? ((long long)g_defer_cap_mib << 20) / ((long long)cout * cin * 4)
                         : ((long long)g_max_partial_kib * 1024) / ((long long)p.tco * p.tci * 4);
  const long long by_blocks = g_max_blocks / ntiles;
  if (smax > by_blocks) smax = by_blocks;
  if (smax < 1) smax = 1;
  long long nsplit = (M + rows - 1) / rows;
  if (nsplit > smax) {
    nsplit = smax;
    rows = (M + nsplit - 1) / nsplit;
  }
  rows = (rows + 7) / 8 * 8;
  p.rows_per_block = rows;
  p.nsplit = (int)((M + rows - 1) / rows);
  return p;
}

__device__ inline float bf16_bits_to_f32(unsigned short u) { return xpt_h2f(u); }

// staged vector: V consecutive bf16 of one activation row
template <int V> struct StageVec;
template <> struct StageVec<8> { typedef uint4 type; };
template <> struct StageVec<4> { typedef uint2 type; };
template <> struct StageVec<2> { typedef unsigned type; };
template <> struct StageVec<1> { typedef unsigned short type; };

// global -> registers: this thread's share of a [RC rows x TW channels] tile (rows >= k_end and channels >= C read as 0)
template <int V, int TW, int RC, int NT>
__device__ inline void stage_load(typename StageVec<V>::type (&reg)[(RC * TW / V + NT - 1) / NT],
                                  const unsigned short* __restrict__ src, long long pitch, long long k0,
                                  long long k_end, int c0, int C) {
  typedef typename StageVec<V>::type vec_t;
  constexpr int VPR = TW / V, NV = RC * VPR, PER = (NV + NT - 1) / NT;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int v = threadIdx.x + i * NT;
    const int row = v / VPR, c = c0 + (v % VPR) * V;
    // unconditional load from a clamped (row, channel) and a select: guarded loads would each get a branch and their
    // own s_waitcnt, serialising the up to 8 loads of a thread
    const bool ok = v < NV && k0 + row < k_end && c < C;
    const long long rr = k0 + row < k_end ? k0 + row : k_end - 1;
    const vec_t val = *(const vec_t*)(src + rr * pitch + (c < C ? c : C - V));
    reg[i] = ok ? val : vec_t();
  }
}

template <int V, int TW, int RC, int NT>
__device__ inline void stage_store(unsigned short* dst, const typename StageVec<V>::type (&reg)[(RC * TW / V + NT - 1) / NT]) {
  typedef typename StageVec<V>::type vec_t;
  constexpr int VPR = TW / V, NV = RC * VPR, PER = (NV + NT - 1) / NT;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int v = threadIdx.x + i * NT;
    if (v < NV) *(vec_t*)(dst + (v / VPR) * TW + (v % VPR) * V) = reg[i];
  }
}

// Optional fusion of the BatchNorm backward that precedes the convolution's backward (conv -> BN in the forward):
//   the incoming gradient dy is the gradient of the BN OUTPUT; g = dy * s (s = gamma * rsqrt(var + eps)) is the gradient
//   of the convolution output.  The kernel scales its A operand by s (so D = g^T x is the filter gradient itself), the
//   workgroups of the first input-channel tile also write g (bf16, for the data-gradient GEMM) and the per-split
//   partials of dbeta = sum dy and dgamma = rstd (sum dy * ypre - mean * sum dy), ypre = the convolution output.
struct BnFuse {
  const float* gamma;
  const float* var;
  const float* mean;
  float eps;
  const unsigned short* ypre;   // [M, cout] bf16, dense
  unsigned short* g_out;        // [M, cout] bf16, dense
  float* partial;               // [nsplit][2][cout]
  // gradient fan-in: dy is the SUM of up to three tensors (the layer's output feeds several branches of a cell); the
  // extras are added on load (fp32, rounded to bf16 once: what the separate sum_rows launch produced)
  const unsigned short* dy_extra[2];
  long long pitch_extra[2];
  int n_extra;
};

__device__ inline unsigned short f32_to_bf16_bits(float f) {   // round to nearest even
  return xpt_f2h_sw(f);
}

// reg += the same elements of up to two more tensors (fp32 accumulation in source order, one bf16 rounding)
template <int V, int TW, int RC, int NT>
__device__ inline void stage_accumulate(typename StageVec<V>::type (&reg)[(RC * TW / V + NT - 1) / NT],
                                        const unsigned short* const (&src)[2], const long long (&pitch)[2], int nsrc,
                                        long long k0, long long k_end, int c0, int C) {
  typedef typename StageVec<V>::type vec_t;
  constexpr int VPR = TW / V, NV = RC * VPR, PER = (NV + NT - 1) / NT;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int v = threadIdx.x + i * NT;
    const int row = v / VPR, c = c0 + (v % VPR) * V;
    const bool ok = v < NV && k0 + row < k_end && c < C;
    const long long rr = k0 + row < k_end ? k0 + row : k_end - 1;
    vec_t extra[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)                      // unconditional loads (clamped), see stage_load
      extra[q] = *(const vec_t*)(src[q < nsrc ? q : 0] + rr * pitch[q < nsrc ? q : 0] + (c < C ? c : C - V));
    vec_t cur = reg[i], out;
    const unsigned short* e = (const unsigned short*)&cur;
    unsigned short* o = (unsigned short*)&out;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float f = bf16_bits_to_f32(e[j]);
#pragma unroll
      for (int q = 0; q < 2; ++q)
        f += q < nsrc ? bf16_bits_to_f32(((const unsigned short*)&extra[q])[j]) : 0.f;
      o[j] = f32_to_bf16_bits(f);
    }
    reg[i] = ok ? out : vec_t();
  }
}

// g = dy * s for this thread's staged vectors of dy, written to g_out (rows / channels outside the tensor skipped)
template <int V, int TW, int RC, int NT>
__device__ inline void store_scaled(unsigned short* __restrict__ g_out, const float* sS,
                                    const typename StageVec<V>::type (&reg)[(RC * TW / V + NT - 1) / NT],
                                    long long pitch, long long k0, long long k_end, int c0, int C) {
  typedef typename StageVec<V>::type vec_t;
  constexpr int VPR = TW / V, NV = RC * VPR, PER = (NV + NT - 1) / NT;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int v = threadIdx.x + i * NT;
    const int row = v / VPR, cl = (v % VPR) * V;
    if (v < NV && k0 + row < k_end && c0 + cl < C) {
      vec_t in = reg[i], out;
      const unsigned short* e = (const unsigned short*)&in;
      unsigned short* o = (unsigned short*)&out;
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = f32_to_bf16_bits(bf16_bits_to_f32(e[j]) * sS[cl + j]);
      *(vec_t*)(g_out + (k0 + row) * pitch + c0 + cl) = out;
    }
  }
}

// TCO x TCI output tile per workgroup of NW waves.  Q = quadrants of 32x32, KS = NW / Q interleaved k slices.
// VA / VB: elements per global load of dy / x (the host picks the widest the pitch, base and channel count allow).
// Several layers of one shape in one launch (the branch convolutions of a cell stage): job = blockIdx.z / nsplit.
#define WG_MAX_JOBS 6
struct WgradMulti {
  const unsigned short* dy[WG_MAX_JOBS];
  const unsigned short* x[WG_MAX_JOBS];
  float* partial[WG_MAX_JOBS];
  long long pitch_dy[WG_MAX_JOBS];
  BnFuse bn[WG_MAX_JOBS];
  int n;                                                    // 0: single layer, the scalar arguments are used
};

// The DATA gradient of the same layer(s), dx = (dy * s) W, computed by EXTRA workgroups of the same launch (the first
// `slices` blockIdx.z slices: they are short, so they are dispatched first and the weight-gradient workgroups of the
// second round take over their compute units early): it needs nothing the weight-gradient workgroups produce (g = dy * s is formed on load), so the
// two kinds of workgroup simply share the launch and run side by side -- one launch per layer group instead of a
// weight-gradient launch followed by a library GEMM (104 GEMM launches, 0.88 ms per step at batch 8).
struct DxFuse {
  unsigned short* dx[WG_MAX_JOBS];          // [M, cin] bf16, dense
  const unsigned short* w[WG_MAX_JOBS];     // [cout, cin] bf16, dense (the bf16 shadow of the weight)
  int job[WG_MAX_JOBS];                     // index into WgradMulti of the layer this data gradient belongs to
  int n;                                    // data gradients wanted
  int slices;                               // leading blockIdx.z slices that belong to the data gradient (0: none)
  int row_blocks, nsub;                     // a workgroup's tile: row_blocks x 32 rows by nsub x 32 input channels
  int wgs_rows, wgs_ci;                     // workgroups per layer along the rows / the input channels
  int vw;                                   // elements per global load of W (its rows and base are aligned to it)
};

typedef xpt_h16x8 bf16x8;      // (8 operands of the build's 16-bit format, xpt_common.h)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// two transposed LDS reads = one bf16 MFMA fragment: per 16-lane group ds_read_b64_tr_b16 takes 4 rows x 16 columns and
// hands lane i the 4 rows of column i (csrc/xpt_conv_wgrad.hip uses the same reads for the k x k weight gradient)
__device__ inline bf16x8 tr_pair(const char* lds, unsigned off0, unsigned off1) {
  typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(lds + off0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(lds + off1));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}
#define DX_KC 64          // rows of W staged per chunk
#define DX_LD 132         // LDS row pitch of the staged W slice in 16-bit elements (8 rows apart = 16 banks apart)

// One workgroup of NT / 64 waves: wave = (row block rb, channel sub-tile ns); v_mfma_f32_32x32x16_bf16 with A = g rows read
// straight from global (lane (r, h): row r, k = 8h..8h+7: consecutive channels of one gradient row), B = W staged in LDS.
// Per chunk of DX_KC output channels every global load (the chunk's A vectors, the W slice) is issued before the first
// wait, so a chunk costs one memory round trip, not one per k step.
template <int VW, int NT> struct DxWRegs {
  static constexpr int VPR = 128 / VW, PER = (DX_KC * VPR + NT - 1) / NT;
  typename StageVec<VW>::type reg[PER];
};

// global -> registers: this thread's share of W[kc0 .. kc0 + DX_KC)[ci_base .. ci_base + 128) (zeros outside the matrix)
template <int VW, int NT>
__device__ inline void dx_load_w(DxWRegs<VW, NT>& t, const unsigned short* __restrict__ w, int kc0, int cout, int cin,
                                 int ci_base) {
  typedef typename StageVec<VW>::type vec_t;
#pragma unroll
  for (int i = 0; i < DxWRegs<VW, NT>::PER; ++i) {
    const int v = threadIdx.x + i * NT;
    const int k = v / DxWRegs<VW, NT>::VPR, c = (v % DxWRegs<VW, NT>::VPR) * VW;
    const int co = kc0 + k, cc = ci_base + c;
    const bool ok = k < DX_KC && co < cout && cc < cin;                    // cin % VW == 0: whole vectors
    const vec_t val = *(const vec_t*)(w + (long long)(co < cout ? co : cout - 1) * cin + (cc < cin ? cc : cin - VW));
    t.reg[i] = ok ? val : vec_t();
  }
}

template <int VW, int NT>
__device__ inline void dx_store_w(const DxWRegs<VW, NT>& t, unsigned short* sW) {
  typedef typename StageVec<VW>::type vec_t;
#pragma unroll
  for (int i = 0; i < DxWRegs<VW, NT>::PER; ++i) {
    const int v = threadIdx.x + i * NT;
    const int k = v / DxWRegs<VW, NT>::VPR, c = (v % DxWRegs<VW, NT>::VPR) * VW;
    if (k < DX_KC) {
      if constexpr (VW == 8) {                                             // LDS rows are 8-byte aligned
        *(uint2*)(sW + k * DX_LD + c) = make_uint2(t.reg[i].x, t.reg[i].y);
        *(uint2*)(sW + k * DX_LD + c + 4) = make_uint2(t.reg[i].z, t.reg[i].w);
      } else {
        *(vec_t*)(sW + k * DX_LD + c) = t.reg[i];
      }
    }
  }
}

// One workgroup of NT / 64 waves: wave = (row block rb, channel sub-tile ns); v_mfma_f32_32x32x16_bf16 with A = g rows read
// straight from global (lane (r, h): row r, k = 8h..8h+7: consecutive channels of one gradient row), B = W staged in LDS
// in chunks of DX_KC output channels.  Every global load of a chunk (its A vectors, its W slice) is issued before the
// first wait; without fan-in pieces and up to 3 chunks (cout <= 192: all of NASNet-Mobile) ALL chunks' loads are issued
// up front, so the workgroup pays one memory round trip in total.
template <int VA, bool EX, int VW, int NT>
__device__ inline void dgrad_body(const unsigned short* __restrict__ dy, long long pitch_dy, const BnFuse& bn,
                                  const unsigned short* __restrict__ w, unsigned short* __restrict__ dx, long long M,
                                  int cout, int cin, int rchunk, int cit, int row_blocks, int nsub,
                                  unsigned char* smem) {
  typedef typename StageVec<VA>::type vec_t;
  constexpr int G = 8 / VA, KSTEPS = DX_KC / 16, HOIST = (EX || VW == 1) ? 1 : (NT >= 1024 ? 3 : 2);   // chunks whose loads are in flight together
  float* sScale = (float*)smem;                                   // [<= 256] BN scales (0 beyond cout)
  unsigned short* sW = (unsigned short*)(smem + 1024);            // [DX_KC][DX_LD]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int rb = wave % row_blocks, ns = wave / row_blocks;
  const bool active = ns < nsub;
  const long long row0 = ((long long)rchunk * row_blocks + rb) * 32;
  const long long row = row0 + r;
  const long long rr = row < M ? row : M - 1;
  const int ci_base = cit * nsub * 32;
  const int ci = ci_base + ns * 32 + r;
  const int kp = (cout + 15) / 16 * 16;
  const int nx = EX ? bn.n_extra : 0;             // EX: the output gradient arrives in up to three pieces (fan-in)
  for (int c = threadIdx.x; c < kp; c += blockDim.x)
    sScale[c] = c < cout ? (bn.gamma ? bn.gamma[c] * rsqrtf(bn.var[c] + bn.eps) : 1.f) : 0.f;   // no BatchNorm: scale 1
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int kg0 = 0; kg0 < cout; kg0 += HOIST * DX_KC) {
    // unconditional loads from clamped addresses (guarded loads would each get their own wait)
    vec_t a0[HOIST][KSTEPS][G], a1[EX ? KSTEPS : 1][G], a2[EX ? KSTEPS : 1][G];
    DxWRegs<VW, NT> wr[HOIST];
#pragma unroll
    for (int q = 0; q < HOIST; ++q) {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int kg = kg0 + q * DX_KC + ks * 16 + 8 * h + g * VA;
          const int kc = kg < cout ? kg : cout - VA;             // cout % VA == 0: a vector is inside or outside as a whole
          a0[q][ks][g] = *(const vec_t*)(dy + rr * pitch_dy + kc);
          if constexpr (EX) {
            if (nx > 0) a1[ks][g] = *(const vec_t*)(bn.dy_extra[0] + rr * bn.pitch_extra[0] + kc);
            if (nx > 1) a2[ks][g] = *(const vec_t*)(bn.dy_extra[1] + rr * bn.pitch_extra[1] + kc);
          }
        }
      dx_load_w<VW, NT>(wr[q], w, kg0 + q * DX_KC, cout, cin, ci_base);
    }
#pragma unroll
    for (int q = 0; q < HOIST; ++q) {
      const int kc0 = kg0 + q * DX_KC;
      if (kc0 >= cout) break;                                     // uniform
      __syncthreads();                                            // the previous chunk's reads are done (and sScale is set)
      dx_store_w<VW, NT>(wr[q], sW);
      __syncthreads();
      if (!active) continue;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int kbase = kc0 + ks * 16;
        if (kbase < cout) {                                       // uniform
          unsigned short a16[8];
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const int kg = kbase + 8 * h + g * VA;
            const bool ok = kg < cout;
#pragma unroll
            for (int j = 0; j < VA; ++j) {
              float f = bf16_bits_to_f32(((const unsigned short*)&a0[q][ks][g])[j]);
              if constexpr (EX) {
                if (nx > 0) {                                     // the fan-in sum, rounded once (as the weight part does)
                  f += bf16_bits_to_f32(((const unsigned short*)&a1[ks][g])[j]);
                  if (nx > 1) f += bf16_bits_to_f32(((const unsigned short*)&a2[ks][g])[j]);
                  f = bf16_bits_to_f32(f32_to_bf16_bits(f));
                }
              }
              a16[g * VA + j] = ok ? f32_to_bf16_bits(f * sScale[ok ? kg + j : 0]) : (unsigned short)0;
            }
          }
          unsigned short b16[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) b16[j] = sW[(ks * 16 + 8 * h + j) * DX_LD + ns * 32 + r];
          uint4 fa, fb;
          fa.x = a16[0] | ((unsigned)a16[1] << 16); fa.y = a16[2] | ((unsigned)a16[3] << 16);
          fa.z = a16[4] | ((unsigned)a16[5] << 16); fa.w = a16[6] | ((unsigned)a16[7] << 16);
          fb.x = b16[0] | ((unsigned)b16[1] << 16); fb.y = b16[2] | ((unsigned)b16[3] << 16);
          fb.z = b16[4] | ((unsigned)b16[5] << 16); fb.w = b16[6] | ((unsigned)b16[7] << 16);
          acc = XPT_MFMA_32X32X16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb),
                                                        acc);
        }
      }
    }
  }
  if (active && ci < cin) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long long orow = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
      if (orow < M) dx[orow * cin + ci] = f32_to_bf16_bits(acc[i]);
    }
  }
}

struct WgGrid {
  unsigned tiles_ci, tiles_co;      // tiles of dW along cin / cout (= gridDim.x / gridDim.y of the 3-D grid)
  int xcd;                          // image-to-XCD numbering of the workgroups (1-D grid)
  unsigned inner_d, inner_w, per_job;                          // slots x ci blocks; jobs x tiles; tiles
  unsigned m_inner_d, m_wgs_ci, m_inner_w, m_per_job, m_gx;    // xpt_magic() of the decode's divisors
};
__device__ __forceinline__ unsigned xpt_xcd_pad_dev(unsigned n) { return (n + 7u) & ~7u; }

template <int TCO, int TCI, int VA, int VB, bool BN, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, NWAVES == 8 ? 4 : 1) void conv1x1_wgrad_kernel(const unsigned short* __restrict__ dy_,
                                                              const unsigned short* __restrict__ x_,
                                                              float* __restrict__ dw, float* __restrict__ partial_,
                                                              unsigned* __restrict__ counters, long long M, int cout,
                                                              int cin, long long pitch_dy_, long long pitch_x,
                                                              long long rows_per_block, int nsplit, int defer,
                                                              BnFuse bn_, WgradMulti mj, DxFuse dxf, WgGrid wg) {
  const unsigned short* __restrict__ dy = dy_;
  const unsigned short* __restrict__ x = x_;
  float* __restrict__ partial = partial_;
  long long pitch_dy = pitch_dy_;
  BnFuse bn = bn_;
  // Workgroup roles.  wg.xcd == 0: the 3-D grid (ci tile, co tile, z), z < dxf.slices = data-gradient workgroups numbered
  // linearly, then job * nsplit + split.  wg.xcd != 0 (image-to-XCD numbering, xpt_common.h): a 1-D grid -- first the data-
  // gradient part, 8 ceil(wgs_rows / 8) row blocks x (slots x ci blocks), then the weight-gradient part, 8 ceil(nsplit / 8)
  // row splits x (jobs x tiles); XCD k (= x % 8 in both parts, their sizes are multiples of 8) owns the row blocks / splits
  // of rows [k M / 8, (k + 1) M / 8).
  unsigned bx = blockIdx.x, by = blockIdx.y;
  int bz = blockIdx.z, dx_slot = -1, dx_rows = 0, dx_ci = 0, wjob = -1, wsplit = 0;
  const unsigned gx = wg.tiles_ci;
  if (wg.xcd) {
    const unsigned d8 = dxf.slices ? xpt_xcd_pad_dev((unsigned)dxf.wgs_rows) * wg.inner_d : 0u;   // (short workgroups: dispatched first)
    if (blockIdx.x < d8) {
      const unsigned q = blockIdx.x >> 3, rl = xpt_fastdiv(q, wg.m_inner_d), in = q - rl * wg.inner_d;
      unsigned rb;
      if (!xpt_xcd_unit(true, (blockIdx.x & 7u) | (rl << 3), (unsigned)dxf.wgs_rows, rb)) return;
      bz = 0;                                               // (a data-gradient workgroup: dxf.slices >= 1 here)
      dx_slot = (int)xpt_fastdiv(in, wg.m_wgs_ci);
      dx_ci = (int)(in - (unsigned)dx_slot * (unsigned)dxf.wgs_ci);
      dx_rows = (int)rb;
    } else {
      const unsigned lin = blockIdx.x - d8;
      const unsigned q = lin >> 3, sl = xpt_fastdiv(q, wg.m_inner_w), in = q - sl * wg.inner_w;
      unsigned sp;
      if (!xpt_xcd_unit(true, (lin & 7u) | (sl << 3), (unsigned)nsplit, sp)) return;
      const unsigned job = xpt_fastdiv(in, wg.m_per_job), t = in - job * wg.per_job;
      by = xpt_fastdiv(t, wg.m_gx);
      bx = t - by * gx;
      bz = dxf.slices + (int)(job * (unsigned)nsplit + sp);
      wjob = (int)job;
      wsplit = (int)sp;
    }
  }
  int split = bz - dxf.slices;                              // < 0: a data-gradient workgroup (below)
  if (!(BN && mj.n > 0) && wjob >= 0) split = wsplit;
  if (BN && mj.n > 0 && split >= 0) {
    const int job = wjob >= 0 ? wjob : split / nsplit;
    split = wjob >= 0 ? wsplit : split - job * nsplit;
    dy = mj.dy[job];
    x = mj.x[job];
    partial = mj.partial[job];
    pitch_dy = mj.pitch_dy[job];
    bn = mj.bn[job];
  }
  constexpr int NW = NWAVES, QA = TCO / 32, QB = TCI / 32, Q = QA * QB, KS = NW / Q, NT = NW * 64;
  constexpr int TILE = TCO * TCI;
  // rows staged per chunk: 32 / 48 / 32 KiB of LDS (BN fusion: a third buffer for ypre, 48 / 40 / 48 KiB)
  // (a 64 x 64 tile on 8 waves = 2 k slices x 16 rows x XPT_PW_WGRAD pairs: the plan's 64 rows per workgroup are one chunk)
  constexpr int RC = Q == 4 ? 64 : (BN ? (Q == 1 ? 256 : 128) : 256);
  constexpr int STAGE_BYTES = RC * (TCO + TCI + (BN ? TCO : 0)) * 2, RED_BYTES = (KS - 1) * Q * 4096;
  constexpr int DX_BYTES = VA > 1 ? 1024 + DX_KC * DX_LD * 2 : 0;
  constexpr int SMEM_A = STAGE_BYTES > RED_BYTES ? STAGE_BYTES : RED_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_A > DX_BYTES ? SMEM_A : DX_BYTES];
  if constexpr (VA > 1)              // (scalar staging = odd channel counts: the host never asks for a data gradient)
  if (bz < dxf.slices) {                                                 // a data-gradient workgroup (see DxFuse)
    static_assert(sizeof(smem) >= 1024 + DX_KC * DX_LD * 2, "the staged W slice must fit the staging buffers");
    int slot = dx_slot;
    if (!wg.xcd) {
      const int lid = (bz * (int)wg.tiles_co + (int)by) * (int)gx + (int)bx;
      const int per_job = dxf.wgs_rows * dxf.wgs_ci;
      slot = lid / per_job;
      if (slot >= dxf.n) return;
      const int rem = lid - slot * per_job;
      dx_rows = rem / dxf.wgs_ci;
      dx_ci = rem - dx_rows * dxf.wgs_ci;
    }
    if (BN && mj.n > 0) {
      const int job = dxf.job[slot];
      dy = mj.dy[job];
      pitch_dy = mj.pitch_dy[job];
      bn = mj.bn[job];
    }
#define XPT_DX(EX, VW)                                                                                               \
  dgrad_body<VA, EX, VW, NT>(dy, pitch_dy, bn, dxf.w[slot], dxf.dx[slot], M, cout, cin, dx_rows, dx_ci, \
                         dxf.row_blocks, dxf.nsub, smem)
    if (bn.n_extra > 0) {
      if (dxf.vw == 8) XPT_DX(true, 8);
      else if (dxf.vw == 4) XPT_DX(true, 4);
      else XPT_DX(true, 1);
    } else {
      if (dxf.vw == 8) XPT_DX(false, 8);
      else if (dxf.vw == 4) XPT_DX(false, 4);
      else XPT_DX(false, 1);
    }
#undef XPT_DX
    return;
  }
  __shared__ unsigned last_flag;
  __shared__ float sS[BN ? TCO : 1];                        // BN scale of the tile's output channels
  __shared__ float bnred[BN ? KS * QA * 64 : 1];            // column sums of the k slices
  unsigned short* sA = (unsigned short*)smem;               // [RC][TCO] rows of dy
  unsigned short* sB = sA + RC * TCO;                       // [RC][TCI] rows of x
  unsigned short* sY = sB + RC * TCI;                       // [RC][TCO] rows of ypre (BN fusion, first ci tile only)
  float* red = (float*)smem;                                // reused once the staging buffers are dead

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int quad = wave % Q, ksub = wave / Q;
  const int qa = quad / QB, qb = quad % QB;
  const int tile = by * gx + bx;
  const int co0 = by * TCO, ci0 = bx * TCI;

  const long long k_begin = (long long)split * rows_per_block;
  long long k_end = k_begin + rows_per_block;
  if (k_end > M) k_end = M;
  const int ci = ci0 + qb * 32 + r;
  const bool ci_ok = ci < cin;
  const bool bn_tile = BN && bx == 0;               // this workgroup also emits g and the BN partial sums
  const bool bn_sums = bn_tile && qb == 0;                  // ... summed by the waves of the first column quadrant
  float sum_dy = 0.f, sum_dyy = 0.f;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  // chunk loop: the activations are staged through LDS with the widest loads the layout allows (a per-lane 2-byte
  // global load costs the texture addresser as much as a 16-byte one), zero-filled outside the tile; each wave then
  // reads its MFMA operands A = dy[k + h][co], B = x[k + h][ci] as 16-bit LDS reads.  The next chunk's global loads
  // are in flight (registers) while this chunk's MFMAs run.
  typename StageVec<VA>::type ga[(RC * TCO / VA + NT - 1) / NT];
  typename StageVec<VB>::type gb[(RC * TCI / VB + NT - 1) / NT];
  typename StageVec<VA>::type gy[BN ? (RC * TCO / VA + NT - 1) / NT : 1];
  stage_load<VA, TCO, RC, NT>(ga, dy, pitch_dy, k_begin, k_end, co0, cout);
  if (BN && bn.n_extra > 0)                          // uniform: the layer's output gradient arrives in pieces
    stage_accumulate<VA, TCO, RC, NT>(ga, bn.dy_extra, bn.pitch_extra, bn.n_extra, k_begin, k_end, co0, cout);
  stage_load<VB, TCI, RC, NT>(gb, x, pitch_x, k_begin, k_end, ci0, cin);
  if constexpr (BN) {
    if (bn_tile) stage_load<VA, TCO, RC, NT>(gy, bn.ypre, cout, k_begin, k_end, co0, cout);
  }
  // the BatchNorm scales: fetched BEHIND the first chunk's loads (their round trip used to head every workgroup)
  if (BN) {
    if (threadIdx.x < TCO) {
      const int c = co0 + threadIdx.x;
      sS[threadIdx.x] = c < cout ? bn.gamma[c] * rsqrtf(bn.var[c] + bn.eps) : 0.f;
    }
    __syncthreads();
  }
  // lane roles of the transposed reads: every lane supplies the address of 4 elements of one staged row
  unsigned a_off[2], b_off[2];
  {
    const int li = lane & 15, q = li >> 2, p = li & 3, half = (lane >> 4) & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pr = 8 * h + 4 * i + q;                     // row of the 16-row step
      a_off[i] = (unsigned)((pr * TCO + qa * 32 + 16 * half + 4 * p) * 2);
      b_off[i] = (unsigned)((pr * TCI + qb * 32 + 16 * half + 4 * p) * 2);
    }
  }
  for (long long k0 = k_begin; k0 < k_end; k0 += RC) {
    __syncthreads();                                        // the previous chunk's operand reads are done
    stage_store<VA, TCO, RC, NT>(sA, ga);
    stage_store<VB, TCI, RC, NT>(sB, gb);
    if constexpr (BN) {
      if (bn_tile) {
        stage_store<VA, TCO, RC, NT>(sY, gy);
        if (bn.g_out) store_scaled<VA, TCO, RC, NT>(bn.g_out, sS, ga, cout, k0, k_end, co0, cout);
      }
    }
    __syncthreads();
    if (k0 + RC < k_end) {
      stage_load<VA, TCO, RC, NT>(ga, dy, pitch_dy, k0 + RC, k_end, co0, cout);
      if (BN && bn.n_extra > 0)
        stage_accumulate<VA, TCO, RC, NT>(ga, bn.dy_extra, bn.pitch_extra, bn.n_extra, k0 + RC, k_end, co0, cout);
      stage_load<VB, TCI, RC, NT>(gb, x, pitch_x, k0 + RC, k_end, ci0, cin);
      if constexpr (BN) {
        if (bn_tile) stage_load<VA, TCO, RC, NT>(gy, bn.ypre, cout, k0 + RC, k_end, co0, cout);
      }
    }
    // 16-row steps ksub, ksub + KS, ... of the chunk (rows past k_end hold zeros); bounds are wave-uniform.  Both
    // operands come out of LDS transposed (ds_read_b64_tr_b16, see tr_pair): lane (r, h) gets rows 8h .. 8h+7 of column r,
    // the fragment v_mfma_f32_32x32x16_bf16 wants -- RAW bf16 gradients: the BatchNorm scale is applied to the
    // accumulator rows once, behind the loop.
    const long long left = k_end - k0;
    const int steps = left >= RC ? RC / 16 : (int)((left + 15) / 16);
    for (int j = ksub; j < steps; j += KS) {
      const bf16x8 fa = tr_pair((const char*)sA, a_off[0] + j * (32 * TCO), a_off[1] + j * (32 * TCO));
      const bf16x8 fb = tr_pair((const char*)sB, b_off[0] + j * (32 * TCI), b_off[1] + j * (32 * TCI));
      if (BN && bn_sums) {
        const s16x8 ua = __builtin_bit_cast(s16x8, fa);
        const s16x8 uy = __builtin_bit_cast(s16x8, tr_pair((const char*)sY, a_off[0] + j * (32 * TCO), a_off[1] + j * (32 * TCO)));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float a_raw = bf16_bits_to_f32((unsigned short)ua[i]);
          sum_dy += a_raw;
          sum_dyy += a_raw * bf16_bits_to_f32((unsigned short)uy[i]);
        }
      }
      acc = XPT_MFMA_32X32X16(fa, fb, acc);
    }
  }
  if (BN && bn_sums) {       // fold the two row halves of the wave, park the k slice's column sums
    sum_dy += __shfl_xor(sum_dy, 32, 64);
    sum_dyy += __shfl_xor(sum_dyy, 32, 64);
    if (lane < 32) {
      bnred[(ksub * QA + qa) * 64 + r] = sum_dy;
      bnred[(ksub * QA + qa) * 64 + 32 + r] = sum_dyy;
    }
  }
  __syncthreads();                                          // staging buffers dead: `red` may overwrite them
  if (BN && bn_sums && ksub == 0 && lane < 32) {            // k slices in order -> this split's share of dbeta / dgamma
    float t1 = 0.f, t2 = 0.f;
    for (int ks = 0; ks < KS; ++ks) {
      t1 += bnred[(ks * QA + qa) * 64 + r];
      t2 += bnred[(ks * QA + qa) * 64 + 32 + r];
    }
    const int c = co0 + qa * 32 + r;
    if (c < cout) {
      const float rstd = rsqrtf(bn.var[c] + bn.eps);
      float* P = bn.partial + (long long)split * 2 * cout;
      P[c] = t1;
      P[cout + c] = rstd * (t2 - bn.mean[c] * t1);
    }
  }

  // combine the k slices of one quadrant through LDS (fixed order: slice 0 + slice 1 + ...)
  if (KS > 1) {
    if (ksub > 0) {
      float* dst = red + ((ksub - 1) * Q + quad) * 1024;
#pragma unroll
      for (int i = 0; i < 16; ++i) dst[i * 64 + lane] = acc[i];
    }
    __syncthreads();
    if (ksub == 0) {
#pragma unroll
      for (int s = 1; s < KS; ++s) {
        const float* src = red + ((s - 1) * Q + quad) * 1024;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += src[i * 64 + lane];
      }
    }
  }

  // accumulator element (reg i, lane) = dW[co0 + qa*32 + (i&3) + 8*(i>>2) + 4*h][ci0 + qb*32 + r]
  if (BN && ksub == 0) {                                     // g = dy * s: the scale of the accumulator's row
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] *= sS[qa * 32 + (i & 3) + 8 * (i >> 2) + 4 * h];
  }
  if (defer) {   // partial[split][cout][cin]: added up later, together with every other layer's, by xpt_reduce_partials
    if (ksub == 0 && ci_ok) {
      float* mine = partial + (long long)split * cout * cin;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = co0 + qa * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (row < cout) mine[(long long)row * cin + ci] = acc[i];
      }
    }
    return;
  }
  if (nsplit == 1) {
    if (ksub == 0 && ci_ok) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = co0 + qa * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (row < cout) dw[(long long)row * cin + ci] = acc[i];
      }
    }
    return;
  }

  float* mine = partial + ((long long)tile * nsplit + split) * TILE;
  if (ksub == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = qa * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
      mine[row * TCI + qb * 32 + r] = acc[i];
    }
  }
  // publish the partial tile, then count arrivals.  One thread fences for the workgroup (the barrier orders the other
  // waves' stores before it): the agent-scope release writes this XCD's L2 back once per workgroup, not once per wave.
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned arrived = atomicAdd(&counters[tile], 1u);
    const unsigned last = (arrived == (unsigned)nsplit - 1u) ? 1u : 0u;
    if (last) __threadfence();
    last_flag = last;
  }
  __syncthreads();
  if (last_flag == 0u) return;
  // the last workgroup adds the partial tiles in split order.  All loads of a batch are issued before the first add
  // (the partials come from other XCDs' L2 / HBM: one exposed latency per batch, not per split).
  const float* base = partial + (long long)tile * nsplit * TILE;
  constexpr int EPT = (TILE + NT - 1) / NT, BATCH = 32 / EPT;
  float sum[EPT];
#pragma unroll
  for (int j = 0; j < EPT; ++j) sum[j] = 0.f;
  for (int sp0 = 0; sp0 < nsplit; sp0 += BATCH) {
    float v[EPT][BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int sp = sp0 + u < nsplit ? sp0 + u : nsplit - 1;
#pragma unroll
      for (int j = 0; j < EPT; ++j) {
        const int e = threadIdx.x + j * NT;
        v[j][u] = __hip_atomic_load(base + (long long)sp * TILE + (e < TILE ? e : 0), __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const bool live = sp0 + u < nsplit;
#pragma unroll
      for (int j = 0; j < EPT; ++j) sum[j] += live ? v[j][u] : 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    const int e = threadIdx.x + j * NT;
    const int row = co0 + e / TCI, col = ci0 + e % TCI;
    if (e < TILE && row < cout && col < cin) dw[(long long)row * cin + col] = sum[j];
  }
  if (threadIdx.x == 0) counters[tile] = 0u;   // ready for the next call on this stream
}

}  // namespace

extern "C" int xpt_conv1x1_bwd_weight_tune(int waves, int pairs_per_wave, int max_blocks, int max_partial_kib) {
  if ((waves != 16 && waves != 8) || pairs_per_wave < 1 || max_blocks < 1 || max_partial_kib < 16) return XPT_ERR_ARG;
  g_waves = waves;
  g_pairs_per_wave = pairs_per_wave;
  g_max_blocks = max_blocks;
  g_max_partial_kib = max_partial_kib;
  return XPT_OK;
}

extern "C" int xpt_conv1x1_bwd_weight_defer_cap(int mib) {
  if (mib < 1 || mib > 64) return XPT_ERR_ARG;
  g_defer_cap_mib = mib;
  return XPT_OK;
}

extern "C" size_t xpt_conv1x1_bwd_weight_workspace_floats(long long M, int cout, int cin) {
  if (M <= 0 || cout <= 0 || cin <= 0) return 0;
  const WgradPlan p = wgrad_plan(M, cout, cin);
  if (p.nsplit == 1) return 1;
  return (size_t)p.tiles_co * p.tiles_ci * p.nsplit * p.tco * p.tci;
}

extern "C" int xpt_conv1x1_bwd_weight_counters(long long M, int cout, int cin) {
  if (M <= 0 || cout <= 0 || cin <= 0) return 0;
  const WgradPlan p = wgrad_plan(M, cout, cin);
  return p.tiles_co * p.tiles_ci;
}

static int wgrad_launch(const void* dy, const void* x, float* dw, float* workspace, unsigned* counters, long long M,
                        int cout, int cin, long long pitch_dy, long long pitch_x, const WgradPlan& p, int defer,
                        void* stream, const BnFuse* bn = nullptr, const WgradMulti* multi = nullptr,
                        const DxFuse* dx = nullptr) {
  hipStream_t s = (hipStream_t)stream;
  const unsigned short* a = (const unsigned short*)dy;
  const unsigned short* b = (const unsigned short*)x;
  // widest staging load (elements) all operands allow: rows must start and end on a vector boundary
  auto width = [](const void* ptr, long long pitch, int C) {
    int v = 8;
    while (v > 1 && (pitch % v != 0 || C % v != 0 || ((uintptr_t)ptr) % (2 * v) != 0)) v >>= 1;
    return v;
  };
  int v = width(dy, pitch_dy, cout);
  const int vb = width(x, pitch_x, cin);
  if (vb < v) v = vb;
  if (bn) {
    const int vy = width(bn->ypre, cout, cout), vg = width(bn->g_out, cout, cout);
    if (vy < v) v = vy;
    if (vg < v) v = vg;
    for (int q = 0; q < bn->n_extra; ++q) {
      const int ve = width(bn->dy_extra[q], bn->pitch_extra[q], cout);
      if (ve < v) v = ve;
    }
  }
  WgradMulti mj{};
  if (multi) {
    mj = *multi;
    for (int j = 0; j < mj.n; ++j) {
      const int c[4] = {width(mj.dy[j], mj.pitch_dy[j], cout), width(mj.x[j], pitch_x, cin),
                        width(mj.bn[j].ypre, cout, cout), width(mj.bn[j].g_out, cout, cout)};
      for (int q = 0; q < 4; ++q)
        if (c[q] < v) v = c[q];
      for (int q = 0; q < mj.bn[j].n_extra; ++q) {
        const int ve = width(mj.bn[j].dy_extra[q], mj.bn[j].pitch_extra[q], cout);
        if (ve < v) v = ve;
      }
    }
  }
  // scalar staging (odd channel counts): the 8-wave instantiation spills at its 128-register cap; 16 waves there
  const int waves = v == 1 ? 16 : p.waves;
  if (v == 1 && dx && dx->n > 0) return XPT_ERR_ARG;      // data gradients ride only in the vector-staged instantiations
  const int wz = p.nsplit * (multi ? multi->n : 1);
  DxFuse dxf{};
  if (dx && dx->n > 0) {
    dxf = *dx;
    dxf.nsub = (cin + 31) / 32 < 4 ? (cin + 31) / 32 : 4;
    dxf.row_blocks = waves / dxf.nsub;
    dxf.wgs_rows = (int)((M + 32 * dxf.row_blocks - 1) / (32 * dxf.row_blocks));
    dxf.wgs_ci = (cin + 32 * dxf.nsub - 1) / (32 * dxf.nsub);
    dxf.vw = 8;
    for (int j = 0; j < dxf.n; ++j)
      while (dxf.vw > 1 && (cin % dxf.vw != 0 || ((uintptr_t)dxf.w[j]) % (2 * dxf.vw) != 0)) dxf.vw >>= 1;
    if (dxf.vw == 2) dxf.vw = 1;            // (8, 4 and 1 are instantiated)
    const long long per_slice = (long long)p.tiles_ci * p.tiles_co;
    dxf.slices = (int)(((long long)dxf.n * dxf.wgs_rows * dxf.wgs_ci + per_slice - 1) / per_slice);
  }
  const long long dz = dxf.slices;
  if (wz + dz > 65535) return XPT_ERR_SHAPE;
  WgGrid wg{};
  wg.tiles_ci = (unsigned)p.tiles_ci;
  wg.tiles_co = (unsigned)p.tiles_co;
  wg.xcd = g_xpt_xcd_affinity && p.nsplit >= 8;      // (fewer row splits than XCDs: every XCD takes part, no numbering)
  wg.per_job = wg.tiles_ci * wg.tiles_co;
  wg.inner_w = wg.per_job * (unsigned)(multi ? multi->n : 1);
  wg.inner_d = (unsigned)(dxf.n * dxf.wgs_ci);
  wg.m_inner_d = xpt_magic(wg.inner_d);
  wg.m_wgs_ci = xpt_magic((unsigned)dxf.wgs_ci);
  wg.m_inner_w = xpt_magic(wg.inner_w);
  wg.m_per_job = xpt_magic(wg.per_job);
  wg.m_gx = xpt_magic(wg.tiles_ci);
  dim3 grid(p.tiles_ci, p.tiles_co, (unsigned)(wz + dz));
  if (wg.xcd) {
    const unsigned long long jobs = multi ? multi->n : 1;
    const unsigned long long total = (unsigned long long)xpt_xcd_pad(p.nsplit) * p.tiles_ci * p.tiles_co * jobs +
                                     (dxf.slices ? (unsigned long long)xpt_xcd_pad(dxf.wgs_rows) * dxf.n * dxf.wgs_ci : 0);
    if (total > 0x7fffffffull) return XPT_ERR_SHAPE;
    grid = dim3((unsigned)total);
  }
  const dim3 block(waves * 64);
  const BnFuse none{};
  XPT_BEGIN_LAUNCH();
#define XPT_WGRAD_W(TCO, TCI, V, NWV)                                                                                 \
  do {                                                                                                                \
    if (bn)                                                                                                           \
      hipLaunchKernelGGL((conv1x1_wgrad_kernel<TCO, TCI, V, V, true, NWV>), grid, block, 0, s, a, b, dw, workspace,   \
                         counters, M, cout, cin, pitch_dy, pitch_x, p.rows_per_block, p.nsplit, defer, *bn, mj, dxf, wg); \
    else                                                                                                              \
      hipLaunchKernelGGL((conv1x1_wgrad_kernel<TCO, TCI, V, V, false, NWV>), grid, block, 0, s, a, b, dw, workspace,  \
                         counters, M, cout, cin, pitch_dy, pitch_x, p.rows_per_block, p.nsplit, defer, none, mj, dxf, wg); \
  } while (0)
#define XPT_WGRAD(TCO, TCI, V)                                                                                        \
  do {                                                                                                                \
    if (waves == 8)                                                                                                   \
      XPT_WGRAD_W(TCO, TCI, V, 8);                                                                                    \
    else                                                                                                              \
      XPT_WGRAD_W(TCO, TCI, V, 16);                                                                                   \
  } while (0)
#define XPT_WGRAD_V(V)                                                                                               \
  do {                                                                                                               \
    if (p.tco == 64 && p.tci == 64)                                                                                  \
      XPT_WGRAD(64, 64, V);                                                                                          \
    else if (p.tco == 64)                                                                                            \
      XPT_WGRAD(64, 32, V);                                                                                          \
    else if (p.tci == 64)                                                                                            \
      XPT_WGRAD(32, 64, V);                                                                                          \
    else                                                                                                             \
      XPT_WGRAD(32, 32, V);                                                                                          \
  } while (0)
  if (v == 8)
    XPT_WGRAD_V(8);
  else if (v == 4)
    XPT_WGRAD_V(4);
  else if (v == 2)
    XPT_WGRAD_V(2);
  else
    XPT_WGRAD_V(1);
#undef XPT_WGRAD_V
#undef XPT_WGRAD
#undef XPT_WGRAD_W
  return xpt_launch_status();
}

extern "C" int xpt_conv1x1_bwd_weight(const void* dy, const void* x, float* dw, float* workspace,
                                      size_t workspace_floats, unsigned* counters, int n_counters, long long M,
                                      int cout, int cin, long long pitch_dy, long long pitch_x, void* stream) {
  XPT_CHECK_PTR(dy);
  XPT_CHECK_PTR(x);
  XPT_CHECK_PTR(dw);
  XPT_CHECK_PTR(workspace);
  XPT_CHECK_PTR(counters);
  if (M <= 0 || cout <= 0 || cin <= 0 || pitch_dy < cout || pitch_x < cin) return XPT_ERR_SHAPE;
  const WgradPlan p = wgrad_plan(M, cout, cin);
  if (workspace_floats < xpt_conv1x1_bwd_weight_workspace_floats(M, cout, cin)) return XPT_ERR_WORKSPACE;
  if (n_counters < p.tiles_co * p.tiles_ci) return XPT_ERR_WORKSPACE;
  if (p.tiles_co > 65535 || p.nsplit > 65535) return XPT_ERR_SHAPE;
  return wgrad_launch(dy, x, dw, workspace, counters, M, cout, cin, pitch_dy, pitch_x, p, 0, stream);
}

extern "C" int xpt_conv1x1_bwd_weight_splits(long long M, int cout, int cin) {
  if (M <= 0 || cout <= 0 || cin <= 0) return 0;
  return wgrad_plan(M, cout, cin, true).nsplit;
}

/* Deferred weight gradient: partials[split][cout][cin] (xpt_conv1x1_bwd_weight_splits() of them), to be added up
 * later by xpt_reduce_partials; no counters, no in-kernel finishing pass. */
extern "C" int xpt_conv1x1_bwd_weight_partials(const void* dy, const void* x, float* partials, size_t partial_floats,
                                               long long M, int cout, int cin, long long pitch_dy, long long pitch_x,
                                               void* stream) {
  XPT_CHECK_PTR(dy);
  XPT_CHECK_PTR(x);
  XPT_CHECK_PTR(partials);
  if (M <= 0 || cout <= 0 || cin <= 0 || pitch_dy < cout || pitch_x < cin) return XPT_ERR_SHAPE;
  const WgradPlan p = wgrad_plan(M, cout, cin, true);
  if (partial_floats < (size_t)p.nsplit * cout * cin) return XPT_ERR_WORKSPACE;
  if (p.tiles_co > 65535 || p.nsplit > 65535) return XPT_ERR_SHAPE;
  return wgrad_launch(dy, x, nullptr, partials, nullptr, M, cout, cin, pitch_dy, pitch_x, p, 1, stream);
}

/* The same partials plus, when dx is not NULL, the data gradient dx [M, cin] bf16 = dy W (w: bf16 [cout, cin], dense)
 * computed by extra workgroups of the same launch (a pointwise convolution WITHOUT a BatchNorm behind it). */
extern "C" int xpt_conv1x1_bwd_fused(const void* dy, const void* x, const void* w, void* dx, float* partials,
                                     size_t partial_floats, long long M, int cout, int cin, long long pitch_dy,
                                     long long pitch_x, void* stream) {
  XPT_CHECK_PTR(dy);
  XPT_CHECK_PTR(x);
  XPT_CHECK_PTR(partials);
  if (dx != nullptr && w == nullptr) return XPT_ERR_NULL;
  if (M <= 0 || cout <= 0 || cin <= 0 || pitch_dy < cout || pitch_x < cin) return XPT_ERR_SHAPE;
  const WgradPlan p = wgrad_plan(M, cout, cin, true);
  if (partial_floats < (size_t)p.nsplit * cout * cin) return XPT_ERR_WORKSPACE;
  if (p.tiles_co > 65535 || p.nsplit > 65535) return XPT_ERR_SHAPE;
  DxFuse d{};
  if (dx) {
    d.n = 1;
    d.dx[0] = (unsigned short*)dx;
    d.w[0] = (const unsigned short*)w;
  }
  return wgrad_launch(dy, x, nullptr, partials, nullptr, M, cout, cin, pitch_dy, pitch_x, p, 1, stream, nullptr, nullptr, &d);
}

/* conv -> BatchNorm backward in one launch (see BnFuse): dy is the gradient of the BN output.
 *   g_out [M, cout] bf16 = dy * gamma * rsqrt(var + eps)      (input of the data-gradient GEMM)
 *   w_partials [splits][cout][cin]                            filter-gradient partials, D = g^T x
 *   bn_partials [splits][2][cout]                             row 0: dbeta partials, row 1: dgamma partials
 * splits = xpt_conv1x1_bwd_weight_splits(M, cout, cin); all partials are finished by xpt_reduce_partials. */
extern "C" int xpt_conv1x1_bn_bwd_partials_sum(const void* dy, const void* dy2, const void* dy3, const void* ypre,
                                               const void* x, const float* gamma, const float* var, const float* mean,
                                               float eps, void* g_out, float* w_partials, size_t w_partial_floats,
                                               float* bn_partials, size_t bn_partial_floats, long long M, int cout,
                                               int cin, long long pitch_dy, long long pitch_dy2, long long pitch_dy3,
                                               long long pitch_x, void* stream) {
  XPT_CHECK_PTR(dy); XPT_CHECK_PTR(ypre); XPT_CHECK_PTR(x); XPT_CHECK_PTR(gamma); XPT_CHECK_PTR(var);
  XPT_CHECK_PTR(mean); XPT_CHECK_PTR(g_out); XPT_CHECK_PTR(w_partials); XPT_CHECK_PTR(bn_partials);
  if (M <= 0 || cout <= 0 || cin <= 0 || pitch_dy < cout || pitch_x < cin) return XPT_ERR_SHAPE;
  if (dy3 != nullptr && dy2 == nullptr) return XPT_ERR_NULL;
  if ((dy2 && pitch_dy2 < cout) || (dy3 && pitch_dy3 < cout)) return XPT_ERR_SHAPE;
  const WgradPlan p = wgrad_plan(M, cout, cin, true);
  if (w_partial_floats < (size_t)p.nsplit * cout * cin) return XPT_ERR_WORKSPACE;
  if (bn_partial_floats < (size_t)p.nsplit * 2 * cout) return XPT_ERR_WORKSPACE;
  if (p.tiles_co > 65535 || p.nsplit > 65535) return XPT_ERR_SHAPE;
  BnFuse bn{gamma, var, mean, eps, (const unsigned short*)ypre, (unsigned short*)g_out, bn_partials,
            {(const unsigned short*)dy2, (const unsigned short*)dy3}, {pitch_dy2, pitch_dy3}, dy3 ? 2 : (dy2 ? 1 : 0)};
  return wgrad_launch(dy, x, nullptr, w_partials, nullptr, M, cout, cin, pitch_dy, pitch_x, p, 1, stream, &bn);
}

/* The whole backward of conv1x1 -> BatchNorm in ONE launch: the partials above plus, when dx is not NULL, the data
 * gradient dx [M, cin] bf16 = (dy (+ dy2 + dy3)) * s  W computed by extra workgroups of the same launch (w: the layer's
 * bf16 weight [cout, cin], dense).  g is not written. */
extern "C" int xpt_conv1x1_bn_bwd_fused(const void* dy, const void* dy2, const void* dy3, const void* ypre, const void* x,
                                        const void* w, const float* gamma, const float* var, const float* mean, float eps,
                                        void* dx, float* w_partials, size_t w_partial_floats, float* bn_partials,
                                        size_t bn_partial_floats, long long M, int cout, int cin, long long pitch_dy,
                                        long long pitch_dy2, long long pitch_dy3, long long pitch_x, void* stream) {
  XPT_CHECK_PTR(dy); XPT_CHECK_PTR(ypre); XPT_CHECK_PTR(x); XPT_CHECK_PTR(gamma); XPT_CHECK_PTR(var);
  XPT_CHECK_PTR(mean); XPT_CHECK_PTR(w_partials); XPT_CHECK_PTR(bn_partials);
  if (dx != nullptr && w == nullptr) return XPT_ERR_NULL;
  if (M <= 0 || cout <= 0 || cin <= 0 || pitch_dy < cout || pitch_x < cin) return XPT_ERR_SHAPE;
  if (dy3 != nullptr && dy2 == nullptr) return XPT_ERR_NULL;
  if ((dy2 && pitch_dy2 < cout) || (dy3 && pitch_dy3 < cout)) return XPT_ERR_SHAPE;
  const WgradPlan p = wgrad_plan(M, cout, cin, true);
  if (w_partial_floats < (size_t)p.nsplit * cout * cin) return XPT_ERR_WORKSPACE;
  if (bn_partial_floats < (size_t)p.nsplit * 2 * cout) return XPT_ERR_WORKSPACE;
  if (p.tiles_co > 65535 || p.nsplit > 65535) return XPT_ERR_SHAPE;
  BnFuse bn{gamma, var, mean, eps, (const unsigned short*)ypre, nullptr, bn_partials,
            {(const unsigned short*)dy2, (const unsigned short*)dy3}, {pitch_dy2, pitch_dy3}, dy3 ? 2 : (dy2 ? 1 : 0)};
  DxFuse d{};
  if (dx) {
    d.n = 1;
    d.dx[0] = (unsigned short*)dx;
    d.w[0] = (const unsigned short*)w;
  }
  return wgrad_launch(dy, x, nullptr, w_partials, nullptr, M, cout, cin, pitch_dy, pitch_x, p, 1, stream, &bn, nullptr, &d);
}

extern "C" int xpt_conv1x1_bn_bwd_partials(const void* dy, const void* ypre, const void* x, const float* gamma,
                                           const float* var, const float* mean, float eps, void* g_out,
                                           float* w_partials, size_t w_partial_floats, float* bn_partials,
                                           size_t bn_partial_floats, long long M, int cout, int cin,
                                           long long pitch_dy, long long pitch_x, void* stream) {
  return xpt_conv1x1_bn_bwd_partials_sum(dy, nullptr, nullptr, ypre, x, gamma, var, mean, eps, g_out, w_partials,
                                         w_partial_floats, bn_partials, bn_partial_floats, M, cout, cin, pitch_dy, 0, 0,
                                         pitch_x, stream);
}

/* The same for n (<= 6) layers of one shape (M, cout, cin, pitch_x) in one launch; arrays of n pointers, pitch_dy per
 * layer (the incoming gradients may be channel slices of different concatenations). */
extern "C" int xpt_conv1x1_bn_multi_bwd_partials(int n, const void* const* dy, const long long* pitch_dy,
                                                 const void* const* ypre, const void* const* x,
                                                 const float* const* gamma, const float* const* var,
                                                 const float* const* mean, float eps, void* const* g_out,
                                                 float* const* w_partials, float* const* bn_partials,
                                                 size_t w_partial_floats, size_t bn_partial_floats, long long M,
                                                 int cout, int cin, long long pitch_x, void* stream) {
  XPT_CHECK_PTR(dy); XPT_CHECK_PTR(pitch_dy); XPT_CHECK_PTR(ypre); XPT_CHECK_PTR(x); XPT_CHECK_PTR(gamma);
  XPT_CHECK_PTR(var); XPT_CHECK_PTR(mean); XPT_CHECK_PTR(g_out); XPT_CHECK_PTR(w_partials); XPT_CHECK_PTR(bn_partials);
  if (n < 1 || n > WG_MAX_JOBS) return XPT_ERR_ARG;
  if (M <= 0 || cout <= 0 || cin <= 0 || pitch_x < cin) return XPT_ERR_SHAPE;
  const WgradPlan p = wgrad_plan(M, cout, cin, true);
  if (w_partial_floats < (size_t)p.nsplit * cout * cin || bn_partial_floats < (size_t)p.nsplit * 2 * cout)
    return XPT_ERR_WORKSPACE;
  if (p.tiles_co > 65535 || (long long)p.nsplit * n > 65535) return XPT_ERR_SHAPE;
  WgradMulti m{};
  m.n = n;
  for (int j = 0; j < n; ++j) {
    if (!dy[j] || !ypre[j] || !x[j] || !gamma[j] || !var[j] || !mean[j] || !g_out[j] || !w_partials[j] || !bn_partials[j])
      return XPT_ERR_NULL;
    if (pitch_dy[j] < cout) return XPT_ERR_SHAPE;
    m.dy[j] = (const unsigned short*)dy[j];
    m.x[j] = (const unsigned short*)x[j];
    m.partial[j] = w_partials[j];
    m.pitch_dy[j] = pitch_dy[j];
    m.bn[j] = BnFuse{gamma[j], var[j], mean[j], eps, (const unsigned short*)ypre[j], (unsigned short*)g_out[j],
                     bn_partials[j]};
  }
  return wgrad_launch(dy[0], x[0], nullptr, w_partials[0], nullptr, M, cout, cin, pitch_dy[0], pitch_x, p, 1, stream,
                      &m.bn[0], &m);
}

static int multi_bwd_fused(int n, const void* const* dy, const long long* pitch_dy, const void* const* dy2,
                           const long long* pitch_dy2, const void* const* dy3, const long long* pitch_dy3,
                           const void* const* ypre, const void* const* x, const void* const* w, const float* const* gamma,
                           const float* const* var, const float* const* mean, float eps, void* const* dx,
                           float* const* w_partials, float* const* bn_partials, size_t w_partial_floats,
                           size_t bn_partial_floats, long long M, int cout, int cin, long long pitch_x, void* stream) {
  if (n < 1 || n > WG_MAX_JOBS) return XPT_ERR_ARG;
  if (M <= 0 || cout <= 0 || cin <= 0 || pitch_x < cin) return XPT_ERR_SHAPE;
  const WgradPlan p = wgrad_plan(M, cout, cin, true);
  if (w_partial_floats < (size_t)p.nsplit * cout * cin || bn_partial_floats < (size_t)p.nsplit * 2 * cout)
    return XPT_ERR_WORKSPACE;
  if (p.tiles_co > 65535 || (long long)p.nsplit * n > 65535) return XPT_ERR_SHAPE;
  WgradMulti m{};
  DxFuse d{};
  m.n = n;
  for (int j = 0; j < n; ++j) {
    if (!dy[j] || !ypre[j] || !x[j] || !gamma[j] || !var[j] || !mean[j] || !w_partials[j] || !bn_partials[j])
      return XPT_ERR_NULL;
    if (dx[j] && !w[j]) return XPT_ERR_NULL;
    if (pitch_dy[j] < cout) return XPT_ERR_SHAPE;
    const void* e1 = dy2 ? dy2[j] : nullptr;
    const void* e2 = dy3 ? dy3[j] : nullptr;
    if (e2 && !e1) return XPT_ERR_NULL;
    if ((e1 && (!pitch_dy2 || pitch_dy2[j] < cout)) || (e2 && (!pitch_dy3 || pitch_dy3[j] < cout))) return XPT_ERR_SHAPE;
    m.dy[j] = (const unsigned short*)dy[j];
    m.x[j] = (const unsigned short*)x[j];
    m.partial[j] = w_partials[j];
    m.pitch_dy[j] = pitch_dy[j];
    m.bn[j] = BnFuse{gamma[j], var[j], mean[j], eps, (const unsigned short*)ypre[j], nullptr, bn_partials[j],
                     {(const unsigned short*)e1, (const unsigned short*)e2},
                     {e1 ? pitch_dy2[j] : 0, e2 ? pitch_dy3[j] : 0}, e2 ? 2 : (e1 ? 1 : 0)};
    if (dx[j]) {
      d.dx[d.n] = (unsigned short*)dx[j];
      d.w[d.n] = (const unsigned short*)w[j];
      d.job[d.n] = j;
      ++d.n;
    }
  }
  return wgrad_launch(dy[0], x[0], nullptr, w_partials[0], nullptr, M, cout, cin, pitch_dy[0], pitch_x, p, 1, stream,
                      &m.bn[0], &m, &d);
}

/* ... and with the data gradients dx[j] [M, cin] bf16 = (dy_j * s_j) W_j of the layers whose dx[j] is not NULL computed
 * by extra workgroups of the same launch (w[j]: bf16 weights [cout, cin], dense); g is not written. */
extern "C" int xpt_conv1x1_bn_multi_bwd_fused(int n, const void* const* dy, const long long* pitch_dy,
                                              const void* const* ypre, const void* const* x, const void* const* w,
                                              const float* const* gamma, const float* const* var,
                                              const float* const* mean, float eps, void* const* dx,
                                              float* const* w_partials, float* const* bn_partials,
                                              size_t w_partial_floats, size_t bn_partial_floats, long long M, int cout,
                                              int cin, long long pitch_x, void* stream) {
  XPT_CHECK_PTR(dy); XPT_CHECK_PTR(pitch_dy); XPT_CHECK_PTR(ypre); XPT_CHECK_PTR(x); XPT_CHECK_PTR(w); XPT_CHECK_PTR(gamma);
  XPT_CHECK_PTR(var); XPT_CHECK_PTR(mean); XPT_CHECK_PTR(dx); XPT_CHECK_PTR(w_partials); XPT_CHECK_PTR(bn_partials);
  return multi_bwd_fused(n, dy, pitch_dy, nullptr, nullptr, nullptr, nullptr, ypre, x, w, gamma, var, mean, eps, dx,
                         w_partials, bn_partials, w_partial_floats, bn_partial_floats, M, cout, cin, pitch_x, stream);
}

/* The same with gradient FAN-IN per layer: the output gradient of layer j is dy[j] + dy2[j] + dy3[j] (the layer's output
 * feeds up to three consumers of the cell; dy2[j] / dy3[j] may be NULL, the arrays themselves too), added on load in fp32 and
 * rounded to bf16 once -- what a separate xpt_sum_rows launch in front would produce. */
extern "C" int xpt_conv1x1_bn_multi_bwd_fused_fan(int n, const void* const* dy, const long long* pitch_dy,
                                                  const void* const* dy2, const long long* pitch_dy2,
                                                  const void* const* dy3, const long long* pitch_dy3,
                                                  const void* const* ypre, const void* const* x, const void* const* w,
                                                  const float* const* gamma, const float* const* var,
                                                  const float* const* mean, float eps, void* const* dx,
                                                  float* const* w_partials, float* const* bn_partials,
                                                  size_t w_partial_floats, size_t bn_partial_floats, long long M,
                                                  int cout, int cin, long long pitch_x, void* stream) {
  XPT_CHECK_PTR(dy); XPT_CHECK_PTR(pitch_dy); XPT_CHECK_PTR(ypre); XPT_CHECK_PTR(x); XPT_CHECK_PTR(w); XPT_CHECK_PTR(gamma);
  XPT_CHECK_PTR(var); XPT_CHECK_PTR(mean); XPT_CHECK_PTR(dx); XPT_CHECK_PTR(w_partials); XPT_CHECK_PTR(bn_partials);
  return multi_bwd_fused(n, dy, pitch_dy, dy2, pitch_dy2, dy3, pitch_dy3, ypre, x, w, gamma, var, mean, eps, dx, w_partials,
                         bn_partials, w_partial_floats, bn_partial_floats, M, cout, cin, pitch_x, stream);
}
