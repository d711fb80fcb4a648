// xpt_conv_wgrad.hip -- weight gradient of the dense k x k convolutions (PoseNetImproved, depth decoder) on the gfx950
// bf16 matrix cores:   dW[n][kh][kw][c] = sum over output pixels m of  g[m][n] * x[pixel(m, kh, kw)][c]
// (tape.gradient of the Conv2D kernels built by CustomConv2D, model/model_util/layer_ops.py:5-36; model/train_val.py:85-86).
//
// The reduction index (pixels) is the SLOW axis of both NHWC operands, the opposite of what an MFMA fragment wants
// (8 consecutive k per lane).  The tiles are therefore staged row-major in LDS exactly as they lie in memory (16-byte
// global loads along the channels) and read back TRANSPOSED with ds_read_b64_tr_b16: per 16-lane group the instruction
// takes 4 pixel rows x 16 channel columns and hands lane i the 4 pixels of channel column i -- two reads give a lane
// the 8 consecutive pixels of its channel that v_mfma_f32_32x32x16_bf16 expects, for A (= g^T, rows = output channels)
// and for B (= x, columns = input channels).  Every lane supplies the address of one pixel row, so the shifted
// windows of the taps and the stride-2 input grid are plain address arithmetic on ONE staged halo tile of x: the input
// is read from HBM/L2 once per unit, not once per tap.
//
// Work decomposition.  A workgroup (4 waves) owns a TNB x TCB (output x input channel) block of dW for all taps and
// loops over its share of the pixel UNITS (R output rows x CB columns of one image); the waves split the block into
// 32 x 32 sub-tiles (WN x WC) and, when the block is small, the taps (WT groups of TAPS).  The workgroup's sum over its
// units is written as one fp32 partial (row-major like the parameter, [N][KH][KW][Cr]); the per-step finishing launch
// (xpt_reduce_partials) adds the partials of all splits into the flat gradient buffer in a fixed order.
#include "xpt_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int WGRAD_MAXCH = 12;     // 16-byte chunks a thread stages per pixel unit (48 KiB of LDS tiles at most)

struct WgradArgs {
  const unsigned short* g;   // [B,OH,OW,N] bf16, pixel pitch gpitch
  const unsigned short* x;   // [B,PH,PW,C] bf16, pixel pitch xpitch (C = channels readable per pixel, multiple of 8)
  float* part;               // [nsplit][N][T][Cr] fp32
  long long gpitch, xpitch, gbytes, xbytes;
  int B, PH, PW, shift, Hlim, Wlim;
  int C, Cr, N, KH, KW, stride, pad_t, pad_l, OH, OW;
  int R, CB, nseg, nrb, units, nsplit;
  int WN, WC, WT;
  int dbg;                   // lab knobs (xpt_conv2d_bwd_weight_tune(-bits, 0)): 1 no products, 2 no refetch, 4 no store
};

__device__ inline bf16x8 tr_pair(const char* lds, unsigned off0, unsigned off1) {
  typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(lds + off0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(lds + off1));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// the 16-byte chunks of one unit's g tile and x halo tile, chunk ck = tid + 256 i, into registers
// Geometry of this thread's staging chunks, computed ONCE per workgroup: chunk ck = tid + 256 i is either 8 output channels
// of one pixel of the g tile or 8 input channels of one pixel of the x halo tile; only the unit's origin changes from unit
// to unit.  (The chunk -> (pixel, channel group) splits are shifts -- TNB / 8 and TCB / 8 are 4 or 8 --, the pixel ->
// (row, column) splits exact float multiplications: with six run-time integer divisions per chunk, ~40 instructions
// each, the staging function was most of the kernel; ablation with the lab knobs: products 7 us, refetch 6 us, store 4 us,
// the rest 9 us of a 26 us launch.)
template <int MAXCH>
struct WgradChunks {
  int row[MAXCH], col[MAXCH];    // pixel of the chunk inside its tile (g: output pixel; x: logical halo pixel)
  unsigned chb[MAXCH];           // byte offset of the chunk's first channel (g: n0 + 8 gc8, x: c0 + 8 xc8); WG_OOB: no chunk
};

// Round 4: staging slot i of every thread is a g chunk for i < gi and an x chunk behind that (the g chunks are padded to whole
// passes of 256), so that each slot reads ONE operand through ONE buffer descriptor: the range check of the buffer load
// supplies the zeros of the padding, of the ragged tiles and of the dead chunks (no masks on the loaded vectors, 32-bit
// offsets instead of 64-bit address arithmetic).
constexpr unsigned WG_OOB = 0x40000000u;

template <int MAXCH>
__device__ __forceinline__ void wgrad_chunks(const WgradArgs& a, int tid, int n0, int c0, int TNB, int TCB, int Wt,
                                             int gi, int gchunks, int xchunks, WgradChunks<MAXCH>& w) {
  const int lg = __builtin_ctz((unsigned)(TNB >> 3)), lx = __builtin_ctz((unsigned)(TCB >> 3));      // TNB, TCB: 32 x {1, 2, 4}
  const float inv_cb = 1.f / (float)a.CB, inv_wt = 1.f / (float)Wt;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    if (i < gi) {                                              // (uniform)
      const int ck = tid + 256 * i;
      const int gc8 = ck & ((1 << lg) - 1), gpix = ck >> lg;
      const int gr = (int)(((float)gpix + 0.5f) * inv_cb);
      w.row[i] = gr;
      w.col[i] = gpix - gr * a.CB;
      const int n = n0 + 8 * gc8;
      w.chb[i] = (ck < gchunks && n < a.N) ? (unsigned)n * 2u : WG_OOB;
    } else {
      const int cx = tid + 256 * (i - gi);
      const int xc8 = cx & ((1 << lx) - 1), xpix = cx >> lx;
      const int xr = (int)(((float)xpix + 0.5f) * inv_wt);
      w.row[i] = xr;
      w.col[i] = xpix - xr * Wt;
      const int c = c0 + 8 * xc8;
      w.chb[i] = (cx < xchunks && c < a.C) ? (unsigned)c * 2u : WG_OOB;
    }
  }
}

// the 16-byte chunks of one unit's g tile and x halo tile into registers
template <int MAXCH>
__device__ __forceinline__ void wgrad_fetch(const WgradArgs& a, int u, int gi, const WgradChunks<MAXCH>& w,
                                            const __amdgpu_buffer_rsrc_t& rg, const __amdgpu_buffer_rsrc_t& rx, u32x4 (&stage)[MAXCH]) {
  unsigned seg, rb;
  const unsigned q1 = xpt_divmod((unsigned)u, (unsigned)a.nseg, seg);
  const unsigned b = xpt_divmod(q1, (unsigned)a.nrb, rb);      // (uniform)
  const int oh0 = (int)rb * a.R, ow0 = (int)seg * a.CB;
  const int th0 = oh0 * a.stride - a.pad_t, tw0 = ow0 * a.stride - a.pad_l;
  const unsigned gimg = b * (unsigned)(a.OH * a.OW), ximg = b * (unsigned)(a.PH * a.PW);
  const unsigned gp2 = (unsigned)(a.gpitch * 2), xp2 = (unsigned)(a.xpitch * 2);
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    if (i < gi) {                                              // (uniform)
      const int r = oh0 + w.row[i], c = ow0 + w.col[i];
      const bool ok = r < a.OH && c < a.OW;
      const unsigned off = ok ? (gimg + (unsigned)(r * a.OW + c)) * gp2 + w.chb[i] : WG_OOB;
      stage[i] = __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0);
    } else {
      const int r = th0 + w.row[i], c = tw0 + w.col[i];
      const bool ok = r >= 0 && c >= 0 && r < a.Hlim && c < a.Wlim;
      const unsigned off = ok ? (ximg + (unsigned)((r >> a.shift) * a.PW + (c >> a.shift))) * xp2 + w.chb[i] : WG_OOB;
      stage[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
    }
  }
}

template <int TAPS>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int TNB = 32 * a.WN, TCB = 32 * a.WC;
  const int T = a.KH * a.KW;
  const int ntc = (a.C + TCB - 1) / TCB;
  const int n0 = (blockIdx.x / ntc) * TNB, c0 = (blockIdx.x % ntc) * TCB;
  const int HR = (a.R - 1) * a.stride + a.KH, Wt = (a.CB - 1) * a.stride + a.KW;
  const int gchunks = a.R * a.CB * (TNB / 8), xchunks = HR * Wt * (TCB / 8);
  const int gi = (gchunks + 255) >> 8;                         // staging slots that hold g chunks (whole passes of 256)
  char* gs = smem;                                             // [R][CB][TNB] bf16
  char* xs = smem + (size_t)gi * 256 * 16;                     // [HR][Wt][TCB] bf16, behind the padded g passes

  // this wave's sub-tile and taps
  const int wt = wave % a.WT, wc = (wave / a.WT) % a.WC, wn = wave / (a.WT * a.WC);
  const int tap0 = wt * TAPS;
  const bool active = tap0 < T;                                // wave-uniform
  // lane roles of the transposed reads
  const int g4 = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3, h = g4 >> 1, half = g4 & 1;
  unsigned a_base[2], b_base[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pc = 8 * h + 4 * i + q;                          // pixel column of this lane's row, read i
    a_base[i] = (unsigned)((pc * TNB + wn * 32 + 16 * half + 4 * p) * 2);
    b_base[i] = (unsigned)((pc * a.stride * TCB + wc * 32 + 16 * half + 4 * p) * 2);
  }
  unsigned toff[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    const int tap = tap0 + t < T ? tap0 + t : 0;
    toff[t] = (unsigned)(((tap / a.KW) * Wt + (tap % a.KW)) * TCB * 2);
  }

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // Staging is software-pipelined: the 16-byte chunks of unit u+1 are fetched into registers while unit u is being
  // multiplied (the loads stay in flight across the MFMA loop), then written to LDS behind the barrier that retires
  // unit u's reads.  MAXCH chunks per thread bound the tile (the plan keeps gchunks + xchunks <= 256 * MAXCH).
  constexpr int MAXCH = WGRAD_MAXCH;
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.g, 0, (int)a.gbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.xbytes, 0x00020000);
  u32x4 stage[MAXCH];
  WgradChunks<MAXCH> chunks;
  wgrad_chunks<MAXCH>(a, tid, n0, c0, TNB, TCB, Wt, gi, gchunks, xchunks, chunks);
  int u = blockIdx.y;
  if (u < a.units) wgrad_fetch<MAXCH>(a, u, gi, chunks, rg, rx, stage);
  for (; u < a.units; u += a.nsplit) {
    __syncthreads();                                           // the previous unit's LDS reads are done
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int ck = tid + 256 * i;                            // slot order = LDS order (g passes padded to 256 chunks)
      if (i < gi ? ck < gchunks : ck - gi * 256 < xchunks) *(u32x4*)(smem + (size_t)ck * 16) = stage[i];
    }
    __syncthreads();
    if (u + a.nsplit < a.units && !(a.dbg & 2))                // in flight during the products below
      wgrad_fetch<MAXCH>(a, u + a.nsplit, gi, chunks, rg, rx, stage);
    if (active && !(a.dbg & 1)) {
      for (int rr = 0; rr < a.R; ++rr) {
        for (int c16 = 0; c16 < a.CB; c16 += 16) {
          const unsigned ga = (unsigned)((rr * a.CB + c16) * TNB * 2);
          const unsigned xa = (unsigned)(((rr * a.stride) * Wt + c16 * a.stride) * TCB * 2);
          const bf16x8 fa = tr_pair(gs, a_base[0] + ga, a_base[1] + ga);
#pragma unroll
          for (int t = 0; t < TAPS; ++t) {
            if (tap0 + t < T) {                                // wave-uniform
              const bf16x8 fb = tr_pair(xs, b_base[0] + xa + toff[t], b_base[1] + xa + toff[t]);
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[t], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  // partial of this split: register e of tap t -> output channel n0 + 32 wn + (e & 3) + 8 (e >> 2) + 4 h,
  // input channel c0 + 32 wc + (lane & 31)
  if (!active || (a.dbg & 4)) return;
  const int c = c0 + wc * 32 + (lane & 31);
  float* dst = a.part + (long long)blockIdx.y * a.N * T * a.Cr;
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    if (tap0 + t >= T) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = n0 + wn * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (n < a.N && c < a.Cr) dst[((long long)n * T + tap0 + t) * a.Cr + c] = acc[t][e];
    }
  }
}

struct WgradPlan {
  int WN, WC, WT, TAPS, R, CB, nseg, nrb, units, nsplit;
  long long blocks;
  size_t lds;
};


int g_wgrad_max_partial_mib = 12, g_wgrad_target_blocks = 512, g_wgrad_dbg = 0;


// one candidate decomposition: sub-tiles (WN x WC), taps split WT ways; pixel units bounded by the staging registers
bool plan_for(int wn, int wc, int B, int C, int N, int KH, int KW, int stride, int OH, int OW, WgradPlan& p) {
  const int T = KH * KW;
  p.WN = wn; p.WC = wc;
  p.WT = 4 / (wn * wc);
  const int taps = (T + p.WT - 1) / p.WT;
  const int allowed[] = {1, 3, 5, 7, 9};
  p.TAPS = 0;
  for (int v : allowed)
    if (v >= taps) { p.TAPS = v; break; }
  if (!p.TAPS) return false;
  const int TNB = 32 * wn, TCB = 32 * wc;
  const size_t max_lds = (size_t)WGRAD_MAXCH * 256 * 16;       // every thread stages at most WGRAD_MAXCH 16-byte chunks
  const int ow16 = (OW + 15) / 16 * 16;
  p.CB = ow16 < 64 ? ow16 : 64;
  p.nseg = (OW + p.CB - 1) / p.CB;
  p.R = 256 / p.CB;
  if (p.R < 1) p.R = 1;
  if (p.R > OH) p.R = OH;
  for (;;) {
    const int HR = (p.R - 1) * stride + KH, Wt = (p.CB - 1) * stride + KW;
    const size_t gch = (size_t)p.R * p.CB * (TNB / 8), xch = (size_t)HR * Wt * (TCB / 8);
    const size_t slots = (gch + 255) / 256 + (xch + 255) / 256;          // staging slots per thread: g passes, then x passes
    p.lds = ((gch + 255) / 256 * 256 + xch) * 16;
    if (slots <= (size_t)WGRAD_MAXCH && p.lds <= max_lds) break;
    if (p.R > 1) p.R -= 1;
    else if (p.CB > 16) { p.CB -= 16; p.nseg = (OW + p.CB - 1) / p.CB; }
    else return false;
  }
  p.nrb = (OH + p.R - 1) / p.R;
  p.units = B * p.nrb * p.nseg;
  const long long tiles = (long long)((N + TNB - 1) / TNB) * ((C + TCB - 1) / TCB);
  long long ns = (g_wgrad_target_blocks + tiles - 1) / tiles;
  const long long bytes = (long long)N * T * C * 4;
  const long long cap = ((long long)g_wgrad_max_partial_mib << 20) / (bytes > 0 ? bytes : 1);
  if (ns > cap) ns = cap;
  if (ns > p.units) ns = p.units;
  if (ns < 1) ns = 1;
  p.nsplit = (int)ns;
  p.blocks = tiles * ns;
  return true;
}

// The kernels are latency-bound at these sizes, so the plan maximises the number of workgroups (up to the target) under
// the cap on partial-sum bytes: large 64 x 64 blocks when they already fill the chip, else smaller blocks (more of them
// for the same partial volume), the taps spread over the waves instead.
bool make_plan(int B, int C, int N, int KH, int KW, int stride, int OH, int OW, WgradPlan& best) {
  const int cand[4][2] = {{2, 2}, {2, 1}, {1, 2}, {1, 1}};
  bool have = false;
  for (const auto& wc : cand) {
    if ((wc[0] == 2 && N <= 32) || (wc[1] == 2 && C <= 32)) continue;
    if (KH * KW > 9 && wc[0] * wc[1] > 1) continue;              // 5x5: 7 taps per wave at most
    WgradPlan p;
    if (!plan_for(wc[0], wc[1], B, C, N, KH, KW, stride, OH, OW, p)) continue;
    if (!have || p.blocks > best.blocks) { best = p; have = true; }
    if (best.blocks * 4 >= (long long)g_wgrad_target_blocks * 3) break;   // good enough: keep the larger block
  }
  return have;
}

}  // namespace

extern "C" int xpt_conv2d_bwd_weight_tune(int max_partial_mib, int target_blocks) {
  if (max_partial_mib < 0) g_wgrad_dbg = -max_partial_mib;          // lab knobs, see WgradArgs::dbg
  if (max_partial_mib > 0) g_wgrad_max_partial_mib = max_partial_mib;
  if (target_blocks > 0) g_wgrad_target_blocks = target_blocks;
  return XPT_OK;
}

extern "C" int xpt_conv2d_bwd_weight_splits(int B, int C, int N, int KH, int KW, int stride, int OH, int OW) {
  WgradPlan p;
  if (B <= 0 || C <= 0 || N <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || OH <= 0 || OW <= 0) return XPT_ERR_SHAPE;
  if (!make_plan(B, C, N, KH, KW, stride, OH, OW, p)) return XPT_ERR_ARG;
  return p.nsplit;
}

/* partials [splits][N][KH][KW][Cr] fp32 of dW = sum_m g[m][n] x[pixel(m,kh,kw)][c]; x has PH x PW physical pixels (the
 * taps index its nearest-2x up-sampling when upsample = 1) of C readable channels (multiple of 8) of which the first Cr
 * are real weight channels; g has N channels (multiple of 8). */
extern "C" int xpt_conv2d_bwd_weight_partials(const void* g, const void* x, float* partials, size_t partial_floats, int B,
                                              int PH, int PW, int C, int Cr, long long xpitch, int N, long long gpitch,
                                              int KH, int KW, int stride, int pad_t, int pad_l, int OH, int OW,
                                              int upsample, void* stream) {
  XPT_CHECK_PTR(g); XPT_CHECK_PTR(x); XPT_CHECK_PTR(partials);
  if (B <= 0 || PH <= 0 || PW <= 0 || C <= 0 || Cr <= 0 || Cr > C || N <= 0 || KH <= 0 || KW <= 0 || OH <= 0 || OW <= 0)
    return XPT_ERR_SHAPE;
  if (C % 8 != 0 || N % 8 != 0 || xpitch < C || xpitch % 8 != 0 || gpitch < N || gpitch % 8 != 0 ||
      ((uintptr_t)g) % 16 != 0 || ((uintptr_t)x) % 16 != 0 || stride < 1 || (upsample != 0 && upsample != 1))
    return XPT_ERR_ARG;
  WgradPlan p;
  if (!make_plan(B, C, N, KH, KW, stride, OH, OW, p)) return XPT_ERR_ARG;
  if (partial_floats < (size_t)p.nsplit * N * KH * KW * Cr) return XPT_ERR_WORKSPACE;
  WgradArgs a{};
  a.g = (const unsigned short*)g; a.x = (const unsigned short*)x; a.part = partials;
  a.gpitch = gpitch; a.xpitch = xpitch;
  a.gbytes = ((long long)B * OH * OW - 1) * gpitch * 2 + (long long)N * 2;
  a.xbytes = ((long long)B * PH * PW - 1) * xpitch * 2 + (long long)C * 2;
  if (a.gbytes >= (1LL << 30) || a.xbytes >= (1LL << 30)) return XPT_ERR_SHAPE;      // 32-bit offsets + the 1 GiB out-of-range marker
  a.B = B; a.PH = PH; a.PW = PW; a.shift = upsample; a.Hlim = PH << upsample; a.Wlim = PW << upsample;
  a.C = C; a.Cr = Cr; a.N = N; a.KH = KH; a.KW = KW; a.stride = stride; a.pad_t = pad_t; a.pad_l = pad_l;
  a.OH = OH; a.OW = OW;
  a.R = p.R; a.CB = p.CB; a.nseg = p.nseg; a.nrb = p.nrb; a.units = p.units; a.nsplit = p.nsplit;
  a.WN = p.WN; a.WC = p.WC; a.WT = p.WT;
  a.dbg = g_wgrad_dbg;
  const int TNB = 32 * p.WN, TCB = 32 * p.WC;
  const dim3 grid(((N + TNB - 1) / TNB) * ((C + TCB - 1) / TCB), p.nsplit);
  hipStream_t s = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
#define XPT_WGRAD_CASE(TP)                                                                                         \
  case TP: {                                                                                                       \
    hipLaunchKernelGGL(conv_wgrad_kernel<TP>, grid, dim3(256), p.lds, s, a);                                       \
  } break;
  switch (p.TAPS) {
    XPT_WGRAD_CASE(1)
    XPT_WGRAD_CASE(3)
    XPT_WGRAD_CASE(5)
    XPT_WGRAD_CASE(7)
    XPT_WGRAD_CASE(9)
    default:
      return XPT_ERR_ARG;
  }
#undef XPT_WGRAD_CASE
  return xpt_launch_status();
}
