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
typedef xpt_h16x8 bf16x8;      // (8 operands of the build's 16-bit format, xpt_common.h)
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
              acc[t] = XPT_MFMA_32X32X16(fa, fb, acc[t]);
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

// ------------------------------------------------------------------------------------------------ specialised: 3 x 3, stride 1
// The weight gradients of the half- / full-resolution decoder layers (dp_up1 / dp_up0: <= 32 output channels on the 64 x 208
// and 128 x 416 maps) were 26 - 30 us each for 20 - 34 MB: PMC counts 2,339 vector + 1,096 scalar instructions per wave of
// conv_wgrad_kernel<3> -- the generic staging (12 unrolled slots with run-time geometry) and the run-time strides of the
// product loop, again the instruction stream.  Here, as in xpt_conv_stream.hip: persistent workgroups over 8 x 16-pixel
// tiles, ONE 32-channel column block of dW per workgroup (acc = 3 taps x 16 registers per wave: wave w owns taps w, w + 4,
// w + 8), the geometry of a thread's staged vectors computed once, interior tiles addressed as lane constant + scalar tile
// offset through buffer loads whose range check supplies the zeros, and a fully unrolled product loop whose LDS addresses are
// lane constants + instruction immediates (tiles start at even rows / columns, so the nearest-2x source row of an output row
// is a compile-time number as well).
struct WfArgs {
  const unsigned short* g;   // [B,OH,OW,N] bf16, pixel pitch gpitch
  const unsigned short* x;   // [B,PH,PW,C] bf16, pixel pitch xpitch
  float* part;               // [nsplit][N][9][Cr] fp32
  long long gpitch, xpitch, gbytes, xbytes;
  int B, PH, PW, OH, OW, C, Cr, N;
  int pad_t, pad_l, taps;
  int tiles_x, tiles_y, ntiles, nsplit, ct, nb; // ct / nb = 32-channel column / row blocks of dW (grid = ct x nb x nsplit)
  int xcd;
};

// K x K filter, stride S (TF SAME: the pads are arguments), UPS: nearest-2x input (3 x 3, stride 1 only)
template <int K, int S, bool UPS>
__global__ __launch_bounds__(256) void conv_wgrad_fast_kernel(WfArgs a) {
  constexpr int HR = UPS ? 6 : 7 * S + K, WR = UPS ? 10 : 15 * S + K, HPIX = HR * WR;
  constexpr int T = K * K, TPW = (T + 3) / 4;                  // taps; taps per wave (wave w owns taps w, w + 4, ...)
  __shared__ __attribute__((aligned(16))) char smem[8192 + HPIX * 64];
  constexpr int NG = 2, NX = (HPIX * 4 + 255) / 256;          // staged 16-byte vectors per thread: g tile, x halo block
  char* const gs = smem;                                       // [8 x 16 px][32 ch] bf16
  char* const xs = smem + 8192;                                // [HR x WR px][32 ch] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // workgroup -> (column block, row block, split).  a.xcd: the split is the fastest index (nsplit % 8 == 0: workgroup on XCD
  // split % 8, whose tiles split, split + nsplit, ... are dealt from that XCD's image-major eighth of the tiles, tile_of())
  int cblk, nblk, split;
  if (a.xcd) {
    unsigned sp_, cb_;
    const unsigned rest = xpt_divmod(blockIdx.x, (unsigned)a.nsplit, sp_);
    nblk = (int)xpt_divmod(rest, (unsigned)a.ct, cb_);
    cblk = (int)cb_;
    split = (int)sp_;
  } else {
    cblk = (int)blockIdx.x % a.ct;
    nblk = ((int)blockIdx.x / a.ct) % a.nb;
    split = (int)blockIdx.x / (a.ct * a.nb);
  }
  const int c0 = cblk * 32, n0 = nblk * 32;
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.g, 0, (int)a.gbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.xbytes, 0x00020000);
  const unsigned gp2 = (unsigned)(a.gpitch * 2), xp2 = (unsigned)(a.xpitch * 2);

  // ---- staging geometry, once
  int grow[NG], gcol[NG];
  unsigned gcb[NG], gconst[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int v = tid + 256 * i, pix = v >> 2, cv = v & 3;
    grow[i] = pix >> 4;
    gcol[i] = pix & 15;
    gcb[i] = n0 + cv * 8 < a.N ? (unsigned)(n0 + cv * 8) * 2u : WG_OOB;
    gconst[i] = (unsigned)(grow[i] * a.OW + gcol[i]) * gp2 + gcb[i];
  }
  int xq[NX], xrem[NX], xlds[NX];
  unsigned xcb[NX], xconst[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int v = tid + 256 * i, pix = v >> 2, cv = v & 3;
    const int q = pix / WR, rem = pix - q * WR;
    const bool live = pix < HPIX;
    xq[i] = q;
    xrem[i] = rem;
    xlds[i] = live ? pix * 64 + cv * 16 : -1;
    xcb[i] = (live && c0 + cv * 8 < a.C) ? (unsigned)(c0 + cv * 8) * 2u : WG_OOB;
    xconst[i] = (unsigned)(q * a.PW + rem) * xp2 + xcb[i];
  }

  // ---- lane roles of the transposed reads (as conv_wgrad_kernel): this lane supplies pixel column pc(i) of the 16-pixel step
  const int g4 = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3, h = g4 >> 1, half = g4 & 1;
  unsigned abase[2];
  int tkh[TPW];                                                // (scalars) tap row of slot tt, -1: no tap
  unsigned bsel[TPW][2];                                       // this lane's halo column offset under the slot's tap column
#pragma unroll
  for (int i = 0; i < 2; ++i) abase[i] = (unsigned)((8 * h + 4 * i + q4) * 64 + (16 * half + 4 * p4) * 2);
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int tap = wave + 4 * tt;
    const int kh = tap / K, kw = tap - K * kh;
    tkh[tt] = tap < T ? kh : -1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = 8 * h + 4 * i + q4;
      const int col = UPS ? ((pc + kw - 1) >> 1) + 1 : pc * S + kw;        // halo column of output column pc under tap column kw
      bsel[tt][i] = (unsigned)(8192 + col * 64 + (16 * half + 4 * p4) * 2);
    }
  }
  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  auto tile_of = [&](int t, int& b, int& oh0, int& ow0) {
    unsigned u = (unsigned)t;
    if (a.xcd) u = (u & 7u) * ((unsigned)a.ntiles >> 3) + (u >> 3);
    unsigned txi, tyi;
    const unsigned q1 = xpt_divmod(u, (unsigned)a.tiles_x, txi);
    b = __builtin_amdgcn_readfirstlane((int)xpt_divmod(q1, (unsigned)a.tiles_y, tyi));
    oh0 = __builtin_amdgcn_readfirstlane((int)tyi * 8);
    ow0 = __builtin_amdgcn_readfirstlane((int)txi * 16);
  };
  u32x4 sg[NG], sx[NX];
  auto fetch = [&](int t) {
    int b, oh0, ow0;
    tile_of(t, b, oh0, ow0);
    const int plo_h = UPS ? (oh0 - 1) >> 1 : oh0 * S - a.pad_t, plo_w = UPS ? (ow0 - 1) >> 1 : ow0 * S - a.pad_l;
    const bool g_in = oh0 + 8 <= a.OH && ow0 + 16 <= a.OW;
    const bool x_in = plo_h >= 0 && plo_h + HR <= a.PH && plo_w >= 0 && plo_w + WR <= a.PW;      // (uniform)
    if (g_in) {
      const unsigned soff = (unsigned)((b * a.OH + oh0) * a.OW + ow0) * gp2;
#pragma unroll
      for (int i = 0; i < NG; ++i) sg[i] = __builtin_amdgcn_raw_buffer_load_b128(rg, gconst[i], soff, 0);
    } else {
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const int r = oh0 + grow[i], c = ow0 + gcol[i];
        const unsigned off = (r < a.OH && c < a.OW) ? (unsigned)((b * a.OH + r) * a.OW + c) * gp2 + gcb[i] : WG_OOB;
        sg[i] = __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0);
      }
    }
    if (x_in) {
      const unsigned soff = (unsigned)((b * a.PH + plo_h) * a.PW + plo_w) * xp2;
#pragma unroll
      for (int i = 0; i < NX; ++i) sx[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, xconst[i], soff, 0);
    } else {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int pr = plo_h + xq[i], pc = plo_w + xrem[i];
        const bool ok = (unsigned)pr < (unsigned)a.PH && (unsigned)pc < (unsigned)a.PW;
        const unsigned off = ok ? (unsigned)((b * a.PH + pr) * a.PW + pc) * xp2 + xcb[i] : WG_OOB;
        sx[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
      }
    }
  };

  // this workgroup's tiles: split, split + nsplit, ...
  int t = split;
  if (t < a.ntiles) fetch(t);
  for (; t < a.ntiles; t += a.nsplit) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NG; ++i) *(u32x4*)(gs + (tid + 256 * i) * 16) = sg[i];
#pragma unroll
    for (int i = 0; i < NX; ++i)
      if (xlds[i] >= 0) *(u32x4*)(xs + xlds[i]) = sx[i];
    __syncthreads();
    if (t + a.nsplit < a.ntiles) fetch(t + a.nsplit);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {                            // one 16-pixel reduction step per tile row
      const bf16x8 fa = tr_pair(smem, abase[0] + rr * 1024, abase[1] + rr * 1024);
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) {
        if (tkh[tt] >= 0) {                                     // (wave-uniform)
          const int hrow = UPS ? ((rr + tkh[tt] - 1) >> 1) + 1 : rr * S + tkh[tt];      // (scalar)
          const unsigned ro = (unsigned)(hrow * WR * 64);
          const bf16x8 fb = tr_pair(smem, bsel[tt][0] + ro, bsel[tt][1] + ro);
          acc[tt] = XPT_MFMA_32X32X16(fa, fb, acc[tt]);
        }
      }
    }
  }

  // ---- this workgroup's partial: register e of tap slot tt -> output channel n0 + (e & 3) + 8 (e >> 2) + 4 h, input channel c0 + lane & 31
  const int c = c0 + (lane & 31);
  float* dst = a.part + (long long)split * a.N * T * a.Cr;
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int tap = wave + 4 * tt;
    if (tap >= T) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = n0 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (n < a.N && c < a.Cr) dst[((long long)n * T + tap) * a.Cr + c] = acc[tt][e];
    }
  }
}

int g_wgrad_fast = 1;            // 0: conv_wgrad_kernel for every layer
int g_wgrad_fast_min_tiles = 1;

// plan of the specialised kernel; false: the layer is outside what it serves
bool fast_plan(int B, int C, int N, int KH, int KW, int stride, int OH, int OW, int upsample, WfArgs& a) {
  if (!g_wgrad_fast || KH != KW || !((KH == 3 && (stride == 1 || stride == 2)) || (KH == 5 && stride == 2))) return false;
  if (upsample && (KH != 3 || stride != 1)) return false;
  (void)upsample;                                              // (a nearest-2x input makes OH, OW even by construction)
  a.tiles_x = (OW + 15) / 16;
  a.tiles_y = (OH + 7) / 8;
  const long long nt = (long long)B * a.tiles_y * a.tiles_x;
  if (nt < g_wgrad_fast_min_tiles || nt > 0x3fffffffLL) return false;
  a.ntiles = (int)nt;
  a.ct = (C + 31) / 32;
  a.nb = (N + 31) / 32;
  a.taps = KH * KW;
  return true;
}

struct WgradPlan {
  int WN, WC, WT, TAPS, R, CB, nseg, nrb, units, nsplit;
  long long blocks;
  size_t lds;
};


int g_wgrad_max_partial_mib = 12, g_wgrad_target_blocks = 512, g_wgrad_dbg = 0;

// splits of the specialised kernel: three workgroups per CU over the column blocks, under the cap on partial-sum bytes
int fast_nsplit(const WfArgs& a, int N, int C) {
  const long long bytes = (long long)N * a.taps * C * 4;
  long long s = (768 + a.ct * a.nb - 1) / (a.ct * a.nb);
  const long long cap = ((long long)g_wgrad_max_partial_mib << 20) / (bytes > 0 ? bytes : 1);
  if (s > cap) s = cap;
  if (s > a.ntiles) s = a.ntiles;
  if (s >= 16) s &= ~7ll;          // whole groups of eight splits (image-to-XCD numbering; with it off too: same partial sums)
  return s < 1 ? 1 : (int)s;
}


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
  if (max_partial_mib == -1000) {      // (lab / tests: the specialised 3 x 3 kernel: 0 off, 1 on, n > 1: on from n tiles)
    g_wgrad_fast = target_blocks != 0;
    g_wgrad_fast_min_tiles = target_blocks > 1 ? target_blocks : 1;
    return XPT_OK;
  }
  if (max_partial_mib < 0) g_wgrad_dbg = -max_partial_mib;          // lab knobs, see WgradArgs::dbg
  if (max_partial_mib > 0) g_wgrad_max_partial_mib = max_partial_mib;
  if (target_blocks > 0) g_wgrad_target_blocks = target_blocks;
  return XPT_OK;
}

extern "C" int xpt_conv2d_bwd_weight_splits(int B, int C, int N, int KH, int KW, int stride, int OH, int OW) {
  WgradPlan p;
  if (B <= 0 || C <= 0 || N <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || OH <= 0 || OW <= 0) return XPT_ERR_SHAPE;
  WfArgs f{};
  if (fast_plan(B, C, N, KH, KW, stride, OH, OW, 0, f)) return fast_nsplit(f, N, C);
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
  {
    WfArgs f{};
    if (fast_plan(B, C, N, KH, KW, stride, OH, OW, upsample, f) &&
        (upsample ? ((long long)PH * 2 == OH && (long long)PW * 2 == OW && pad_t == 1 && pad_l == 1)
                  : ((long long)(OH - 1) * stride - pad_t < PH && (long long)(OW - 1) * stride - pad_l < PW))) {
      f.nsplit = fast_nsplit(f, N, C);
      if (partial_floats < (size_t)f.nsplit * N * KH * KW * Cr) return XPT_ERR_WORKSPACE;
      f.g = (const unsigned short*)g; f.x = (const unsigned short*)x; f.part = partials;
      f.gpitch = gpitch; f.xpitch = xpitch;
      f.gbytes = ((long long)B * OH * OW - 1) * gpitch * 2 + (long long)N * 2;
      f.xbytes = ((long long)B * PH * PW - 1) * xpitch * 2 + (long long)C * 2;
      if (f.gbytes >= (1LL << 30) || f.xbytes >= (1LL << 30)) return XPT_ERR_SHAPE;
      f.B = B; f.PH = PH; f.PW = PW; f.OH = OH; f.OW = OW; f.C = C; f.Cr = Cr; f.N = N; f.pad_t = pad_t; f.pad_l = pad_l;
      f.xcd = (g_xpt_xcd_affinity != 0 && f.ntiles % 8 == 0 && f.nsplit % 8 == 0) ? 1 : 0;
      const dim3 grid(f.ct * f.nb * f.nsplit);
      hipStream_t st = (hipStream_t)stream;
      XPT_BEGIN_LAUNCH();
      if (upsample) hipLaunchKernelGGL((conv_wgrad_fast_kernel<3, 1, true>), grid, dim3(256), 0, st, f);
      else if (KH == 3 && stride == 1) hipLaunchKernelGGL((conv_wgrad_fast_kernel<3, 1, false>), grid, dim3(256), 0, st, f);
      else if (KH == 3) hipLaunchKernelGGL((conv_wgrad_fast_kernel<3, 2, false>), grid, dim3(256), 0, st, f);
      else hipLaunchKernelGGL((conv_wgrad_fast_kernel<5, 2, false>), grid, dim3(256), 0, st, f);
      return xpt_launch_status();
    }
  }
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
