// xpt_conv_splitk.hip -- the deep decoder convolutions (1,056 / 432 / 256 / 216 reduction channels on the 8 x 26 and
// 16 x 52 maps) as LDS-tiled implicit GEMMs with a DETERMINISTIC split of the reduction axis (round 4).
//
// Replaces, for those layers, the kernels of xpt_conv.hip behind keras Conv2D(padding="same") of DepthNetNoResize's decoder
// (model/build_model/depth_net.py:101-109: UpSampling2D(2, "nearest") -> conv 3x3 -> concat -> conv 3x3) and the
// tape.gradient of those layers w.r.t. their inputs (model/train_val.py:85-86).
//
// Why.  dp_up4_conv1 is a GEMM of 1,664 pixels x 256 channels x 9,504 reduction elements (8.1 GFLOP at batch 8): 104 tiles of
// 32 x 32 ... 64 x 64 cannot fill 256 CUs, every tile streams the whole reduction axis (500 MB through the L2s for the two
// deepest layers) and the launch ran at 148 TFLOP/s.  Here a workgroup owns TM = 128 pixels x TN = 128 (64) channels and ONE
// SLICE of the reduction axis (16-byte "pieces": 8 channels of one tap); both operands are staged through LDS in chunks of 8
// pieces with coalesced 16-byte loads (register double buffer), every wave runs 2 x 2 (2 x 1) v_mfma_f32_32x32x16_bf16
// tiles per 16 reduction elements, and the fp32 partial tile goes to a workspace [split][pixel][channel].  A second launch
// adds the splits IN ORDER, applies bias + LeakyReLU (and the 2 x 2 fold of a nearest-2x input in the data gradient) and
// writes bf16: no atomics, bit-repeatable.  Workgroups are numbered so that each XCD works on ONE slice of the reduction
// axis where the split count allows it: the slice's weights (0.6 MB of the 4.9 MB of dp_up4_conv1) stay in that XCD's L2.
#include "xpt_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef xpt_h16x8 bf16x8;      // (8 operands of the build's 16-bit format, xpt_common.h)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned short f2bf_sk(float f) { return xpt_f2h(f); }

struct SkArgs {
  const unsigned short* x;   // NHWC bf16 activations (forward: layer input; transposed: gradient at the layer output)
  const unsigned short* w;   // packed weights [N][T][C] bf16 (forward layout, or the transposed-mode layout [Cp][T][Np])
  float* part;               // [nsplit][M][N] fp32
  long long xpitch;
  long long xbytes, wbytes;  // extents of the two operands (buffer range checks)
  long long M;               // pixels the kernel enumerates (quad mode: parents x 4 children)
  int B, PH, PW;             // physical input extent
  int Hlim, Wlim, shift;     // logical extent the taps index (PH << shift)
  int C, N, KH, KW;
  int sgn, off_h, off_w;     // +1: t = o + k + off (forward, stride 1); -1 (transposed): t = o + off - k
  int OH, OW, quad;
  int npieces, per_split, nsplit;
  int tiles_m, tiles_n;
};

// TN = 64 RN output channels per workgroup.  NW = 4 waves: (2 pixel halves) x (2 channel halves), RN 32-channel tiles per
// wave; NW = 8 waves (TN = 128 only): (2 pixel halves) x (4 channel quarters), one tile per wave -- two waves per SIMD, so that
// one wave's LDS / barrier / staging latencies are another wave's issue slots.
template <int RN, int NW>
__global__ __launch_bounds__(64 * NW) void conv_splitk_kernel(SkArgs a) {
  constexpr int TN = 64 * RN, TP = 128, PITCH = 64 * 2 + 16;
  constexpr int RPP = 8 * NW, APASS = TN / RPP, BPASS = TP / RPP;   // staging: rows per pass, passes per operand
  constexpr int WI = NW == 4 ? RN : 1;                                            // 32-channel tiles per wave
  static_assert(NW == 4 || (NW == 8 && RN == 2), "8 waves: the 128-channel tile only");
  __shared__ __attribute__((aligned(16))) unsigned char lds[(TN + TP) * PITCH];
  unsigned char* const lA = lds;
  unsigned char* const lB = lds + TN * PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = NW == 4 ? wave >> 1 : wave >> 2, wn = NW == 4 ? wave & 1 : wave & 3;

  // ---- which (split, tile) this workgroup is: workgroups are dealt round-robin over the 8 XCDs
  const unsigned bid = blockIdx.x;
  int split, tile;
  if (a.nsplit >= 8) {                    // nsplit = 8 q: XCD x works on splits x, x + 8, ...
    split = (int)(bid % (unsigned)a.nsplit);
    tile = (int)(bid / (unsigned)a.nsplit);
  } else {                                // nsplit in {1, 2, 4}: XCD x works on split x % nsplit
    const int per = 8 / a.nsplit;
    split = (int)(bid & 7u) % a.nsplit;
    tile = (int)(bid >> 3) * per + (int)(bid & 7u) / a.nsplit;
  }
  if (tile >= a.tiles_m * a.tiles_n) return;                         // uniform per workgroup
  const int tn = tile % a.tiles_n, tm = tile / a.tiles_n;
  const long long m0 = (long long)tm * TP;
  const int n0 = tn * TN;
  const int p_begin = split * a.per_split;
  const int p_end = p_begin + a.per_split < a.npieces ? p_begin + a.per_split : a.npieces;
  const int nchunks = p_end > p_begin ? (p_end - p_begin + 7) >> 3 : 0;
  const int cp8 = a.C >> 3;
  const float inv_cp8 = 1.f / (float)cp8, inv_kw = 1.f / (float)a.KW;

  const unsigned M32 = (unsigned)a.M;
  auto decode = [&](long long m64, int& b, int& oh, int& ow) -> bool {
    const bool ok = m64 < a.M;
    const unsigned m = ok ? (unsigned)m64 : M32 - 1u;
    if (a.quad) {
      const unsigned child = m & 3u;
      unsigned c, rr;
      const unsigned q1 = xpt_divmod(m >> 2, (unsigned)a.OW >> 1, c);
      b = (int)xpt_divmod(q1, (unsigned)a.OH >> 1, rr);
      oh = (int)(2u * rr + (child >> 1));
      ow = (int)(2u * c + (child & 1u));
    } else {
      unsigned c, rr;
      const unsigned q1 = xpt_divmod(m, (unsigned)a.OW, c);
      b = (int)xpt_divmod(q1, (unsigned)a.OH, rr);
      oh = (int)rr;
      ow = (int)c;
    }
    return ok;
  };

  // ---- staging role: piece slot sp of the rows srow + 32 i.  The thread walks ITS pieces p = p_begin + sp, + 8, + 16, ...:
  // tap and channel offset advance incrementally (the tap changes every C / 64 chunks), the tap-dependent part of the
  // addresses -- pixel offsets, validity -- is recomputed only then.  (PMC of the first version: 4,150 vector instructions per
  // wave against 304 MFMAs -- one wave per SIMD issues an instruction every ~4.5 cycles, the launch was bound by the address
  // arithmetic of its own staging: 26 us whatever the slice count.)
  const int sp = tid & 7, srow = tid >> 3;
  int sb[BPASS], base_h[BPASS], base_w[BPASS];
  bool sok[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    int b, oh, ow;
    sok[i] = decode(m0 + srow + RPP * i, b, oh, ow);
    sb[i] = b;
    base_h[i] = oh + a.off_h;
    base_w[i] = ow + a.off_w;
  }
  const int T = a.KH * a.KW;
  // Both operands are read with BUFFER loads whose range check supplies the zeros: a weight row past N, a tap outside the
  // image and a piece past the slice get an offset beyond the buffer (markers of 1 GiB each: their sums cannot wrap, the
  // launcher refuses operands of 1 GiB or more) and come back as 0 -- no masks, no selects on the loaded vectors.
  constexpr unsigned OOB = 0x40000000u;
  unsigned wro[APASS];                        // byte offset of weight row n0 + srow + RPP i
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int n = n0 + srow + RPP * i;
    wro[i] = n < a.N ? (unsigned)n * (unsigned)(T * a.C) * 2u : OOB;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.wbytes, 0x00020000);

  f32x16 acc[WI][2];
#pragma unroll
  for (int i = 0; i < WI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // running piece of this thread
  int pc = p_begin + sp;                      // piece index
  int it, c0;                                 // its tap and channel offset
  {
    const int pcl = pc < a.npieces ? pc : a.npieces - 1;
    it = (int)(((float)pcl + 0.5f) * inv_cp8);
    c0 = (pcl - it * cp8) * 8;
  }
  unsigned xo[BPASS];                         // byte offset of pixel row i at the current tap (channel 0), OOB outside the image
  auto set_tap = [&]() {
    const int kh = (int)(((float)it + 0.5f) * inv_kw);
    const int kw = it - kh * a.KW;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const int th = base_h[i] + a.sgn * kh, tw = base_w[i] + a.sgn * kw;
      const bool ok = sok[i] && th >= 0 && th < a.Hlim && tw >= 0 && tw < a.Wlim;
      const int row = th >> a.shift, col = tw >> a.shift;
      xo[i] = ok ? (unsigned)((sb[i] * a.PH + row) * a.PW + col) * (unsigned)(a.xpitch * 2) : OOB;
    }
  };
  set_tap();

  // Two register sets: the loads of chunks ck + 1 and ck + 2 are in flight while chunk ck is multiplied.
  struct Stage {
    u32x4 ra[APASS], rb[BPASS];
  };
  auto fetch = [&](Stage& st) {               // the thread's next piece
    const unsigned dead = pc < p_end ? 0u : OOB;
    const unsigned cb = (unsigned)(it * a.C + c0) * 2u + dead, xb = (unsigned)c0 * 2u + dead;
#pragma unroll
    for (int i = 0; i < APASS; ++i) st.ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, wro[i] + cb, 0, 0);
#pragma unroll
    for (int i = 0; i < BPASS; ++i) st.rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, xo[i] + xb, 0, 0);
    pc += 8;
    c0 += 64;
    if (c0 >= a.C) {                          // next tap (per thread: the 8 piece slots of a chunk may straddle two taps)
      do {
        c0 -= a.C;
        ++it;
      } while (c0 >= a.C);                    // (fewer than 64 channels per tap: several taps per chunk)
      it = it < T ? it : T - 1;               // (past the last tap the piece is dead anyway)
      set_tap();
    }
  };
  auto stash = [&](const Stage& st) {
#pragma unroll
    for (int i = 0; i < APASS; ++i) *(u32x4*)(lA + (srow + RPP * i) * PITCH + sp * 16) = st.ra[i];
#pragma unroll
    for (int i = 0; i < BPASS; ++i) *(u32x4*)(lB + (srow + RPP * i) * PITCH + sp * 16) = st.rb[i];
  };
  const unsigned char* const fB = lB + (wm * 64 + r) * PITCH + h * 16;
  const unsigned char* const fA = lA + (wn * 32 * WI + r) * PITCH + h * 16;
  // all 16 reduction-element steps of a chunk unconditionally (pieces past the slice were staged as zeros): straight-line
  // code, every fragment read of the chunk in flight before the first MFMA waits for its operands
  auto multiply = [&]() {
    u32x4 fb[4][2], fa[4][WI];
#pragma unroll
    for (int k16 = 0; k16 < 4; ++k16) {
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[k16][j] = *(const u32x4*)(fB + 32 * j * PITCH + k16 * 32);
#pragma unroll
      for (int i = 0; i < WI; ++i) fa[k16][i] = *(const u32x4*)(fA + 32 * i * PITCH + k16 * 32);
    }
#pragma unroll
    for (int k16 = 0; k16 < 4; ++k16)
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = XPT_MFMA_32X32X16(__builtin_bit_cast(bf16x8, fa[k16][i]), __builtin_bit_cast(bf16x8, fb[k16][j]),
                                                              acc[i][j]);
  };

  Stage s0, s1;
  if (nchunks > 0) {
    fetch(s0);
    if (nchunks > 1) fetch(s1);
    stash(s0);
    if (nchunks > 2) fetch(s0);
    __syncthreads();
  }
  // invariant at the top of step ck (even): LDS holds chunk ck, s1 chunk ck + 1, s0 chunk ck + 2 (both in flight or landed)
  for (int ck = 0; ck < nchunks; ck += 2) {
    multiply();
    if (ck + 1 < nchunks) {
      __syncthreads();
      stash(s1);
      if (ck + 3 < nchunks) fetch(s1);
      __syncthreads();
      multiply();
      if (ck + 2 < nchunks) {
        __syncthreads();
        stash(s0);
        if (ck + 4 < nchunks) fetch(s0);
        __syncthreads();
      }
    }
  }

  // ---- partial tile: register q of tile (i, j) = channel n0 + 32 RN wn + 32 i + (q & 3) + 8 (q >> 2) + 4 h,
  //      pixel m0 + 64 wm + 32 j + r
  float* const out = a.part + (long long)split * a.M * a.N;
  const bool vec_ok = (a.N & 3) == 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long long m = m0 + 64 * wm + 32 * j + r;
    if (m >= a.M) continue;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
#pragma unroll
      for (int qg = 0; qg < 4; ++qg) {
        const int n = n0 + 32 * WI * wn + 32 * i + 8 * qg + 4 * h;
        if (n >= a.N) continue;
        float* dst = out + m * a.N + n;
        if (vec_ok) {
          *(float4*)dst = make_float4(acc[i][j][4 * qg], acc[i][j][4 * qg + 1], acc[i][j][4 * qg + 2], acc[i][j][4 * qg + 3]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < a.N) dst[e] = acc[i][j][4 * qg + e];
        }
      }
    }
  }
}

// y[pixel][n..n+3] = act(sum over splits (and the 4 children of a parent) of part + bias): one thread per (output pixel, 4 channels).
// NS = the slice count (compile time): all NS loads of a child are in flight before the first add (a run-time loop waited
// for every load in turn: 1.5 us per slice); the sum runs over the slices in order, then over the children in order.
template <int NS>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                                 unsigned short* __restrict__ y, long long ypitch, unsigned Mout,
                                                                 long long M, int N, int quad, float slope) {
  const unsigned n4 = (unsigned)(N + 3) >> 2;
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= Mout * n4) return;
  unsigned rem;
  const unsigned mo = xpt_divmod(idx, n4, rem);
  const int n = (int)rem * 4;
  const int nchild = quad ? 4 : 1;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  const bool vec = (N & 3) == 0;
  const long long slab = M * N;
  for (int c = 0; c < nchild; ++c) {
    const float* src = part + ((long long)mo * nchild + c) * N + n;
    if (vec) {
      float4 v[NS];
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) v[sp] = *(const float4*)(src + sp * slab);
#pragma unroll
      for (int sp = 0; sp < NS; ++sp) { s[0] += v[sp].x; s[1] += v[sp].y; s[2] += v[sp].z; s[3] += v[sp].w; }
    } else {
#pragma unroll
      for (int sp = 0; sp < NS; ++sp)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < N) s[e] += src[sp * slab + e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (bias != nullptr && n + e < N) s[e] += bias[n + e];
    s[e] = s[e] > 0.f ? s[e] : s[e] * slope;
  }
  unsigned short* dst = y + (long long)mo * ypitch + n;
  if (vec && (ypitch & 3) == 0 && (((uintptr_t)y) & 7) == 0) {
    uint2 pk;
    pk.x = (unsigned)f2bf_sk(s[0]) | ((unsigned)f2bf_sk(s[1]) << 16);
    pk.y = (unsigned)f2bf_sk(s[2]) | ((unsigned)f2bf_sk(s[3]) << 16);
    *(uint2*)dst = pk;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n + e < N) dst[e] = f2bf_sk(s[e]);
  }
}

int g_sk_skip_finish = 0;      // lab: launch the tile kernel only
int g_sk_waves = 8;             // waves per workgroup of the 128-channel tile (4 or 8)
int g_sk_enable = 1;            // 0: the split-K path answers "not served" for every layer
int g_sk_force_split = 0;       // > 0: this many slices whatever the shape (lab)
int g_sk_min_k = 1024;          // serve layers with at least this many reduction elements ...
int g_sk_max_pixels = 8192;     // ... and at most this many output pixels (the 8 x 26 / 16 x 52 maps at batch 8)

struct SkPlan {
  int nsplit, per_split, tiles_m, tiles_n, rn;
};

// the plan of a stride-1 layer, nsplit = 0 when the split-K path does not serve it
SkPlan sk_plan(long long M, int N, int C, int T) {
  SkPlan p{};
  if (!g_sk_enable || (C & 7) != 0 || M <= 0 || M >= 0x7fffffffLL) return p;
  const long long K = (long long)C * T;
  // measured at batch 8 (tools/bench_splitk.py, profiles/r04_b_bench_splitk.txt): faster than the kernels of xpt_conv.hip on
  // dp_up4_conv1 / conv2 both ways (55 -> 30, 41 -> 27, 27 -> 20, 28 -> 20 us) and on the forwards of dp_up3 (28 -> 23.5, 27 -> 23);
  // slower on dp_up3's data gradients (6,656 pixels x 1,152 reduction elements) and on everything from the 32 x 104 map up
  if (!g_sk_force_split && (K < g_sk_min_k || M > g_sk_max_pixels || (M > 2048 && K < 1536))) return p;
  p.rn = N > 64 ? 2 : 1;
  const int TN = 64 * p.rn;
  p.tiles_m = (int)((M + 127) / 128);
  p.tiles_n = (N + TN - 1) / TN;
  const int tiles = p.tiles_m * p.tiles_n;
  const int npieces = (int)(K / 8);
  int ns = 1;
  if (g_sk_force_split > 0) {
    ns = g_sk_force_split;
  } else {
    while (ns < 16 && tiles * ns * 2 <= 320) ns *= 2;             // ~ one workgroup per CU (256), power of two
    while (ns > 1 && npieces / ns < 32) ns >>= 1;                  // at least 4 chunks of 8 pieces per slice
  }
  if (ns != 1 && ns != 2 && ns != 4 && ns != 8 && ns != 16) return SkPlan{};
  p.per_split = ((npieces + ns - 1) / ns + 7) & ~7;               // whole chunks
  p.nsplit = ns;
  while (p.nsplit > 1 && (p.nsplit - 1) * p.per_split >= npieces) --p.nsplit;   // no empty slice
  if (p.nsplit != 1 && p.nsplit != 2 && p.nsplit != 4 && p.nsplit != 8 && p.nsplit != 16) {
    // (rounding to whole chunks emptied a slice: take the next power of two below)
    int q = 1;
    while (q * 2 <= p.nsplit) q *= 2;
    p.nsplit = q;
    p.per_split = ((npieces + q - 1) / q + 7) & ~7;
  }
  return p;
}

int launch_sk(SkArgs& a, const SkPlan& p, const float* bias, unsigned short* y, long long ypitch, float slope, hipStream_t s) {
  a.nsplit = p.nsplit; a.per_split = p.per_split; a.tiles_m = p.tiles_m; a.tiles_n = p.tiles_n;
  const long long tiles = (long long)p.tiles_m * p.tiles_n;
  long long blocks;
  if (p.nsplit >= 8) blocks = tiles * p.nsplit;
  else {
    const int per = 8 / p.nsplit;
    blocks = ((tiles + per - 1) / per) * 8;
  }
  if (blocks > 0x7fffffffLL) return XPT_ERR_SHAPE;
  XPT_BEGIN_LAUNCH();
  if (p.rn == 2 && g_sk_waves == 8) hipLaunchKernelGGL((conv_splitk_kernel<2, 8>), dim3((unsigned)blocks), dim3(512), 0, s, a);
  else if (p.rn == 2) hipLaunchKernelGGL((conv_splitk_kernel<2, 4>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_splitk_kernel<1, 4>), dim3((unsigned)blocks), dim3(256), 0, s, a);
  const long long Mout = a.quad ? a.M / 4 : a.M;
  const long long threads = Mout * ((a.N + 3) / 4);
  if (g_sk_skip_finish) return xpt_launch_status();
  const dim3 fgrid((unsigned)((threads + 255) / 256));
#define SK_FINISH(NS) hipLaunchKernelGGL(conv_splitk_finish_kernel<NS>, fgrid, dim3(256), 0, s, a.part, bias, y, ypitch, (unsigned)Mout, \
                                         a.M, a.N, a.quad, slope)
  switch (p.nsplit) {
    case 1: SK_FINISH(1); break;
    case 2: SK_FINISH(2); break;
    case 4: SK_FINISH(4); break;
    case 8: SK_FINISH(8); break;
    default: SK_FINISH(16); break;
  }
#undef SK_FINISH
  return xpt_launch_status();
}

}  // namespace

extern "C" int xpt_conv2d_splitk_tune(int enable, int force_split, int min_k, int max_pixels) {
  if (force_split < 0 || force_split > 16 || min_k < 0 || max_pixels < 0) return XPT_ERR_ARG;
  g_sk_enable = enable != 0;
  g_sk_skip_finish = enable == 2;      // (lab: 2 = tile kernel only, results are not finished)
  if (enable >= 4) { g_sk_waves = enable; g_sk_enable = 1; }      // (lab: 4 / 8 = waves per workgroup of the 128-channel tile)
  g_sk_force_split = force_split;
  if (min_k > 0) g_sk_min_k = min_k;
  if (max_pixels > 0) g_sk_max_pixels = max_pixels;
  return XPT_OK;
}

/* floats of workspace xpt_conv2d_{fwd,bwd_data}_splitk need for this layer; 0 = the layer is not served by the split-K
 * kernels (use xpt_conv2d_fwd / xpt_conv2d_bwd_data).  pixels = B x OH x OW of the grid the launch enumerates (data
 * gradient with fold2x2: B x 2 IH x 2 IW), out_channels / red_channels / taps of THAT launch (data gradient: out = C,
 * red = Np). */
extern "C" size_t xpt_conv2d_splitk_workspace_floats(long long pixels, int out_channels, int red_channels, int taps, int stride) {
  if (stride != 1) return 0;
  const SkPlan p = sk_plan(pixels, out_channels, red_channels, taps);
  return p.nsplit ? (size_t)p.nsplit * (size_t)pixels * (size_t)out_channels : 0;
}

extern "C" int xpt_conv2d_fwd_splitk(const void* x, const void* w, const float* bias, void* y, int B, int PH, int PW, int C,
                                     long long xpitch, int N, int KH, int KW, int pad_t, int pad_l, int OH, int OW,
                                     long long ypitch, int upsample, float slope, float* workspace, size_t workspace_floats,
                                     void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(w); XPT_CHECK_PTR(y); XPT_CHECK_PTR(workspace);
  if (B <= 0 || PH <= 0 || PW <= 0 || C <= 0 || N <= 0 || KH <= 0 || KW <= 0 || OH <= 0 || OW <= 0) return XPT_ERR_SHAPE;
  if (C % 8 != 0 || xpitch < C || xpitch % 8 != 0 || ypitch < N || ((uintptr_t)x) % 16 != 0 || ((uintptr_t)w) % 16 != 0 ||
      ((uintptr_t)workspace) % 16 != 0)
    return XPT_ERR_ARG;
  if ((upsample != 0 && upsample != 1) || pad_t < 0 || pad_l < 0) return XPT_ERR_ARG;
  SkArgs a{};
  a.x = (const unsigned short*)x; a.w = (const unsigned short*)w; a.part = workspace;
  a.xpitch = xpitch;
  a.B = B; a.PH = PH; a.PW = PW; a.shift = upsample; a.Hlim = PH << upsample; a.Wlim = PW << upsample;
  a.C = C; a.N = N; a.KH = KH; a.KW = KW;
  a.sgn = 1; a.off_h = -pad_t; a.off_w = -pad_l;
  a.OH = OH; a.OW = OW; a.quad = 0;
  a.M = (long long)B * OH * OW;
  a.npieces = KH * KW * (C / 8);
  if ((long long)(OH - 1) - pad_t >= a.Hlim || (long long)(OW - 1) - pad_l >= a.Wlim) return XPT_ERR_SHAPE;
  // the staging loop addresses both operands with 32-bit byte offsets
  a.xbytes = ((long long)B * PH * PW - 1) * xpitch * 2 + (long long)C * 2;
  a.wbytes = (long long)N * KH * KW * C * 2;
  if (a.xbytes >= (1LL << 30) || a.wbytes >= (1LL << 30)) return XPT_ERR_SHAPE;
  const SkPlan p = sk_plan(a.M, N, C, KH * KW);
  if (!p.nsplit) return XPT_ERR_ARG;
  if (workspace_floats < (size_t)p.nsplit * (size_t)a.M * (size_t)N) return XPT_ERR_WORKSPACE;
  return launch_sk(a, p, bias, (unsigned short*)y, ypitch, slope, (hipStream_t)stream);
}

extern "C" int xpt_conv2d_bwd_data_splitk(const void* g, const void* wb, void* dx, int B, int OH, int OW, int Np, long long gpitch,
                                          int C, int KH, int KW, int pad_t, int pad_l, int IH, int IW, long long dxpitch,
                                          int fold2x2, float* workspace, size_t workspace_floats, void* stream) {
  XPT_CHECK_PTR(g); XPT_CHECK_PTR(wb); XPT_CHECK_PTR(dx); XPT_CHECK_PTR(workspace);
  if (B <= 0 || OH <= 0 || OW <= 0 || Np <= 0 || C <= 0 || KH <= 0 || KW <= 0 || IH <= 0 || IW <= 0) return XPT_ERR_SHAPE;
  if (Np % 8 != 0 || gpitch < Np || gpitch % 8 != 0 || dxpitch < C || ((uintptr_t)g) % 16 != 0 || ((uintptr_t)wb) % 16 != 0 ||
      ((uintptr_t)workspace) % 16 != 0)
    return XPT_ERR_ARG;
  if (fold2x2 != 0 && fold2x2 != 1) return XPT_ERR_ARG;
  SkArgs a{};
  a.x = (const unsigned short*)g; a.w = (const unsigned short*)wb; a.part = workspace;
  a.xpitch = gpitch;
  a.B = B; a.PH = OH; a.PW = OW; a.shift = 0; a.Hlim = OH; a.Wlim = OW;
  a.C = Np; a.N = C; a.KH = KH; a.KW = KW;
  a.sgn = -1; a.off_h = pad_t; a.off_w = pad_l;
  a.OH = fold2x2 ? 2 * IH : IH; a.OW = fold2x2 ? 2 * IW : IW; a.quad = fold2x2;
  if (fold2x2 && (a.OH != OH || a.OW != OW)) return XPT_ERR_SHAPE;      // stride 1: the gradient grid IS the up-sampled input grid
  a.M = (long long)B * a.OH * a.OW;
  a.npieces = KH * KW * (Np / 8);
  a.xbytes = ((long long)B * OH * OW - 1) * gpitch * 2 + (long long)Np * 2;
  a.wbytes = (long long)C * KH * KW * Np * 2;
  if (a.xbytes >= (1LL << 30) || a.wbytes >= (1LL << 30)) return XPT_ERR_SHAPE;
  const SkPlan p = sk_plan(a.M, C, Np, KH * KW);
  if (!p.nsplit) return XPT_ERR_ARG;
  if (workspace_floats < (size_t)p.nsplit * (size_t)a.M * (size_t)C) return XPT_ERR_WORKSPACE;
  return launch_sk(a, p, nullptr, (unsigned short*)dx, dxpitch, 1.f, (hipStream_t)stream);
}
