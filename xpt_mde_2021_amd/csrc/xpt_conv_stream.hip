// xpt_conv_stream.hip -- the 3 x 3 stride-1 convolutions of the decoder's half- and full-resolution levels (dp_up1, dp_up0:
// 16 ... 80 channels on the 64 x 208 and 128 x 416 maps) as PERSISTENT, weight-stationary workgroups (round 4).
//
// Replaces, for those layers, conv_halo_kernel (xpt_conv.hip) behind keras Conv2D(padding="same") + bias + LeakyReLU of
// DepthNetNoResize's decoder (model/build_model/depth_net.py:101-109: UpSampling2D(2, "nearest") -> conv 3x3 -> concat ->
// conv 3x3) and the tape.gradient of those layers w.r.t. their inputs (model/train_val.py:85-86).
//
// Why.  Those launches move 20 - 34 MB and ran at 8 - 17 % of the HBM roofline (24 us for 20 MB).  conv_halo_kernel gives every
// 8 x 16-pixel tile its own workgroup, and each of the 3,328 workgroups of a full-resolution launch stages the layer's WHOLE
// weight slab (9 KB for 3 KB of input halo), computes the geometry of every staged vector from scratch (~40 vector
// instructions per 16 bytes: float reciprocals, 64-bit offsets, selects and masks) and then waits for one memory round trip
// before its 18 MFMAs.  PMC on the sibling kernels says it plainly: one wave per SIMD issues an instruction every ~4.5 cycles,
// these launches are bound by the instruction stream of their own staging.  Here
//   * a workgroup stages the weights ONCE and then walks over tiles (grid = 2 workgroups per CU, tiles dealt image-major per
//     XCD: a tile's halo rows were written by the producer of the same image on the same XCD);
//   * the geometry of a thread's staged vectors (halo pixel, channel group, LDS address) is computed once per workgroup;
//     per tile it is two adds, a range check and a select per vector;
//   * both operands come through BUFFER loads whose range check supplies the zeros of the padding, of the channel pad and of
//     the ragged edges -- no masks on loaded data;
//   * the next tile's halo is in flight (registers) while the current one is multiplied and stored.
// Modes and epilogue are conv_halo_kernel's (forward, transposed = data gradient, nearest-2x input, 2 x 2 fold).
#include "xpt_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef xpt_h16x8 bf16x8;      // (8 operands of the build's 16-bit format, xpt_common.h)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned short f2bf_st(float f) { return xpt_f2h(f); }

constexpr int ST_MAXV = 8;                 // 16-byte halo vectors a thread stages per tile
constexpr unsigned ST_OOB = 0x40000000u;   // offset marker beyond any operand (operands are < 1 GiB: the launcher checks)

struct StArgs {
  const unsigned short* x;   // NHWC bf16 activations (forward: layer input; transposed: gradient at the layer output)
  const unsigned short* w;   // packed weights [N][9][C] bf16 (forward layout, or the transposed-mode layout [Cp][9][Np])
  const float* bias;         // [N] or null
  unsigned short* y;
  long long xpitch, ypitch, xbytes, wbytes, ybytes;
  int B, PH, PW, Hlim, Wlim, shift;
  int C, Cc, N;              // reduction channels (multiple of 8), the same rounded up to 16, output channels
  int sgn, off_h, off_w;
  int OH, OW, quad;
  float slope;
  int tiles_x, tiles_y, ntiles;
  int HR, WR, XP, WP;        // physical halo extent; LDS pitches in bytes per halo pixel / per weight row
  int xcd;                   // 1: tiles dealt image-major per XCD
  int k5s2;                  // 1: 5 x 5 filter, stride 2, forward (specialised instantiation only)
};

template <int RM>
__global__ __launch_bounds__(256) void conv_stream_kernel(StArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sl[];
  constexpr int TN = 32 * RM, T = 9;
  unsigned char* const lW = sl;
  unsigned char* const lX = sl + TN * a.WP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int cvp = a.Cc >> 3;               // 16-byte vectors per pixel / per (row, tap) in LDS (channel pad included)
  const int HPIX = a.HR * a.WR;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.wbytes, 0x00020000);

  // ---- the weight slab, once: [TN rows][9 taps][Cc] with rows past N and channels past C as zeros (range check)
  {
    const unsigned per_row = (unsigned)(T * cvp), nvec = (unsigned)TN * per_row;
    for (unsigned v = tid; v < nvec; v += 256) {
      unsigned rest, cv;
      const unsigned n = xpt_divmod(v, per_row, rest);
      const unsigned tap = xpt_divmod(rest, (unsigned)cvp, cv);
      const bool ok = (int)n < a.N && (int)(cv * 8) < a.C;
      const unsigned off = ok ? ((n * T + tap) * (unsigned)a.C + cv * 8) * 2u : ST_OOB;
      const u32x4 val = __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0);
      *(u32x4*)(lW + n * a.WP + (tap * a.Cc + cv * 8) * 2) = val;
    }
  }

  // ---- geometry of this thread's halo vectors, once: vector v = tid + 256 i = (halo pixel v / cvp, channel group v % cvp)
  int vq[ST_MAXV], vrem[ST_MAXV], vlds[ST_MAXV];
  unsigned vcb[ST_MAXV];                   // channel byte offset, ST_OOB for pad channels / no vector
#pragma unroll
  for (int i = 0; i < ST_MAXV; ++i) {
    const unsigned v = (unsigned)tid + 256u * i;
    unsigned cv, rem;
    const unsigned pix = xpt_divmod(v, (unsigned)cvp, cv);
    const unsigned q = xpt_divmod(pix, (unsigned)a.WR, rem);
    const bool live = (int)pix < HPIX;
    vq[i] = (int)q;
    vrem[i] = (int)rem;
    vlds[i] = live ? (int)(pix * a.XP + cv * 16) : -1;
    vcb[i] = (live && (int)(cv * 8) < a.C) ? cv * 16u : ST_OOB;
  }

  // this lane's output pixel inside a tile
  int ty, tx;
  if (a.quad) {
    const int child = r & 3;
    ty = 2 * wave + (child >> 1);
    tx = 2 * (r >> 2) + (child & 1);
  } else {
    ty = 2 * wave + (r >> 4);
    tx = r & 15;
  }

  auto tile_of = [&](int t, int& b, int& oh0, int& ow0) {
    unsigned u = (unsigned)t;
    if (a.xcd) u = (u & 7u) * ((unsigned)a.ntiles >> 3) + (u >> 3);      // XCD x owns a contiguous (image-major) range of tiles
    unsigned txi, tyi;
    const unsigned q1 = xpt_divmod(u, (unsigned)a.tiles_x, txi);
    b = (int)xpt_divmod(q1, (unsigned)a.tiles_y, tyi);
    oh0 = (int)tyi * 8;
    ow0 = (int)txi * 16;
  };

  u32x4 stage[ST_MAXV];
  auto fetch = [&](int t) {
    int b, oh0, ow0;
    tile_of(t, b, oh0, ow0);
    const int lo_h = oh0 + a.off_h - (a.sgn > 0 ? 0 : 2), lo_w = ow0 + a.off_w - (a.sgn > 0 ? 0 : 2);
    const int plo_h = lo_h >> a.shift, plo_w = lo_w >> a.shift;          // arithmetic shifts: floor for negative coordinates
    const unsigned pitch2 = (unsigned)(a.xpitch * 2);
#pragma unroll
    for (int i = 0; i < ST_MAXV; ++i) {
      const int pr = plo_h + vq[i], pc = plo_w + vrem[i];
      const bool ok = pr >= 0 && pr < a.PH && pc >= 0 && pc < a.PW;
      const unsigned off = ok ? (unsigned)((b * a.PH + pr) * a.PW + pc) * pitch2 + vcb[i] : ST_OOB;
      stage[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < ST_MAXV; ++i)
      if (vlds[i] >= 0) *(u32x4*)(lX + vlds[i]) = stage[i];
  };

  const int k16n = a.Cc >> 4;
  const bool vec_ok = (a.N % 4 == 0) && (a.ypitch % 4 == 0) && (((uintptr_t)a.y) % 8 == 0);
  int t = (int)blockIdx.x;
  if (t < a.ntiles) fetch(t);
  for (; t < a.ntiles; t += (int)gridDim.x) {
    __syncthreads();                                   // the previous tile's operand reads are done (and the weight slab is written)
    stash();
    __syncthreads();
    int b, oh0, ow0;
    tile_of(t, b, oh0, ow0);
    if (t + (int)gridDim.x < a.ntiles) fetch(t + (int)gridDim.x);        // in flight behind the products and the stores below

    const int lo_h = oh0 + a.off_h - (a.sgn > 0 ? 0 : 2), lo_w = ow0 + a.off_w - (a.sgn > 0 ? 0 : 2);
    const int plo_h = lo_h >> a.shift, plo_w = lo_w >> a.shift;
    const int oh = oh0 + ty, ow = ow0 + tx;
    const bool pok = oh < a.OH && ow < a.OW;
    const int base_h = (pok ? oh : oh0) + a.off_h, base_w = (pok ? ow : ow0) + a.off_w;

    f32x16 acc[RM];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    const unsigned char* const wA = lW + r * a.WP + h * 16;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int th = base_h + a.sgn * kh, tw = base_w + a.sgn * kw;
        const int prow = (th >> a.shift) - plo_h, pcol = (tw >> a.shift) - plo_w;
        const unsigned char* const xB = lX + (prow * a.WR + pcol) * a.XP + h * 16;
        const unsigned char* const wT = wA + (kh * 3 + kw) * a.Cc * 2;
        for (int k16 = 0; k16 < k16n; ++k16) {
          const u32x4 fb = *(const u32x4*)(xB + k16 * 32);
#pragma unroll
          for (int i = 0; i < RM; ++i) {
            const u32x4 fa = *(const u32x4*)(wT + 32 * i * a.WP + k16 * 32);
            acc[i] = XPT_MFMA_32X32X16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc[i]);
          }
        }
      }
    }

    // ---- epilogue (conv_halo_kernel's): register q of tile i = channel 32 i + (q & 3) + 8 (q >> 2) + 4 h
    const long long opix = a.quad ? ((long long)b * (a.OH >> 1) + (oh >> 1)) * (a.OW >> 1) + (ow >> 1)
                                  : ((long long)b * a.OH + oh) * a.OW + ow;
#pragma unroll
    for (int i = 0; i < RM; ++i) {
#pragma unroll
      for (int qg = 0; qg < 4; ++qg) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * qg + e];
        if (a.quad) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] += __shfl_xor(v[e], 1, 64);
            v[e] += __shfl_xor(v[e], 2, 64);
          }
        }
        const int n = 32 * i + 8 * qg + 4 * h;
        if (!pok || (a.quad && (r & 3) != 0) || n >= a.N) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (a.bias != nullptr && n + e < a.N) v[e] += a.bias[n + e];
          v[e] = v[e] > 0.f ? v[e] : v[e] * a.slope;
        }
        unsigned short* dst = a.y + opix * a.ypitch + n;
        if (vec_ok) {
          uint2 pk;
          pk.x = (unsigned)f2bf_st(v[0]) | ((unsigned)f2bf_st(v[1]) << 16);
          pk.y = (unsigned)f2bf_st(v[2]) | ((unsigned)f2bf_st(v[3]) << 16);
          *(uint2*)dst = pk;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < a.N) dst[e] = f2bf_st(v[e]);
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------ specialised instantiations
// PMC of the kernel above on dp_up0_conv1 (profiles/r04_c_pmc_stream.md): 612 vector instructions per tile and wave around 18
// MFMAs -- like conv_halo_kernel (877) the launch is bound by its own instruction stream (8 - 11 M wave-instructions over
// 1,024 SIMDs), not by bytes or by the matrix pipe.  For the four layer shapes of the half- / full-resolution levels the
// kernel is therefore instantiated with everything the instruction stream does not need to compute at run time:
//   MODE (0 forward, 1 forward with a nearest-2x input, 2 data gradient, 3 data gradient with the 2 x 2 fold), CC = reduction
//   channels / 16 and RM = output channels / 32 are template parameters: LDS pitches, halo extent, vectors per thread and
//   every loop bound are constants;
//   the LDS address of a lane's operand row does not depend on the tile (tiles start at even rows / columns): nine
//   lane-constant addresses, the channel steps and the weight rows are instruction immediates -- no address arithmetic
//   between the MFMAs; small weight slabs (<= 18 fragments) live in registers;
//   for a tile whose halo lies inside the image a staged vector's global offset is lane-constant + a scalar tile offset
//   (buffer soffset); only border tiles compute range checks;
//   the output address is lane-constant + a scalar tile offset as well (buffer stores), the bias sits in registers,
//   LeakyReLU is max(v, slope v).
template <int RM, int CC, int MODE>
__global__ __launch_bounds__(256) void conv_stream_fast_kernel(StArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sl[];
  // MODE 4: forward of a 5 x 5 stride-2 layer (PoseNetImproved's first two convolutions, pose_net.py:60-61)
  constexpr bool UPS = MODE == 1, QUAD = MODE == 3;
  constexpr int KK = MODE == 4 ? 5 : 3, SS = MODE == 4 ? 2 : 1, TT = KK * KK;
  constexpr int SGN = (MODE == 2 || MODE == 3) ? -1 : 1, BACK = (MODE == 2 || MODE == 3) ? 2 : 0;
  constexpr int TN = 32 * RM, Cc = 16 * CC, cvp = 2 * CC, XP = Cc * 2 + 16, WP = TT * Cc * 2 + 16;
  constexpr int HR = UPS ? 6 : 7 * SS + KK, WR = UPS ? 10 : 15 * SS + KK, HPIX = HR * WR, NV = (HPIX * cvp + 255) / 256;
  constexpr bool AREG = RM * CC * TT <= 18;          // the lane's weight fragments stay in registers
  unsigned char* const lW = sl;
  unsigned char* const lX = sl + TN * WP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (int)a.wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (int)a.ybytes, 0x00020000);

  // ---- the weight slab, once
  {
    constexpr unsigned per_row = (unsigned)TT * cvp, nvec = (unsigned)TN * per_row;
    for (unsigned v = tid; v < nvec; v += 256) {
      const unsigned n = v / per_row, rest = v - n * per_row;      // (compile-time divisors)
      const unsigned tap = rest / cvp, cv = rest - tap * cvp;
      const bool ok = (int)n < a.N && (int)(cv * 8) < a.C;
      const unsigned off = ok ? ((n * (unsigned)TT + tap) * (unsigned)a.C + cv * 8) * 2u : ST_OOB;
      *(u32x4*)(lW + n * WP + (tap * Cc + cv * 8) * 2) = __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0);
    }
  }

  // ---- this thread's halo vectors: lane constants
  int vq[NV], vrem[NV], vlds[NV];
  unsigned vcb[NV], vconst[NV];            // channel byte offset (ST_OOB: none); offset of the vector relative to the halo's first pixel
  const unsigned pitch2 = (unsigned)(a.xpitch * 2);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const unsigned v = (unsigned)tid + 256u * i;
    const unsigned pix = v / cvp, cv = v - pix * cvp;
    const unsigned q = pix / WR, rem = pix - q * WR;
    const bool live = (int)pix < HPIX;
    vq[i] = (int)q;
    vrem[i] = (int)rem;
    vlds[i] = live ? (int)(pix * XP + cv * 16) : -1;
    vcb[i] = (live && (int)(cv * 8) < a.C) ? cv * 16u : ST_OOB;
    vconst[i] = (q * (unsigned)a.PW + rem) * pitch2 + vcb[i];      // (>= ST_OOB for dead vectors: operands are < 1 GiB)
  }

  // ---- this lane's output pixel inside a tile, the LDS addresses of its operand rows, its output offset
  int ty, tx;
  if (QUAD) {
    const int child = r & 3;
    ty = 2 * wave + (child >> 1);
    tx = 2 * (r >> 2) + (child & 1);
  } else {
    ty = 2 * wave + (r >> 4);
    tx = r & 15;
  }
  // nearest-2x input: nine lane constants (the source row of an output row depends on its parity); otherwise ONE lane constant,
  // the taps are instruction immediates
  unsigned bofs[UPS ? 3 : 1][UPS ? 3 : 1];
  if constexpr (UPS) {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        // tiles start at even rows / columns: ((oh0 + t) >> 1) - (oh0 + off >> 1) does not depend on oh0
        const int prow = ((ty + a.off_h + kh) >> 1) - (a.off_h >> 1), pcol = ((tx + a.off_w + kw) >> 1) - (a.off_w >> 1);
        bofs[kh][kw] = (unsigned)(TN * WP + (prow * WR + pcol) * XP + h * 16);
      }
  } else {
    bofs[0][0] = (unsigned)(TN * WP + ((ty * SS + BACK) * WR + tx * SS + BACK) * XP + h * 16);
  }
  const unsigned wofs = (unsigned)(r * WP + h * 16);
  const unsigned yp2 = (unsigned)(a.ypitch * 2);
  const unsigned yconst = QUAD ? ((unsigned)((ty >> 1) * (a.OW >> 1) + (tx >> 1)) * yp2 + 8u * h)
                               : ((unsigned)(ty * a.OW + tx) * yp2 + 8u * h);
  float bias[RM][4][4];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int qg = 0; qg < 4; ++qg)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = 32 * i + 8 * qg + 4 * h + e;
        bias[i][qg][e] = (a.bias != nullptr && n < a.N) ? a.bias[n] : 0.f;
      }

  auto tile_of = [&](int t, int& b, int& oh0, int& ow0) {
    unsigned u = (unsigned)t;
    if (a.xcd) u = (u & 7u) * ((unsigned)a.ntiles >> 3) + (u >> 3);
    unsigned txi, tyi;
    const unsigned q1 = xpt_divmod(u, (unsigned)a.tiles_x, txi);
    b = (int)xpt_divmod(q1, (unsigned)a.tiles_y, tyi);
    oh0 = (int)tyi * 8;
    ow0 = (int)txi * 16;
  };

  u32x4 stage[NV];
  auto fetch = [&](int t) {
    int b, oh0, ow0;
    tile_of(t, b, oh0, ow0);
    b = __builtin_amdgcn_readfirstlane(b); oh0 = __builtin_amdgcn_readfirstlane(oh0); ow0 = __builtin_amdgcn_readfirstlane(ow0);
    const int plo_h = (oh0 * SS + a.off_h - BACK) >> (UPS ? 1 : 0), plo_w = (ow0 * SS + a.off_w - BACK) >> (UPS ? 1 : 0);
    const bool inside = plo_h >= 0 && plo_h + HR <= a.PH && plo_w >= 0 && plo_w + WR <= a.PW;      // (uniform)
    if (inside) {
      const unsigned soff = (unsigned)((b * a.PH + plo_h) * a.PW + plo_w) * pitch2;
#pragma unroll
      for (int i = 0; i < NV; ++i) stage[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, vconst[i], soff, 0);
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int pr = plo_h + vq[i], pc = plo_w + vrem[i];
        const bool ok = (unsigned)pr < (unsigned)a.PH && (unsigned)pc < (unsigned)a.PW;
        const unsigned off = ok ? (unsigned)((b * a.PH + pr) * a.PW + pc) * pitch2 + vcb[i] : ST_OOB;
        stage[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
      }
    }
  };

  u32x4 fareg[AREG ? TT : 1][AREG ? CC : 1][AREG ? RM : 1];
  bool have_a = false;
  int t = (int)blockIdx.x;
  if (t < a.ntiles) fetch(t);
  for (; t < a.ntiles; t += (int)gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (vlds[i] >= 0) *(u32x4*)(lX + vlds[i]) = stage[i];
    __syncthreads();
    if (AREG && !have_a) {                               // (the slab is in LDS behind the first barrier pair)
#pragma unroll
      for (int tap = 0; tap < TT; ++tap)
#pragma unroll
        for (int k = 0; k < CC; ++k)
#pragma unroll
          for (int i = 0; i < RM; ++i) fareg[tap][k][i] = *(const u32x4*)(sl + wofs + 32 * i * WP + (tap * Cc + 16 * k) * 2);
      have_a = true;
    }
    int b, oh0, ow0;
    tile_of(t, b, oh0, ow0);
    b = __builtin_amdgcn_readfirstlane(b); oh0 = __builtin_amdgcn_readfirstlane(oh0); ow0 = __builtin_amdgcn_readfirstlane(ow0);
    if (t + (int)gridDim.x < a.ntiles) fetch(t + (int)gridDim.x);

    f32x16 acc[RM];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
#pragma unroll
    for (int kh = 0; kh < KK; ++kh)
#pragma unroll
      for (int kw = 0; kw < KK; ++kw)
#pragma unroll
        for (int k = 0; k < CC; ++k) {
          const u32x4 fb = UPS ? *(const u32x4*)(sl + bofs[UPS ? kh : 0][UPS ? kw : 0] + 32 * k)
                               : *(const u32x4*)(sl + bofs[0][0] + (SGN * (kh * WR + kw)) * XP + 32 * k);
#pragma unroll
          for (int i = 0; i < RM; ++i) {
            const u32x4 fa = AREG ? fareg[AREG ? kh * KK + kw : 0][AREG ? k : 0][AREG ? i : 0]
                                  : *(const u32x4*)(sl + wofs + 32 * i * WP + ((kh * KK + kw) * Cc + 16 * k) * 2);
            acc[i] = XPT_MFMA_32X32X16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), acc[i]);
          }
        }

    // ---- epilogue: register q of tile i = channel 32 i + (q & 3) + 8 (q >> 2) + 4 h of this lane's pixel
    const bool pok = oh0 + ty < a.OH && ow0 + tx < a.OW;
    const unsigned ysoff = QUAD ? (unsigned)((b * (a.OH >> 1) + (oh0 >> 1)) * (a.OW >> 1) + (ow0 >> 1)) * yp2
                                : (unsigned)((b * a.OH + oh0) * a.OW + ow0) * yp2;
#pragma unroll
    for (int i = 0; i < RM; ++i) {
#pragma unroll
      for (int qg = 0; qg < 4; ++qg) {
        if (32 * i + 8 * qg >= a.N) continue;                       // (uniform: N is a multiple of 8)
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * qg + e];
        if (QUAD) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] += __shfl_xor(v[e], 1, 64);
            v[e] += __shfl_xor(v[e], 2, 64);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] += bias[i][qg][e];
          v[e] = __builtin_fmaxf(v[e], v[e] * a.slope);             // LeakyReLU for 0 <= slope <= 1 (1: linear)
        }
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        u32x2 pk;
        pk.x = (unsigned)f2bf_st(v[0]) | ((unsigned)f2bf_st(v[1]) << 16);
        pk.y = (unsigned)f2bf_st(v[2]) | ((unsigned)f2bf_st(v[3]) << 16);
        const bool mine = pok && (!QUAD || (r & 3) == 0);
        __builtin_amdgcn_raw_buffer_store_b64(pk, ry, mine ? yconst + (unsigned)(64 * i + 16 * qg) : ST_OOB, ysoff, 0);
      }
    }
  }
}

int g_st_enable = 1;
int g_st_min_tiles = 512;        // serve layers with at least this many 8 x 16 tiles ...
int g_st_max_lds = 80 * 1024;    // ... whose weight slab + halo fit (two workgroups per CU)
int g_st_wgs_per_cu = 3;
int g_st_fast = 1;                // 0: the generic kernel for every shape (lab / tests)

// fills the plan fields of `a`; false: the layer is outside what this kernel serves
bool st_plan(StArgs& a, int& rm, size_t& lds) {
  if (!g_st_enable || (a.C & 7) != 0 || a.N > 96) return false;
  if (a.quad && ((a.OH | a.OW) & 1)) return false;
  rm = (a.N + 31) / 32;
  a.Cc = (a.C + 15) / 16 * 16;
  a.tiles_x = (a.OW + 15) / 16;
  a.tiles_y = (a.OH + 7) / 8;
  const long long nt = (long long)a.B * a.tiles_y * a.tiles_x;
  // (the 5 x 5 stride-2 layers are served from 128 tiles on: no second tile to amortise the slab over, but 25 taps as immediates)
  if (nt < (a.k5s2 && g_st_min_tiles > 128 ? 128 : g_st_min_tiles) || nt > 0x3fffffffLL) return false;
  a.ntiles = (int)nt;
  const int rows = a.k5s2 ? 7 * 2 + 5 : 8 + 2, cols = a.k5s2 ? 15 * 2 + 5 : 16 + 2;
  a.HR = a.shift ? rows / 2 + 1 : rows;
  a.WR = a.shift ? cols / 2 + 1 : cols;
  a.XP = a.Cc * 2 + 16;
  a.WP = (a.k5s2 ? 25 : 9) * a.Cc * 2 + 16;
  if (a.k5s2 && (rm != 1 || a.Cc > 32 || a.shift || a.quad || a.sgn < 0 || !g_st_fast)) return false;      // (two instantiations exist)
  if (!a.k5s2 && a.HR * a.WR * (a.Cc / 8) > 256 * ST_MAXV) return false;
  lds = (size_t)32 * rm * a.WP + (size_t)a.HR * a.WR * a.XP;
  if (lds > (a.k5s2 && g_st_max_lds < 128 * 1024 ? (size_t)128 * 1024 : (size_t)g_st_max_lds)) return false;
  if (a.xbytes >= (1LL << 30) || a.wbytes >= (1LL << 30)) return false;
  return true;
}

int launch_st(StArgs& a, int rm, size_t lds, hipStream_t s) {
  int cus = 256;
  {
    static int cached = 0;
    if (!cached) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cached = prop.multiProcessorCount;
      else
        cached = 256;
    }
    cus = cached;
  }
  // workgroups per CU: as many as asked for and as the LDS holds (a grid larger than what is resident leaves the
  // persistent loop with a ragged second round: dp_up1_conv2 at 3 x 78 KB ran 16.2 us instead of 12.8)
  int per_cu = (int)((size_t)160 * 1024 / (lds ? lds : 1));
  if (per_cu > g_st_wgs_per_cu) per_cu = g_st_wgs_per_cu;
  if (per_cu < 1) per_cu = 1;
  int grid = cus * per_cu;
  grid -= grid % 8;
  if (grid > a.ntiles) grid = a.ntiles;
  a.xcd = (g_xpt_xcd_affinity != 0 && a.ntiles % 8 == 0 && grid % 8 == 0) ? 1 : 0;
  XPT_BEGIN_LAUNCH();
  // the specialised instantiations: the decoder's half- / full-resolution layers (forward and data gradient)
  const int mode = a.k5s2 ? 4 : (a.sgn > 0 ? (a.shift ? 1 : 0) : (a.quad ? 3 : 2)), cc = a.Cc / 16;
  const bool fast_ok = g_st_fast && a.N % 8 == 0 && a.ypitch % 4 == 0 && ((uintptr_t)a.y) % 8 == 0 && a.ybytes < (1LL << 30) &&
                       a.slope >= 0.f && a.slope <= 1.f && (!a.shift || (a.off_h == -1 && a.off_w == -1));
#define ST_FAST(RM_, CC_, MODE_)                                                                                        \
  if (fast_ok && rm == RM_ && cc == CC_ && mode == MODE_) {                                                               \
    hipLaunchKernelGGL((conv_stream_fast_kernel<RM_, CC_, MODE_>), dim3(grid), dim3(256), lds, s, a);                      \
    return xpt_launch_status();                                                                                           \
  }
  if (a.k5s2) {                                                           // PoseNetImproved conv1 (15 -> 32) / conv2 (32 -> 32)
    if (!fast_ok) return XPT_ERR_ARG;
    ST_FAST(1, 1, 4) ST_FAST(1, 2, 4)
    return XPT_ERR_ARG;
  }
  ST_FAST(1, 4, 1) ST_FAST(1, 5, 0) ST_FAST(1, 2, 1) ST_FAST(1, 2, 0)      // dp_up1_conv1 / conv2, dp_up0_conv1 / conv2
  ST_FAST(2, 2, 3) ST_FAST(3, 2, 2) ST_FAST(1, 1, 3) ST_FAST(1, 1, 2)      // their data gradients
#undef ST_FAST
  switch (rm) {
    case 1: hipLaunchKernelGGL(conv_stream_kernel<1>, dim3(grid), dim3(256), lds, s, a); break;
    case 2: hipLaunchKernelGGL(conv_stream_kernel<2>, dim3(grid), dim3(256), lds, s, a); break;
    case 3: hipLaunchKernelGGL(conv_stream_kernel<3>, dim3(grid), dim3(256), lds, s, a); break;
    default: return XPT_ERR_ARG;
  }
  return xpt_launch_status();
}

}  // namespace

extern "C" int xpt_conv2d_stream_tune(int enable, int min_tiles, int wgs_per_cu, int max_lds_kib) {
  if (min_tiles < 0 || wgs_per_cu < 0 || wgs_per_cu > 8 || max_lds_kib < 0 || max_lds_kib > 160) return XPT_ERR_ARG;
  g_st_enable = enable != 0;
  g_st_fast = enable != 2;                 // (2: the generic kernel for every shape)
  if (min_tiles > 0) g_st_min_tiles = min_tiles;
  if (wgs_per_cu > 0) g_st_wgs_per_cu = wgs_per_cu;
  if (max_lds_kib > 0) g_st_max_lds = max_lds_kib * 1024;
  return XPT_OK;
}

/* 1 when xpt_conv2d_fwd_stream / xpt_conv2d_bwd_data_stream serve this 3 x 3 stride-1 layer (pixels = B x OH x OW of the grid
 * the launch enumerates -- data gradient with fold2x2: B x 2 IH x 2 IW --, out / red channels of THAT launch), else 0. */
extern "C" int xpt_conv2d_stream_serves(int B, int OH, int OW, int out_channels, int red_channels, int KH, int KW, int stride,
                                        int upsample_or_fold) {
  const bool k5s2 = KH == 5 && KW == 5 && stride == 2 && !upsample_or_fold;
  if ((!(KH == 3 && KW == 3 && stride == 1) && !k5s2) || B <= 0 || OH <= 0 || OW <= 0) return 0;
  StArgs a{};
  a.B = B; a.OH = OH; a.OW = OW; a.N = out_channels; a.C = red_channels; a.shift = 0; a.quad = 0; a.sgn = 1; a.k5s2 = k5s2;
  // (forward with a nearest-2x input: smaller halo; data gradient with the fold: quad -- the worst case of the two decides)
  a.xbytes = 1; a.wbytes = 1;
  if (upsample_or_fold && ((OH | OW) & 1)) return 0;
  int rm; size_t lds;
  return st_plan(a, rm, lds) ? 1 : 0;
}

extern "C" int xpt_conv2d_fwd_stream(const void* x, const void* w, const float* bias, void* y, int B, int PH, int PW, int C,
                                     long long xpitch, int N, int pad_t, int pad_l, int OH, int OW, long long ypitch, int upsample,
                                     float slope, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(w); XPT_CHECK_PTR(y);
  if (B <= 0 || PH <= 0 || PW <= 0 || C <= 0 || N <= 0 || OH <= 0 || OW <= 0) return XPT_ERR_SHAPE;
  if (C % 8 != 0 || xpitch < C || xpitch % 8 != 0 || ypitch < N || ((uintptr_t)x) % 16 != 0 || ((uintptr_t)w) % 16 != 0)
    return XPT_ERR_ARG;
  if ((upsample != 0 && upsample != 1) || pad_t < 0 || pad_l < 0) return XPT_ERR_ARG;
  StArgs a{};
  a.x = (const unsigned short*)x; a.w = (const unsigned short*)w; a.bias = bias; a.y = (unsigned short*)y;
  a.xpitch = xpitch; a.ypitch = ypitch;
  a.xbytes = ((long long)B * PH * PW - 1) * xpitch * 2 + (long long)C * 2;
  a.wbytes = (long long)N * 9 * C * 2;
  a.ybytes = ((long long)B * OH * OW - 1) * ypitch * 2 + (long long)N * 2;
  a.B = B; a.PH = PH; a.PW = PW; a.shift = upsample; a.Hlim = PH << upsample; a.Wlim = PW << upsample;
  a.C = C; a.N = N;
  a.sgn = 1; a.off_h = -pad_t; a.off_w = -pad_l;
  a.OH = OH; a.OW = OW; a.quad = 0; a.slope = slope;
  if ((long long)(OH - 1) - pad_t >= a.Hlim || (long long)(OW - 1) - pad_l >= a.Wlim) return XPT_ERR_SHAPE;
  int rm; size_t lds;
  if (!st_plan(a, rm, lds)) return XPT_ERR_ARG;
  return launch_st(a, rm, lds, (hipStream_t)stream);
}

/* the same for a 5 x 5 stride-2 layer (PoseNetImproved's first two convolutions, pose_net.py:60-61); upsample must be 0 */
extern "C" int xpt_conv2d_fwd_stream_k5s2(const void* x, const void* w, const float* bias, void* y, int B, int PH, int PW, int C,
                                          long long xpitch, int N, int pad_t, int pad_l, int OH, int OW, long long ypitch, float slope,
                                          void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(w); XPT_CHECK_PTR(y);
  if (B <= 0 || PH <= 0 || PW <= 0 || C <= 0 || N <= 0 || OH <= 0 || OW <= 0) return XPT_ERR_SHAPE;
  if (C % 8 != 0 || xpitch < C || xpitch % 8 != 0 || ypitch < N || ((uintptr_t)x) % 16 != 0 || ((uintptr_t)w) % 16 != 0)
    return XPT_ERR_ARG;
  if (pad_t < 0 || pad_l < 0) return XPT_ERR_ARG;
  StArgs a{};
  a.x = (const unsigned short*)x; a.w = (const unsigned short*)w; a.bias = bias; a.y = (unsigned short*)y;
  a.xpitch = xpitch; a.ypitch = ypitch;
  a.xbytes = ((long long)B * PH * PW - 1) * xpitch * 2 + (long long)C * 2;
  a.wbytes = (long long)N * 25 * C * 2;
  a.ybytes = ((long long)B * OH * OW - 1) * ypitch * 2 + (long long)N * 2;
  a.B = B; a.PH = PH; a.PW = PW; a.shift = 0; a.Hlim = PH; a.Wlim = PW;
  a.C = C; a.N = N;
  a.sgn = 1; a.off_h = -pad_t; a.off_w = -pad_l;
  a.OH = OH; a.OW = OW; a.quad = 0; a.slope = slope; a.k5s2 = 1;
  if ((long long)(OH - 1) * 2 - pad_t >= PH || (long long)(OW - 1) * 2 - pad_l >= PW) return XPT_ERR_SHAPE;
  int rm; size_t lds;
  if (!st_plan(a, rm, lds)) return XPT_ERR_ARG;
  return launch_st(a, rm, lds, (hipStream_t)stream);
}

extern "C" int xpt_conv2d_bwd_data_stream(const void* g, const void* wb, void* dx, int B, int OH, int OW, int Np, long long gpitch,
                                          int C, int pad_t, int pad_l, int IH, int IW, long long dxpitch, int fold2x2, void* stream) {
  XPT_CHECK_PTR(g); XPT_CHECK_PTR(wb); XPT_CHECK_PTR(dx);
  if (B <= 0 || OH <= 0 || OW <= 0 || Np <= 0 || C <= 0 || IH <= 0 || IW <= 0) return XPT_ERR_SHAPE;
  if (Np % 8 != 0 || gpitch < Np || gpitch % 8 != 0 || dxpitch < C || ((uintptr_t)g) % 16 != 0 || ((uintptr_t)wb) % 16 != 0)
    return XPT_ERR_ARG;
  if (fold2x2 != 0 && fold2x2 != 1) return XPT_ERR_ARG;
  StArgs a{};
  a.x = (const unsigned short*)g; a.w = (const unsigned short*)wb; a.bias = nullptr; a.y = (unsigned short*)dx;
  a.xpitch = gpitch; a.ypitch = dxpitch;
  a.xbytes = ((long long)B * OH * OW - 1) * gpitch * 2 + (long long)Np * 2;
  a.wbytes = (long long)C * 9 * Np * 2;
  a.ybytes = ((long long)B * IH * IW - 1) * dxpitch * 2 + (long long)C * 2;
  a.B = B; a.PH = OH; a.PW = OW; a.shift = 0; a.Hlim = OH; a.Wlim = OW;
  a.C = Np; a.N = C;
  a.sgn = -1; a.off_h = pad_t; a.off_w = pad_l;
  a.OH = fold2x2 ? 2 * IH : IH; a.OW = fold2x2 ? 2 * IW : IW; a.quad = fold2x2; a.slope = 1.f;
  if (a.OH != OH || a.OW != OW) return XPT_ERR_SHAPE;                      // stride 1: the gradient grid is the (up-sampled) input grid
  int rm; size_t lds;
  if (!st_plan(a, rm, lds)) return XPT_ERR_ARG;
  return launch_st(a, rm, lds, (hipStream_t)stream);
}
