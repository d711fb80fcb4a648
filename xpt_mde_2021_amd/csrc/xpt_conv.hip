// xpt_conv.hip -- the dense k x k convolutions of PoseNetImproved and of the depth decoder as implicit GEMMs on the
// gfx950 bf16 matrix cores: forward (+ bias + LeakyReLU, TF-SAME asymmetric zero padding, nearest-2x up-sampled input,
// output written into a channel slice of a wider tensor), data gradient (the same kernel in its transposed mode, + the
// 2x2 fold of a nearest-2x input) and a per-step weight packer.  The weight gradient lives in xpt_conv_wgrad.hip.
//
// Replaces keras Conv2D(padding="same") as built by CustomConv2D (model/model_util/layer_ops.py:5-36) for
// PoseNetImproved (model/build_model/pose_net.py:57-91) and DepthNetNoResize's decoder
// (model/build_model/depth_net.py:101-109, 137-167: UpSampling2D(2, "nearest") -> conv -> concat -> conv), and the
// tape.gradient of those layers w.r.t. their inputs (model/train_val.py:85-86).
//
// Why own kernels: no library workspace (MIOpen's bf16 backward solvers accumulate in an fp32 workspace handed out by
// the capture allocator; inside the captured training step that path returned one 64-channel tile of garbage from the
// second replay on -- DESIGN.md section 6), and the pad / up-sample / concat / bias / activation launches around a
// library convolution disappear.
//
// GEMM view.  D[n][m] = sum_{tap, c} Wp[n][tap][c] * X[pixel(m, tap)][c]: A = packed weights (rows n, k-contiguous),
// B = NHWC activations (columns = pixels, k-contiguous channels) -- both operands are k-contiguous, which is the operand
// layout of v_mfma_f32_32x32x16_bf16 (lane (r = lane & 31, h = lane >> 5) holds A[row r][8h..8h+7] and B[8h..8h+7][col r]),
// so every fragment is ONE 16-byte global load with no LDS staging and no transposition.  The accumulator of a lane is
// 4 groups of 4 CONSECUTIVE output channels of one pixel: 8-byte bf16 stores into the NHWC output.
// A wave owns RM x RN tiles of 32 x 32; the K loop keeps G k-steps of loads in flight ahead of their MFMAs.
//   large pixel counts: 4 waves of a workgroup stack along the pixel axis (weights shared through the L1);
//   small pixel counts (the 2x7 ... 8x26 maps): the NKW waves of a workgroup split the K loop of ONE tile and add their
//   accumulators through LDS, so that even a 112-pixel layer spreads over hundreds of waves.
// Transposed mode (data gradient of a stride-s convolution): input coordinate t = o + pad - k must be a multiple of s;
// output pixels are enumerated per residue class (blockIdx.z) so that only the taps that contribute are visited.
#include "xpt_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef xpt_h16x8 bf16x8;      // (8 operands of the build's 16-bit format, xpt_common.h)

__device__ inline unsigned short f2bf(float f) {   // round to nearest even (inputs are finite or NaN-preserving enough here)
  return xpt_f2h(f);
}

struct ConvArgs {
  const unsigned short* x;   // NHWC bf16 activations, physical extent PH x PW, pixel pitch xpitch elements
  const unsigned short* w;   // packed weights [N][T][C] bf16 (T = KH*KW taps, C = padded reduction channels)
  const float* bias;         // [N] or null
  unsigned short* y;         // NHWC bf16 output, pixel pitch ypitch elements
  long long xpitch, ypitch;
  int B, PH, PW;             // physical input extent
  int Hlim, Wlim;            // logical extent the taps index: PH << shift
  int shift;                 // 0 direct; 1 nearest-2x input (source = t >> 1) or transposed with forward stride 2 (t / 2)
  int C, N, KH, KW;
  int so;                    // output -> input step (the forward stride; 1 in transposed mode)
  int sgn;                   // +1: t = o*so + k + off; -1 (transposed): t = o + off - k
  int off_h, off_w;          // -pad (forward) / +pad (transposed)
  int OH, OW;                // pixel grid the kernel enumerates
  int xs;                    // residue classes per axis (transposed with forward stride 2: 2; else 1)
  int quad;                  // 1: pixels are enumerated as (parent, child) and the 4 children are summed into the parent
  float slope;               // LeakyReLU slope of the epilogue (1 = linear)
  int xcd;                   // 1: workgroups renumbered image-major onto the XCDs (xpt_common.h)
};

template <int RM, int RN, int NKW>
__global__ __launch_bounds__(NKW > 1 ? 64 * NKW : 256) void conv_igemm_kernel(ConvArgs a) {
  constexpr int G = (RM * RN >= 4) ? 4 : 8;      // k-steps whose loads are in flight together
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;

  // residue class of this part of the grid (transposed stride-2 mode), its pixel grid and its taps
  const int cls = blockIdx.z, ch = cls / a.xs, cw = cls % a.xs;
  const int rows_c = (a.OH - ch + a.xs - 1) / a.xs, cols_c = (a.OW - cw + a.xs - 1) / a.xs;
  const long long Mc = (long long)a.B * rows_c * cols_c;
  int kh0 = 0, kw0 = 0, kstep = 1;
  if (a.xs > 1) {
    kh0 = (ch + a.off_h) % a.xs;
    kw0 = (cw + a.off_w) % a.xs;
    kstep = a.xs;
  }
  const int nkh = kh0 < a.KH ? (a.KH - kh0 + kstep - 1) / kstep : 0;
  const int nkw = kw0 < a.KW ? (a.KW - kw0 + kstep - 1) / kstep : 0;
  const int csteps = (a.C + 15) >> 4;
  const int nsteps = nkh * nkw * csteps;

  unsigned bx = blockIdx.x, by = blockIdx.y;
  xpt_xcd_remap(a.xcd != 0, bx, by, gridDim.x, gridDim.y);
  const long long tile_m = (NKW > 1) ? (long long)by : (long long)by * 4 + wave;
  const long long m0 = tile_m * (32 * RN);
  const int n0 = bx * (32 * RM);
  if (m0 >= Mc) return;                          // uniform per wave; with NKW > 1 uniform per workgroup

  // ---- per-lane pixel geometry (columns of the B operand)
  int pb[RN], poh[RN], pow_[RN];
  bool pok[RN];
#pragma unroll
  for (int j = 0; j < RN; ++j) {
    long long m = m0 + 32 * j + r;
    pok[j] = m < Mc;
    if (!pok[j]) m = Mc - 1;
    int oh, ow, b;
    // (pixel index below 2^31 -- the launcher checks --: split without the ~120-instruction 64-bit divisions)
    if (a.quad) {                                // m = parent * 4 + child over the (OH/2) x (OW/2) parent grid
      const int child = (int)(m & 3);
      unsigned c_, rr_;
      const unsigned q_ = xpt_divmod((unsigned)(m >> 2), (unsigned)(a.OW >> 1), c_);
      b = (int)xpt_divmod(q_, (unsigned)(a.OH >> 1), rr_);
      oh = 2 * (int)rr_ + (child >> 1);
      ow = 2 * (int)c_ + (child & 1);
    } else {
      unsigned c_, rr_;
      const unsigned q_ = xpt_divmod((unsigned)m, (unsigned)cols_c, c_);
      b = (int)xpt_divmod(q_, (unsigned)rows_c, rr_);
      oh = (int)rr_ * a.xs + ch;
      ow = (int)c_ * a.xs + cw;
    }
    pb[j] = b;
    poh[j] = oh;
    pow_[j] = ow;
  }
  int base_h[RN], base_w[RN];
#pragma unroll
  for (int j = 0; j < RN; ++j) {
    base_h[j] = poh[j] * a.so + a.off_h;
    base_w[j] = pow_[j] * a.so + a.off_w;
  }
  // ---- weight rows (rows of the A operand)
  const int T = a.KH * a.KW;
  const unsigned short* wrow[RM];
  bool nok[RM];
#pragma unroll
  for (int i = 0; i < RM; ++i) {
    const int n = n0 + 32 * i + r;
    nok[i] = n < a.N;
    wrow[i] = a.w + (long long)(nok[i] ? n : a.N - 1) * T * a.C;
  }

  f32x16 acc[RM][RN];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // ---- K loop: this wave's range of k-steps (all of them unless the workgroup splits K)
  int s_begin = 0, s_end = nsteps;
  if (NKW > 1) {
    const int per = (nsteps + NKW - 1) / NKW;
    s_begin = wave * per;
    s_end = s_begin + per < nsteps ? s_begin + per : nsteps;
  }
  // running (tap row, tap column, channel step) of the NEXT step to issue
  int it = 0, khi = 0, kwi = 0, cs = 0;
  if (s_begin > 0 && s_begin < nsteps) {
    it = s_begin / csteps;
    cs = s_begin % csteps;
    khi = it / nkw;
    kwi = it % nkw;
  }
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  for (int s0 = s_begin; s0 < s_end; s0 += G) {
    uint4 fa[G][RM], fb[G][RN];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int kh = kh0 + khi * kstep, kw = kw0 + kwi * kstep;
      const int c0 = cs * 16 + 8 * h;
      const bool live = (s0 + g < s_end) && (c0 < a.C);        // steps past the range load from offset 0 and are zeroed
      const int cc = live ? c0 : 0;
      const long long woff = (long long)(kh * a.KW + kw) * a.C + cc;
#pragma unroll
      for (int i = 0; i < RM; ++i) {
        const uint4 v = *(const uint4*)(wrow[i] + (s0 + g < s_end ? woff : 0));
        fa[g][i] = (live && nok[i]) ? v : zero4;
      }
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        const int th = base_h[j] + a.sgn * kh, tw = base_w[j] + a.sgn * kw;
        const bool ok = live && pok[j] && th >= 0 && th < a.Hlim && tw >= 0 && tw < a.Wlim;
        const int row = ok ? (th >> a.shift) : 0, col = ok ? (tw >> a.shift) : 0;
        const long long xoff = (((long long)pb[j] * a.PH + row) * a.PW + col) * a.xpitch + cc;
        const uint4 v = *(const uint4*)(a.x + (ok ? xoff : 0));
        fb[g][j] = ok ? v : zero4;
      }
      // advance to the next step (scalar state)
      if (++cs == csteps) {
        cs = 0;
        if (++kwi == nkw) {
          kwi = 0;
          ++khi;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (s0 + g < s_end) {                                    // wave-uniform
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
          for (int j = 0; j < RN; ++j)
            acc[i][j] = XPT_MFMA_32X32X16(__builtin_bit_cast(bf16x8, fa[g][i]),
                                                                __builtin_bit_cast(bf16x8, fb[g][j]), acc[i][j]);
      }
    }
  }

  // ---- split-K: add the waves' accumulators through LDS; wave w then owns the register groups q with q % NKW == w
  __shared__ float red[(NKW > 1) ? NKW * 16 * 64 : 1];
  if (NKW > 1) {
#pragma unroll
    for (int q = 0; q < 16; ++q) red[(wave * 16 + q) * 64 + lane] = acc[0][0][q];
    __syncthreads();
    if (wave >= 4) return;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (((q >> 2) % (NKW < 4 ? NKW : 4)) != (wave % 4)) continue;      // wave-uniform: only this wave's groups
      float sum = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NKW; ++w2) sum += red[(w2 * 16 + q) * 64 + lane];
      acc[0][0][q] = sum;
    }
  }

  // ---- epilogue.  Register q of tile (i, j): channel n0 + 32 i + (q & 3) + 8 (q >> 2) + 4 h, pixel m0 + 32 j + r
  const bool vec_ok = (a.N % 4 == 0) && (a.ypitch % 4 == 0) && (((uintptr_t)a.y) % 8 == 0);
#pragma unroll
  for (int j = 0; j < RN; ++j) {
    long long opix;
    bool store = pok[j];
    if (a.quad) {
      opix = (m0 + 32 * j + r) >> 2;
    } else {
      opix = ((long long)pb[j] * a.OH + poh[j]) * a.OW + pow_[j];
    }
#pragma unroll
    for (int i = 0; i < RM; ++i) {
#pragma unroll
      for (int qg = 0; qg < 4; ++qg) {
        if (NKW > 1 && (qg % (NKW < 4 ? NKW : 4)) != (wave % 4)) continue;   // wave-uniform: this wave's share
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * qg + e];
        if (a.quad) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] += __shfl_xor(v[e], 1, 64);
            v[e] += __shfl_xor(v[e], 2, 64);
          }
        }
        const int n = n0 + 32 * i + 8 * qg + 4 * h;
        if (!store || (a.quad && (r & 3) != 0) || n >= a.N) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (a.bias != nullptr && n + e < a.N) v[e] += a.bias[n + e];
          v[e] = v[e] > 0.f ? v[e] : v[e] * a.slope;
        }
        unsigned short* dst = a.y + opix * a.ypitch + n;
        if (vec_ok) {
          uint2 pk;
          pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
          pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
          *(uint2*)dst = pk;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < a.N) dst[e] = f2bf(v[e]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ LDS-staged variant
// The direct kernel above issues one 16-byte load per lane per fragment; the 64 lanes of such a load touch 32 different
// cache lines (32 rows x 32 bytes), so on the large maps it runs at the L1's line rate, not at its byte rate (measured:
// ~2x slower than the library on the 64x208 ... 128x416 layers).  This variant is the classic tiling for those layers:
// a workgroup owns TN = 32 RM output channels x 128 pixels, the K axis is enumerated in 16-byte "pieces" (8 channels of
// one tap; K = taps x C/8 pieces, dense even when C is 16) and consumed in chunks of 8 pieces (64 elements).  All 256
// threads stage a chunk with COALESCED loads (8 consecutive lanes read 128 contiguous bytes of one pixel / one weight
// row) into LDS rows of 64 elements padded to 144 bytes (ds_read_b128 of 16 consecutive rows is conflict free), the next
// chunk is fetched into registers while the MFMAs of the current one run.  Pixel decoding, residue classes, the quad fold
// and the epilogue are those of the direct kernel, so every mode it serves is served here.
template <int RM>
__global__ __launch_bounds__(256) void conv_lds_kernel(ConvArgs a) {
  constexpr int TN = 32 * RM, TP = 128, PITCH = 64 * 2 + 16;
  __shared__ __attribute__((aligned(16))) unsigned char lds[(TN + TP) * PITCH];
  unsigned char* const lA = lds;
  unsigned char* const lB = lds + TN * PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  const int cls = blockIdx.z, ch = cls / a.xs, cw = cls % a.xs;
  const int rows_c = (a.OH - ch + a.xs - 1) / a.xs, cols_c = (a.OW - cw + a.xs - 1) / a.xs;
  const long long Mc = (long long)a.B * rows_c * cols_c;
  int kh0 = 0, kw0 = 0, kstep = 1;
  if (a.xs > 1) {
    kh0 = (ch + a.off_h) % a.xs;
    kw0 = (cw + a.off_w) % a.xs;
    kstep = a.xs;
  }
  const int nkh = kh0 < a.KH ? (a.KH - kh0 + kstep - 1) / kstep : 0;
  const int nkw = kw0 < a.KW ? (a.KW - kw0 + kstep - 1) / kstep : 0;
  const int cp8 = a.C >> 3;
  const int npieces = nkh * nkw * cp8;
  const int nchunks = (npieces + 7) >> 3;
  const float inv_cp8 = 1.f / (float)cp8, inv_nkw = nkw > 0 ? 1.f / (float)nkw : 0.f;

  unsigned bx = blockIdx.x, by = blockIdx.y;
  xpt_xcd_remap(a.xcd != 0, bx, by, gridDim.x, gridDim.y);
  const long long m0 = (long long)by * TP;
  const int n0 = bx * TN;
  if (m0 >= Mc) return;                                              // uniform per workgroup

  // (32-bit index arithmetic: the launcher refuses pixel grids of 2^31 or more; a 64-bit division costs ~10x a 32-bit one
  //  and a thread decodes five pixels)
  const unsigned Mc32 = (unsigned)Mc;
  auto decode = [&](long long m64, int& b, int& oh, int& ow) -> bool {
    const bool ok = m64 < Mc;
    const unsigned m = ok ? (unsigned)m64 : Mc32 - 1u;
    if (a.quad) {
      const unsigned child = m & 3u;
      unsigned c, rr;
      const unsigned q1 = xpt_divmod(m >> 2, (unsigned)a.OW >> 1, c);
      b = (int)xpt_divmod(q1, (unsigned)a.OH >> 1, rr);
      oh = (int)(2u * rr + (child >> 1));
      ow = (int)(2u * c + (child & 1u));
    } else {
      unsigned c, rr;
      const unsigned q1 = xpt_divmod(m, (unsigned)cols_c, c);
      b = (int)xpt_divmod(q1, (unsigned)rows_c, rr);
      oh = (int)rr * a.xs + ch;
      ow = (int)c * a.xs + cw;
    }
    return ok;
  };

  // ---- staging role of this thread: piece slot sp of the rows srow + 32 i
  const int sp = tid & 7, srow = tid >> 3;
  int sb[4], base_h[4], base_w[4];
  bool sok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int b, oh, ow;
    sok[i] = decode(m0 + srow + 32 * i, b, oh, ow);
    sb[i] = b;
    base_h[i] = oh * a.so + a.off_h;
    base_w[i] = ow * a.so + a.off_w;
  }
  const int T = a.KH * a.KW;
  const unsigned short* wrow[RM];
  bool nok[RM];
#pragma unroll
  for (int i = 0; i < RM; ++i) {
    const int n = n0 + srow + 32 * i;
    nok[i] = n < a.N;
    wrow[i] = a.w + (long long)(nok[i] ? n : a.N - 1) * T * a.C;
  }

  f32x16 acc[RM];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

  uint4 ra[RM], rb[4];
  auto fetch = [&](int ck) {
    const int p = ck * 8 + sp;
    const bool live = p < npieces;
    const int it = live ? (int)(((float)p + 0.5f) * inv_cp8) : 0;     // tap index of the piece (exact: operands are small)
    const int c0 = live ? (p - it * cp8) * 8 : 0;
    const int khi = (int)(((float)it + 0.5f) * inv_nkw);
    const int kwi = it - khi * nkw;
    const int kh = kh0 + khi * kstep, kw = kw0 + kwi * kstep;
    const long long woff = (long long)(kh * a.KW + kw) * a.C + c0;
#pragma unroll
    for (int i = 0; i < RM; ++i) {
      const uint4 v = *(const uint4*)(wrow[i] + (live ? woff : 0));
      const unsigned keep = (live && nok[i]) ? 0xffffffffu : 0u;      // and-mask, not a 16-byte select (that went to scratch)
      ra[i] = make_uint4(v.x & keep, v.y & keep, v.z & keep, v.w & keep);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int th = base_h[i] + a.sgn * kh, tw = base_w[i] + a.sgn * kw;
      const bool ok = live && sok[i] && th >= 0 && th < a.Hlim && tw >= 0 && tw < a.Wlim;
      const int row = ok ? (th >> a.shift) : 0, col = ok ? (tw >> a.shift) : 0;
      const long long xoff = (((long long)sb[i] * a.PH + row) * a.PW + col) * a.xpitch + c0;
      const uint4 v = *(const uint4*)(a.x + (ok ? xoff : 0));
      const unsigned keep = ok ? 0xffffffffu : 0u;
      rb[i] = make_uint4(v.x & keep, v.y & keep, v.z & keep, v.w & keep);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < RM; ++i) *(uint4*)(lA + (srow + 32 * i) * PITCH + sp * 16) = ra[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *(uint4*)(lB + (srow + 32 * i) * PITCH + sp * 16) = rb[i];
  };

  if (nchunks > 0) {
    fetch(0);
    stash();
    __syncthreads();
  }
  const unsigned char* const fB = lB + (wave * 32 + r) * PITCH + h * 16;
  const unsigned char* const fA = lA + r * PITCH + h * 16;
  for (int ck = 0; ck < nchunks; ++ck) {
    const bool more = ck + 1 < nchunks;
    if (more) fetch(ck + 1);
    const int left = npieces - ck * 8;                                // pieces of this chunk that hold data (uniform)
#pragma unroll
    for (int k16 = 0; k16 < 4; ++k16) {
      if (2 * k16 < left) {
        const uint4 fb = *(const uint4*)(fB + k16 * 32);
#pragma unroll
        for (int i = 0; i < RM; ++i) {
          const uint4 fa = *(const uint4*)(fA + 32 * i * PITCH + k16 * 32);
          acc[i] = XPT_MFMA_32X32X16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb),
                                                           acc[i]);
        }
      }
    }
    if (more) {
      __syncthreads();
      stash();
      __syncthreads();
    }
  }

  // ---- epilogue (as in the direct kernel; this wave's pixels are m0 + 32 wave + r)
  int pb, poh, pow_;
  const long long m = m0 + 32 * wave + r;
  const bool store = decode(m, pb, poh, pow_);
  const long long opix = a.quad ? (m >> 2) : ((long long)pb * a.OH + poh) * a.OW + pow_;
  const bool vec_ok = (a.N % 4 == 0) && (a.ypitch % 4 == 0) && (((uintptr_t)a.y) % 8 == 0);
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
      const int n = n0 + 32 * i + 8 * qg + 4 * h;
      if (!store || (a.quad && (r & 3) != 0) || n >= a.N) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (a.bias != nullptr && n + e < a.N) v[e] += a.bias[n + e];
        v[e] = v[e] > 0.f ? v[e] : v[e] * a.slope;
      }
      unsigned short* dst = a.y + opix * a.ypitch + n;
      if (vec_ok) {
        uint2 pk;
        pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
        pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
        *(uint2*)dst = pk;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < a.N) dst[e] = f2bf(v[e]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ halo-tile variant
// Stride-1 layers on the large maps (the decoder from 32 x 104 up, forward and data gradient).  The im2col view of the two
// kernels above fetches every input pixel once per tap (9 x for a 3 x 3 filter) and, in the LDS-staged kernel, pays one
// memory round trip per 64 reduction elements with 4 RM MFMAs per wave behind it: those layers ran at one global-load
// latency per chunk (22 - 33 us for 3 - 4 GFLOP and 20 MB).  Here a workgroup owns an 8 x 16 tile of output pixels and
// TN = 32 RM output channels; per chunk of CC input channels it stages the tile's INPUT HALO ((8 + KH - 1) x (16 + KW - 1)
// logical pixels, fewer physical ones when the input is up-sampled) and the weight slab [TN][taps][CC] in LDS -- every
// input element is fetched once, in one burst of coalesced 16-byte loads -- and the taps are LDS address offsets:
// taps x CC / 16 x RM MFMAs per wave and round trip (36 RM for a 3 x 3 filter on 64 channels).  Wave w owns rows 2w, 2w+1
// of the tile; pixel decoding differs from the kernels above (tiles, not a flat pixel index), modes and epilogue are theirs.
struct HaloPlan {
  int CC;          // input channels per chunk (16 / 32 / 64)
  int HR, WR;      // physical halo extent (upper bound used for the LDS layout)
  int XP, WP;      // LDS pitches in bytes: per halo pixel, per weight row
  int tiles_x, tiles_y;
};

template <int RM, bool K3>
__global__ __launch_bounds__(256) void conv_halo_kernel(ConvArgs a, HaloPlan hp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hl[];
  constexpr int TN = 32 * RM;
  const int XBYTES = hp.HR * hp.WR * hp.XP;
  unsigned char* const lX = hl;
  unsigned char* const lW = hl + XBYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int T = a.KH * a.KW;

  unsigned bx = blockIdx.x, by = blockIdx.y;
  xpt_xcd_remap(a.xcd != 0, bx, by, gridDim.x, gridDim.y);
  int tile = (int)by;
  const int txi = tile % hp.tiles_x; tile /= hp.tiles_x;
  const int tyi = tile % hp.tiles_y;
  const int b = tile / hp.tiles_y;
  const int oh0 = tyi * 8, ow0 = txi * 16;
  const int n0 = bx * TN;

  // logical input window of the tile and its physical footprint
  const int lo_h = oh0 + a.off_h - (a.sgn > 0 ? 0 : a.KH - 1), lo_w = ow0 + a.off_w - (a.sgn > 0 ? 0 : a.KW - 1);
  const int plo_h = lo_h >> a.shift, plo_w = lo_w >> a.shift;      // arithmetic shifts: floor for negative coordinates
  const int WR = hp.WR, HPIX = hp.HR * WR;

  // this lane's output pixel
  int ty, tx;
  if (a.quad) {
    const int child = r & 3;
    ty = 2 * wave + (child >> 1);
    tx = 2 * (r >> 2) + (child & 1);
  } else {
    ty = 2 * wave + (r >> 4);
    tx = r & 15;
  }
  const int oh = oh0 + ty, ow = ow0 + tx;
  const bool pok = oh < a.OH && ow < a.OW;
  const int base_h = (pok ? oh : oh0) + a.off_h, base_w = (pok ? ow : ow0) + a.off_w;

  f32x16 acc[RM];
#pragma unroll
  for (int i = 0; i < RM; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

  // staging: vector v < xvecs of the halo = (pixel v >> lc, 8-channel group v & (cv8 - 1)); the vectors behind them are the
  // weight slab, (row, tap, group).  cv8 is a power of two; the two divisions by small run-time numbers (halo width, taps)
  // are exact float multiplications (an integer division costs ~40 instructions).  No register prefetch across chunks: it
  // cost 200+ VGPRs (2 workgroups per CU); at ~90 VGPRs the other workgroups of the CU cover the round trip.  A thread
  // issues SB loads before the first LDS store.  Rows of the slab past N are not staged (their accumulator rows are
  // never stored).
  const int CC = hp.CC, cv8 = CC >> 3, lc = cv8 == 8 ? 3 : (cv8 == 4 ? 2 : 1);
  const int nrows = a.N - n0 < TN ? a.N - n0 : TN;
  const int xvecs = HPIX * cv8, nvecs = xvecs + nrows * T * cv8;
  const float inv_wr = 1.f / (float)WR, inv_t = 1.f / (float)T;
  constexpr int SB = 8;
  auto stage = [&](int c0) {
    for (int v0 = tid; v0 < nvecs; v0 += 256 * SB) {
      uint4 val[SB];
      int lo[SB];
#pragma unroll
      for (int i = 0; i < SB; ++i) {
        const int v = v0 + 256 * i;
        const bool isx = v < xvecs;
        const int u = isx ? v : v - xvecs;
        const int cv = u & (cv8 - 1), rest = u >> lc;
        const int c = c0 + cv * 8;
        // halo vector: rest = pixel; slab vector: rest = row * T + tap
        const int q = (int)(((float)rest + 0.5f) * (isx ? inv_wr : inv_t)), rem = rest - q * (isx ? WR : T);
        const int pr = plo_h + q, pc = plo_w + rem;
        const bool ok = v < nvecs && c < a.C && (!isx || (pr >= 0 && pr < a.PH && pc >= 0 && pc < a.PW));
        const long long xoff = (((long long)b * a.PH + pr) * a.PW + pc) * a.xpitch + c;
        const long long woff = ((long long)(n0 + q) * T + rem) * a.C + c;
        const unsigned short* src = isx ? a.x + (ok ? xoff : 0) : a.w + (ok ? woff : 0);
        const uint4 ld = *(const uint4*)src;
        const unsigned keep = ok ? 0xffffffffu : 0u;
        val[i] = make_uint4(ld.x & keep, ld.y & keep, ld.z & keep, ld.w & keep);
        lo[i] = v >= nvecs ? -1 : (isx ? rest * hp.XP + cv * 16 : XBYTES + q * hp.WP + (rem * CC + cv * 8) * 2);
      }
#pragma unroll
      for (int i = 0; i < SB; ++i)
        if (lo[i] >= 0) *(uint4*)(hl + lo[i]) = val[i];
    }
  };

  const int nchunks = (a.C + CC - 1) / CC;
  for (int ck = 0; ck < nchunks; ++ck) {
    if (ck > 0) __syncthreads();                     // the previous chunk's operand reads are done
    stage(ck * CC);
    __syncthreads();
    const int c_left = a.C - ck * CC;
    const int k16n = c_left >= CC ? CC >> 4 : (c_left + 15) >> 4;      // uniform: 16-channel steps of this chunk that hold data
    // (a single 32-row tile with fewer than 32 output channels: rows past N are not in the slab -- any staged row will do)
    const unsigned char* const wA = lW + ((RM == 1 && r >= nrows) ? 0 : r) * hp.WP + h * 16;
    auto tap_mfma = [&](int kh, int kw) {
      const int th = base_h + a.sgn * kh, tw = base_w + a.sgn * kw;
      const int prow = (th >> a.shift) - plo_h, pcol = (tw >> a.shift) - plo_w;
      const unsigned char* const xB = lX + (prow * WR + pcol) * hp.XP + h * 16;
      const unsigned char* const wT = wA + (kh * a.KW + kw) * CC * 2;
      for (int k16 = 0; k16 < k16n; ++k16) {
        const uint4 fb = *(const uint4*)(xB + k16 * 32);
#pragma unroll
        for (int i = 0; i < RM; ++i) {
          const uint4 fa = *(const uint4*)(wT + 32 * i * hp.WP + k16 * 32);
          acc[i] = XPT_MFMA_32X32X16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb),
                                                           acc[i]);
        }
      }
    };
    if constexpr (K3) {                                // 3 x 3 filters (the whole decoder): the taps unrolled
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) tap_mfma(kh, kw);
    } else {
      for (int kh = 0; kh < a.KH; ++kh)
        for (int kw = 0; kw < a.KW; ++kw) tap_mfma(kh, kw);
    }
  }

  // ---- epilogue (that of the kernels above): register q of tile i = channel n0 + 32 i + (q & 3) + 8 (q >> 2) + 4 h
  const long long opix = a.quad ? ((long long)b * (a.OH >> 1) + (oh >> 1)) * (a.OW >> 1) + (ow >> 1)
                                : ((long long)b * a.OH + oh) * a.OW + ow;
  const bool vec_ok = (a.N % 4 == 0) && (a.ypitch % 4 == 0) && (((uintptr_t)a.y) % 8 == 0);
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
      const int n = n0 + 32 * i + 8 * qg + 4 * h;
      if (!pok || (a.quad && (r & 3) != 0) || n >= a.N) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (a.bias != nullptr && n + e < a.N) v[e] += a.bias[n + e];
        v[e] = v[e] > 0.f ? v[e] : v[e] * a.slope;
      }
      unsigned short* dst = a.y + opix * a.ypitch + n;
      if (vec_ok) {
        uint2 pk;
        pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
        pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
        *(uint2*)dst = pk;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < a.N) dst[e] = f2bf(v[e]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight packer
// One launch per step converts every dense convolution weight from its fp32 master copy (any strides; the flat parameter
// buffer keeps them in channels_last order [N][KH][KW][C]) into the two bf16 operand layouts of the kernels above:
//   fwd [N][T][Cp]   (Cp = C rounded up to a multiple of 8, zero filled)
//   bwd [Cp][T][Np]  (the transposed-mode operand: rows = input channels (rows >= C zero, so that the data gradient of a
//                     padded channel is written as 0), reduction = output channels, Np = N rounded up to 8, zero filled)
struct PackJob {
  const float* src;          // fp32 master weight; element (n, kh, kw, c) at n*sn + kh*sh + kw*sw + c*sc
  unsigned short* fwd;
  unsigned short* bwd;       // may be null (layer whose input needs no gradient)
  long long sn, sc, sh, sw;
  int N, T, KW, C, Cp, Np;
  long long first_block;     // first workgroup of this job in the launch
};

__global__ __launch_bounds__(256) void conv_pack_kernel(const PackJob* __restrict__ jobs, int njobs) {
  // find the job of this workgroup (jobs are few: linear scan by one thread, broadcast through LDS)
  __shared__ int sj;
  __shared__ float tile[64][65];            // [n][c] of one tap; 65: conflict-free transposed reads
  if (threadIdx.x == 0) {
    int j = 0;
    while (j + 1 < njobs && jobs[j + 1].first_block <= (long long)blockIdx.x) ++j;
    sj = j;
  }
  __syncthreads();
  const PackJob jb = jobs[sj];
  // a workgroup owns one tap and a 64 (output channels) x 64 (input channels) tile: the fp32 master is read along its
  // contiguous axis, the forward layout [n][t][c] is written along c and the backward layout [c][t][n] along n (through
  // the LDS transpose) -- one thread per element with a strided read made this launch 64 us per step
  const int tiles_c = (jb.Cp + 63) / 64;
  const int nmax = jb.Np > jb.N ? jb.Np : jb.N;
  const int tiles_n = (nmax + 63) / 64;
  int blk = (int)((long long)blockIdx.x - jb.first_block);
  if (blk >= jb.T * tiles_n * tiles_c) return;
  const int tc = blk % tiles_c; blk /= tiles_c;
  const int tn = blk % tiles_n;
  const int t = blk / tiles_n;
  const int n0 = tn * 64, c0 = tc * 64;
  const int kh = t / jb.KW, kw = t - kh * jb.KW;
  const bool c_fast = jb.sc <= jb.sn;        // which axis of the master copy is the contiguous one
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int i = c_fast ? e >> 6 : e & 63, j = c_fast ? e & 63 : e >> 6;      // i: n offset, j: c offset
    const int n = n0 + i, c = c0 + j;
    tile[i][j] = (n < jb.N && c < jb.C) ? jb.src[n * jb.sn + kh * jb.sh + kw * jb.sw + c * jb.sc] : 0.f;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {           // forward layout: c fastest
    const int i = e >> 6, j = e & 63;
    const int n = n0 + i, c = c0 + j;
    if (n < jb.N && c < jb.Cp) jb.fwd[((long long)n * jb.T + t) * jb.Cp + c] = f2bf(tile[i][j]);
  }
  if (jb.bwd != nullptr) {
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {         // backward layout: n fastest
      const int j = e >> 6, i = e & 63;
      const int n = n0 + i, c = c0 + j;
      if (c < jb.Cp && n < jb.Np) jb.bwd[((long long)c * jb.T + t) * jb.Np + n] = f2bf(tile[i][j]);
    }
  }
}

// PoseNet input: restack_on_channels (pose_net.py:44-50) of the fp32 snippet [B,S,H,W,3] into bf16 [B,H,W,Cp]
// (channel = frame * 3 + c, channels >= 3 S zero) -- the layout the first convolution reads.
__global__ __launch_bounds__(256) void restack_kernel(const float* __restrict__ img, unsigned short* __restrict__ out,
                                                      int B, int S, long long HW, int Cp) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;     // one thread per (pixel, frame-or-pad slot)
  const int slots = S + ((Cp - 3 * S + 2) / 3);
  const long long total = (long long)B * HW * slots;
  if (i >= total) return;
  unsigned slu, pu;                                   // (total < 2^31: the launcher checks)
  const unsigned bpu = xpt_divmod((unsigned)i, (unsigned)slots, slu);
  const long long b = (long long)xpt_divmod(bpu, (unsigned)HW, pu);
  const int sl = (int)slu;
  const long long bp = (long long)bpu, p = (long long)pu;
  unsigned short* o = out + bp * Cp + 3 * sl;
  if (sl < S) {
    const float* src = img + ((b * S + sl) * HW + p) * 3;
    o[0] = f2bf(src[0]);
    o[1] = f2bf(src[1]);
    o[2] = f2bf(src[2]);
  } else {
    for (int c = 3 * sl; c < Cp && c < 3 * sl + 3; ++c) out[bp * Cp + c] = 0;
  }
}

template <int RM, int RN, int NKW>
int launch_igemm(const ConvArgs& a, long long Mmax, int classes, hipStream_t s) {
  const long long mtiles = (Mmax + 32 * RN - 1) / (32 * RN);
  const long long gy = NKW > 1 ? mtiles : (mtiles + 3) / 4;
  if (gy > 65535 || Mmax >= 0x7fffffffLL) return XPT_ERR_SHAPE;
  const dim3 grid((a.N + 32 * RM - 1) / (32 * RM), (unsigned)gy, classes);
  ConvArgs b = a;
  b.xcd = xpt_xcd_ok((unsigned long long)grid.x * grid.y);
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL((conv_igemm_kernel<RM, RN, NKW>), grid, dim3(NKW > 1 ? 64 * NKW : 256), 0, s, b);
  return xpt_launch_status();
}

template <int RM>
int launch_lds(const ConvArgs& a, long long Mmax, int classes, hipStream_t s) {
  const long long gy = (Mmax + 127) / 128;
  if (gy > 65535 || Mmax >= 0x7fffffffLL) return XPT_ERR_SHAPE;
  const dim3 grid((a.N + 32 * RM - 1) / (32 * RM), (unsigned)gy, classes);
  ConvArgs b = a;
  b.xcd = xpt_xcd_ok((unsigned long long)grid.x * grid.y);
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL((conv_lds_kernel<RM>), grid, dim3(256), 0, s, b);
  return xpt_launch_status();
}

int g_conv_plan = 0;   // 0 = automatic; otherwise RM*100 + RN*10 + log2(NKW), or 900 + RM for the LDS-staged kernel
int g_conv_lds_min_blocks = 512;
int g_conv_halo_min_tiles = 100;   // halo-tile kernel from this many 8 x 16 pixel tiles on (0 disables it)

// the halo-tile kernel's plan for a layer; false when the layer is outside what it serves
bool halo_plan(const ConvArgs& a, int rm, HaloPlan& hp, size_t& lds) {
  if (a.so != 1 || a.xs != 1 || a.C % 8 != 0) return false;
  if (a.quad && ((a.OH | a.OW) & 1)) return false;
  const int T = a.KH * a.KW;
  const int c16 = (a.C + 15) / 16 * 16;
  hp.tiles_x = (a.OW + 15) / 16;
  hp.tiles_y = (a.OH + 7) / 8;
  const int rows = 8 + a.KH - 1, cols = 16 + a.KW - 1;
  hp.HR = a.shift ? rows / 2 + 1 : rows;
  hp.WR = a.shift ? cols / 2 + 1 : cols;
  for (int cc = 64; cc >= 16; cc >>= 1) {
    hp.CC = cc < c16 ? cc : c16;
    hp.XP = hp.CC * 2 + 16;
    hp.WP = T * hp.CC * 2 + 16;
    const int xvecs = hp.HR * hp.WR * (hp.CC / 8), wvecs = 32 * rm * T * (hp.CC / 8);
    const int wrows = (rm == 1 && a.N < 32) ? a.N : 32 * rm;      // rows past N are neither staged nor read
    lds = (size_t)hp.HR * hp.WR * hp.XP + (size_t)wrows * hp.WP;
    (void)xvecs; (void)wvecs;
    if (lds <= 40 * 1024) return true;
  }
  return false;
}

template <int RM>
int launch_halo(const ConvArgs& a, const HaloPlan& hp, size_t lds, hipStream_t s) {
  const long long gy = (long long)a.B * hp.tiles_y * hp.tiles_x;
  if (gy > 65535) return XPT_ERR_SHAPE;
  const dim3 grid((a.N + 32 * RM - 1) / (32 * RM), (unsigned)gy);
  ConvArgs b = a;
  b.xcd = xpt_xcd_ok((unsigned long long)grid.x * grid.y);
  XPT_BEGIN_LAUNCH();
  if (a.KH == 3 && a.KW == 3)
    hipLaunchKernelGGL((conv_halo_kernel<RM, true>), grid, dim3(256), lds, s, b, hp);
  else
    hipLaunchKernelGGL((conv_halo_kernel<RM, false>), grid, dim3(256), lds, s, b, hp);
  return xpt_launch_status();
}

int dispatch_igemm(const ConvArgs& a, hipStream_t s) {
  const int classes = a.xs * a.xs;
  // pixels of the largest residue class
  const long long Mmax = (long long)a.B * ((a.OH + a.xs - 1) / a.xs) * ((a.OW + a.xs - 1) / a.xs);
  const long long ntile = (a.N + 31) / 32, mtile = (Mmax + 31) / 32;
  const long long waves11 = ntile * mtile * classes;
  int rm = 1, rn = 1, nkw = 1;
  int plan = g_conv_plan;
  if ((!plan && g_conv_halo_min_tiles > 0) || plan == 911 || plan == 912) {
    // large stride-1 maps: the halo-tile kernel (every input element fetched once per workgroup, taps as LDS offsets)
    const int hrm = plan == 912 ? 2 : 1;       // (automatic: 32 output channels per workgroup -- more, smaller workgroups won on every layer)
    HaloPlan hp;
    size_t lds = 0;
    if (halo_plan(a, hrm, hp, lds) && (long long)a.B * hp.tiles_y * hp.tiles_x <= 65535) {
      const long long tiles = (long long)a.B * hp.tiles_y * hp.tiles_x * ((a.N + 32 * hrm - 1) / (32 * hrm));
      // measured hot (tools/hot_replay.py, batch 8, forward and data gradient of every decoder layer): faster than the two
      // kernels above on every stride-1 layer with at least ~100 tiles and at most 256 reduction channels (13.9 - 30 us against
      // 22 - 70 us); slower on the 432- / 1056-channel layers (17 - 33 chunks of one round trip each on 16 - 128 tiles) and
      // on PoseNet's 2 x 7 maps
      if (plan || (tiles >= g_conv_halo_min_tiles && a.C <= 256))
        return hrm == 2 ? launch_halo<2>(a, hp, lds, s) : launch_halo<1>(a, hp, lds, s);
    }
    plan = 0;                      // (a forced halo plan on a layer it does not serve: the automatic choice)
  }
  if (plan == 901) return launch_lds<1>(a, Mmax, classes, s);
  if (plan == 902) return launch_lds<2>(a, Mmax, classes, s);
  if (!plan && g_conv_lds_min_blocks > 0) {
    // large maps: the LDS-staged kernel (coalesced staging instead of 32 cache lines per fragment load)
    const int lrm = a.N > 32 ? 2 : 1;
    const long long blocks = ((a.N + 32 * lrm - 1) / (32 * lrm)) * ((Mmax + 127) / 128) * classes;
    if (blocks >= g_conv_lds_min_blocks) return lrm == 2 ? launch_lds<2>(a, Mmax, classes, s) : launch_lds<1>(a, Mmax, classes, s);
  }
  if (plan) {
    rm = plan / 100;
    rn = (plan / 10) % 10;
    nkw = 1 << (plan % 10);
  } else {
    // measured per layer (tools/bench_conv.py, batch 8): 64 x 64 wave tiles pay when the output has at least two
    // 32-channel tiles AND the launch still fills the chip (4 MFMAs per 4 fragment loads); everything else runs best on
    // 32 x 32 tiles at 91 VGPRs (4-5 waves per SIMD), the K loop split over the waves of a workgroup when even those
    // are too few.  (The 32 x 64 instantiation lost everywhere: 155 VGPRs for 2 MFMAs per 3 loads.)
    const long long waves22 = ((a.N + 63) / 64) * ((Mmax + 63) / 64) * classes;
    if (ntile >= 2 && waves22 >= 768) {
      rm = 2;
      rn = 2;
    } else if (waves11 >= 1024) {
    } else if (waves11 >= 384) {
      nkw = 4;
    } else {
      nkw = 8;
    }
  }
  if (rm == 2 && rn == 2 && nkw == 1) return launch_igemm<2, 2, 1>(a, Mmax, classes, s);
  if (rm == 1 && rn == 2 && nkw == 1) return launch_igemm<1, 2, 1>(a, Mmax, classes, s);
  if (rm == 1 && rn == 1 && nkw == 1) return launch_igemm<1, 1, 1>(a, Mmax, classes, s);
  if (rm == 1 && rn == 1 && nkw == 2) return launch_igemm<1, 1, 2>(a, Mmax, classes, s);
  if (rm == 1 && rn == 1 && nkw == 4) return launch_igemm<1, 1, 4>(a, Mmax, classes, s);
  if (rm == 1 && rn == 1 && nkw == 8) return launch_igemm<1, 1, 8>(a, Mmax, classes, s);
  return XPT_ERR_ARG;
}

}  // namespace

extern "C" int xpt_conv2d_tune(int plan) {
  if (plan <= -100000) {        // -100000 - n: minimum tiles for the halo-tile kernel (n = 0 disables it)
    g_conv_halo_min_tiles = -plan - 100000;
    return XPT_OK;
  }
  if (plan <= -1000) {          // -1000 - n: minimum workgroups for the LDS-staged kernel (n = 0 disables it)
    g_conv_lds_min_blocks = -plan - 1000;
    return XPT_OK;
  }
  g_conv_plan = plan;
  return XPT_OK;
}

extern "C" int xpt_conv2d_fwd(const void* x, const void* w, const float* bias, void* y, int B, int PH, int PW, int C,
                              long long xpitch, int N, int KH, int KW, int stride, int pad_t, int pad_l, int OH, int OW,
                              long long ypitch, int upsample, float slope, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(w); XPT_CHECK_PTR(y);
  if (B <= 0 || PH <= 0 || PW <= 0 || C <= 0 || N <= 0 || KH <= 0 || KW <= 0 || OH <= 0 || OW <= 0) return XPT_ERR_SHAPE;
  if (C % 8 != 0 || xpitch < C || xpitch % 8 != 0 || ypitch < N || ((uintptr_t)x) % 16 != 0 || ((uintptr_t)w) % 16 != 0)
    return XPT_ERR_ARG;
  if (stride < 1 || (upsample != 0 && upsample != 1) || pad_t < 0 || pad_l < 0) return XPT_ERR_ARG;
  ConvArgs a{};
  a.x = (const unsigned short*)x; a.w = (const unsigned short*)w; a.bias = bias; a.y = (unsigned short*)y;
  a.xpitch = xpitch; a.ypitch = ypitch;
  a.B = B; a.PH = PH; a.PW = PW; a.shift = upsample; a.Hlim = PH << upsample; a.Wlim = PW << upsample;
  a.C = C; a.N = N; a.KH = KH; a.KW = KW;
  a.so = stride; a.sgn = 1; a.off_h = -pad_t; a.off_w = -pad_l;
  a.OH = OH; a.OW = OW; a.xs = 1; a.quad = 0; a.slope = slope;
  // the last window must start inside the padded input
  if ((long long)(OH - 1) * stride - pad_t >= a.Hlim || (long long)(OW - 1) * stride - pad_l >= a.Wlim) return XPT_ERR_SHAPE;
  return dispatch_igemm(a, (hipStream_t)stream);
}

/* data gradient: g [B,OH,OW,Np] (gradient at the convolution output, channels padded to Np with zeros or any finite
 * values times zero weights) -> dx [B,IH,IW,C] where IH x IW is the PHYSICAL input extent; fold2x2 = the forward input
 * was the nearest-2x up-sampling of x (the gradient of the four children is summed). */
extern "C" int xpt_conv2d_bwd_data(const void* g, const void* wb, void* dx, int B, int OH, int OW, int Np, long long gpitch,
                                   int C, int KH, int KW, int stride, int pad_t, int pad_l, int IH, int IW,
                                   long long dxpitch, int fold2x2, void* stream) {
  XPT_CHECK_PTR(g); XPT_CHECK_PTR(wb); XPT_CHECK_PTR(dx);
  if (B <= 0 || OH <= 0 || OW <= 0 || Np <= 0 || C <= 0 || KH <= 0 || KW <= 0 || IH <= 0 || IW <= 0) return XPT_ERR_SHAPE;
  if (Np % 8 != 0 || gpitch < Np || gpitch % 8 != 0 || dxpitch < C || ((uintptr_t)g) % 16 != 0 || ((uintptr_t)wb) % 16 != 0)
    return XPT_ERR_ARG;
  if ((stride != 1 && stride != 2) || (fold2x2 != 0 && fold2x2 != 1) || (fold2x2 && stride != 1)) return XPT_ERR_ARG;
  ConvArgs a{};
  a.x = (const unsigned short*)g; a.w = (const unsigned short*)wb; a.bias = nullptr; a.y = (unsigned short*)dx;
  a.xpitch = gpitch; a.ypitch = dxpitch;
  a.B = B; a.PH = OH; a.PW = OW;
  a.shift = stride == 2 ? 1 : 0; a.Hlim = OH * stride; a.Wlim = OW * stride;
  a.C = Np; a.N = C; a.KH = KH; a.KW = KW;
  a.so = 1; a.sgn = -1; a.off_h = pad_t; a.off_w = pad_l;
  a.OH = fold2x2 ? 2 * IH : IH; a.OW = fold2x2 ? 2 * IW : IW;
  a.xs = stride; a.quad = fold2x2; a.slope = 1.f;
  return dispatch_igemm(a, (hipStream_t)stream);
}

extern "C" int xpt_conv_pack_job_bytes(void) { return (int)sizeof(PackJob); }

extern "C" int xpt_conv_pack_weights(const void* jobs, int njobs, long long nblocks, void* stream) {
  XPT_CHECK_PTR(jobs);
  if (njobs <= 0 || nblocks <= 0 || nblocks > 0x7fffffffLL) return XPT_ERR_SHAPE;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, (const PackJob*)jobs,
                     njobs);
  return xpt_launch_status();
}

extern "C" int xpt_restack_bf16(const float* image5d, void* out, int B, int S, int H, int W, int Cp, void* stream) {
  XPT_CHECK_PTR(image5d); XPT_CHECK_PTR(out);
  if (B <= 0 || S <= 0 || H <= 0 || W <= 0 || Cp < 3 * S) return XPT_ERR_SHAPE;
  const long long HW = (long long)H * W;
  const int slots = S + ((Cp - 3 * S + 2) / 3);
  const long long total = (long long)B * HW * slots;
  const long long blocks = (total + 255) / 256;
  if (total >= 0x7fffffffLL || HW >= (1LL << 24)) return XPT_ERR_SHAPE;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(restack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, image5d,
                     (unsigned short*)out, B, S, HW, Cp);
  return xpt_launch_status();
}
