// xpt_headconv.hip -- the depth decoder's prediction heads: keras Conv2D(1, 3, padding="same", activation linear) on a
// C-channel feature map (get_scaled_depth, model/build_model/depth_net.py:87-92 via CustomConv2D,
// model/model_util/layer_ops.py:5-36), forward and the complete backward, as bandwidth-bound VALU kernels.
//
// A single output channel makes this a per-pixel dot product over 9 C values: no matrix-core shape fits (31 of 32 MFMA
// rows would be zero) and the library's fp32 solvers for it are exactly the kind of launch (zeroing memset + atomics)
// that does not survive hipGraph replay on this stack.  The activations are read as bf16 NHWC
// (16 bytes = 8 channels per lane, LPP = C / 8 lanes per pixel), the 9 C weights stay fp32 (the master copy is read
// directly: a [1, C, 3, 3] channels_last kernel is [kh][kw][C] in memory) and the prediction is produced in fp32 -- it
// feeds the depth that the warp kernels consume unquantised.
//   forward : pre[p]        = bias + sum_{tap, c} x[p + off(tap)][c] w[tap][c]
//   backward: dx[p][c]      = sum_tap g[p - off(tap)] w[tap][c]                (bf16, every channel written once)
//             dW[tap][c]    = sum_p g[p] x[p + off(tap)][c],  dbias = sum_p g[p]
//   as per-workgroup partials [blocks][9 C + 1] for the step's finishing launch (xpt_reduce_partials); fixed order.
#include "xpt_common.h"

namespace {

__device__ inline float bf_lo(unsigned v) { return xpt_h2f_lo(v); }
__device__ inline float bf_hi(unsigned v) { return xpt_h2f_hi(v); }
__device__ inline unsigned pack_bf2(float a, float b) {
  return (unsigned)xpt_f2h(a) | ((unsigned)xpt_f2h(b) << 16);
}
__device__ inline void unpack8(const uint4& v, float (&f)[8]) {
  f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
  f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}

// LPP lanes share one pixel (8 channels each); a wave covers 64 / LPP consecutive pixels
template <int LPP>
__global__ __launch_bounds__(256) void head_fwd_kernel(const unsigned short* __restrict__ x, long long xpitch,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ pre, int B, int H, int W) {
  constexpr int C = 8 * LPP, PPW = 64 / LPP;
  const int lane = threadIdx.x & 63, sub = lane % LPP, pl = lane / LPP;
  const long long P = (long long)B * H * W;
  float wt[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) wt[t][e] = w[t * C + 8 * sub + e];
  const float b0 = bias ? bias[0] : 0.f;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  for (long long p0 = wave * PPW; p0 < P; p0 += nwaves * PPW) {
    const long long p = p0 + pl;
    const bool live = p < P;
    const long long pc = live ? p : P - 1;
    unsigned colu, rowu;                          // (P < 2^31: the launcher checks; a 64-bit % is ~120 instructions on this ISA)
    xpt_divmod(xpt_divmod((unsigned)pc, (unsigned)W, colu), (unsigned)H, rowu);
    const int col = (int)colu, row = (int)rowu;
    float acc = 0.f;
    uint4 v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {                 // unconditional loads from a clamped pixel, zeroed by select
      const int dr = t / 3 - 1, dc = t % 3 - 1;
      const bool ok = row + dr >= 0 && row + dr < H && col + dc >= 0 && col + dc < W;
      const long long q = ok ? pc + (long long)dr * W + dc : pc;
      const uint4 ld = *(const uint4*)(x + q * xpitch + 8 * sub);
      v[t] = ok ? ld : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float f[8];
      unpack8(v[t], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc = fmaf(f[e], wt[t][e], acc);
    }
#pragma unroll
    for (int o = 1; o < LPP; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (live && sub == 0) pre[p] = acc + b0;
  }
}

template <int LPP>
__global__ __launch_bounds__(256) void head_bwd_kernel(const unsigned short* __restrict__ x, long long xpitch,
                                                        const float* __restrict__ w, const float* __restrict__ g,
                                                        const unsigned short* __restrict__ addend, long long apitch,
                                                        unsigned short* __restrict__ dx, float* __restrict__ partials,
                                                        int B, int H, int W) {
  constexpr int C = 8 * LPP, PPW = 64 / LPP;
  __shared__ float red[4][9 * C + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, sub = lane % LPP, pl = lane / LPP;
  const long long P = (long long)B * H * W;
  float wt[9][8], dw[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      wt[t][e] = w[t * C + 8 * sub + e];
      dw[t][e] = 0.f;
    }
  float db = 0.f;
  const long long wave = (long long)blockIdx.x * 4 + wv, nwaves = (long long)gridDim.x * 4;
  for (long long p0 = wave * PPW; p0 < P; p0 += nwaves * PPW) {
    const long long p = p0 + pl;
    const bool live = p < P;
    const long long pc = live ? p : P - 1;
    unsigned colu, rowu;                          // (P < 2^31: the launcher checks; a 64-bit % is ~120 instructions on this ISA)
    xpt_divmod(xpt_divmod((unsigned)pc, (unsigned)W, colu), (unsigned)H, rowu);
    const int col = (int)colu, row = (int)rowu;
    const float gp = live ? g[pc] : 0.f;
    if (sub == 0) db += gp;
    // Both gradients from the MIRRORED neighbours of g (one channel: 4-byte loads that hit the cache) and ONE 16-byte load
    // of this pixel's features:  dx[p][c] = sum_tap g[p - off(tap)] w[tap][c]  and, re-indexed over the input pixel,
    // dW[tap][c] = sum_q g[q - off(tap)] x[q][c]  (round 4; the first version read x[p + off(tap)] for all nine taps:
    // nine 16-byte loads per lane and pixel, 24 us at full resolution).
    float gn[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dr = t / 3 - 1, dc = t % 3 - 1;
      const bool ok2 = live && row - dr >= 0 && row - dr < H && col - dc >= 0 && col - dc < W;
      const long long q2 = ok2 ? pc - (long long)dr * W - dc : pc;
      const float gl = g[q2];
      gn[t] = ok2 ? gl : 0.f;
    }
    float fc[8];
    unpack8(*(const uint4*)(x + pc * xpitch + 8 * sub), fc);
    float d[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        dw[t][e] = fmaf(gn[t], fc[e], dw[t][e]);
        d[e] = fmaf(gn[t], wt[t][e], d[e]);
      }
    }
    if (live) {
      if (addend != nullptr) {                     // the gradient the feature map's other consumer left (fan-in folded in)
        float fa[8];
        unpack8(*(const uint4*)(addend + p * apitch + 8 * sub), fa);
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] += fa[e];
      }
      uint4 o;
      o.x = pack_bf2(d[0], d[1]); o.y = pack_bf2(d[2], d[3]); o.z = pack_bf2(d[4], d[5]); o.w = pack_bf2(d[6], d[7]);
      *(uint4*)(dx + p * C + 8 * sub) = o;
    }
  }
  // wave sum over its PPW pixel slots (lanes with equal `sub`), then the 4 waves through LDS, in wave order
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int o = LPP; o < 64; o <<= 1) dw[t][e] += __shfl_xor(dw[t][e], o, 64);
#pragma unroll
  for (int o = LPP; o < 64; o <<= 1) db += __shfl_xor(db, o, 64);
  if (pl == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wv][t * C + 8 * sub + e] = dw[t][e];
    if (sub == 0) red[wv][9 * C] = db;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 9 * C + 1; i += 256)
    partials[(long long)blockIdx.x * (9 * C + 1) + i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
}

int g_head_passes = 4, g_head_max_blocks = 512;   // in-step sweep (passes, blocks): 8,256 6.726 ms; 4,512 6.706; 2,1024 6.723; 1,2048 6.744

int head_blocks(long long P, int C) {
  const long long ppb = 4 * (64 / (C / 8));            // pixels per workgroup pass
  long long blocks = (P + ppb * g_head_passes - 1) / (ppb * g_head_passes);    // passes per workgroup
  if (blocks > g_head_max_blocks) blocks = g_head_max_blocks;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace

extern "C" int xpt_headconv_tune(int passes, int max_blocks) {
  if (passes < 1 || max_blocks < 1) return XPT_ERR_ARG;
  g_head_passes = passes;
  g_head_max_blocks = max_blocks;
  return XPT_OK;
}

extern "C" int xpt_headconv_bwd_blocks(int B, int H, int W, int C) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
  return head_blocks((long long)B * H * W, C);
}

/* pre [B,H,W] fp32 = bias + conv3x3_same(x [B,H,W,C] bf16 (pixel pitch xpitch), w [3][3][C] fp32); C in {16,32,64,128} */
extern "C" int xpt_headconv_fwd(const void* x, long long xpitch, const float* w, const float* bias, float* pre, int B, int H,
                                int W, int C, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(w); XPT_CHECK_PTR(pre);
  if (B <= 0 || H <= 0 || W <= 0 || (long long)B * H * W >= 0x7fffffffLL) return XPT_ERR_SHAPE;
  if ((C != 16 && C != 32 && C != 64 && C != 128) || xpitch < C || xpitch % 8 != 0 || ((uintptr_t)x) % 16 != 0) return XPT_ERR_ARG;
  const long long P = (long long)B * H * W;
  const long long ppb = 4 * (64 / (C / 8));
  long long blocks = (P + ppb - 1) / ppb;
  if (blocks > 2048) blocks = 2048;
  hipStream_t s = (hipStream_t)stream;
  const unsigned short* xp = (const unsigned short*)x;
  XPT_BEGIN_LAUNCH();
  switch (C / 8) {
    case 2: hipLaunchKernelGGL(head_fwd_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, xp, xpitch, w, bias, pre, B, H, W); break;
    case 4: hipLaunchKernelGGL(head_fwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, xp, xpitch, w, bias, pre, B, H, W); break;
    case 8: hipLaunchKernelGGL(head_fwd_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, s, xp, xpitch, w, bias, pre, B, H, W); break;
    default: hipLaunchKernelGGL(head_fwd_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, xp, xpitch, w, bias, pre, B, H, W); break;
  }
  return xpt_launch_status();
}

extern "C" int xpt_headconv_bwd_add(const void* x, long long xpitch, const float* w, const float* g, const void* addend,
                                    long long addend_pitch, void* dx, float* partials, size_t partial_floats, int B, int H, int W,
                                    int C, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(w); XPT_CHECK_PTR(g); XPT_CHECK_PTR(dx); XPT_CHECK_PTR(partials);
  if (B <= 0 || H <= 0 || W <= 0 || (long long)B * H * W >= 0x7fffffffLL) return XPT_ERR_SHAPE;
  if ((C != 16 && C != 32 && C != 64 && C != 128) || xpitch < C || xpitch % 8 != 0 || ((uintptr_t)x) % 16 != 0 ||
      ((uintptr_t)dx) % 16 != 0)
    return XPT_ERR_ARG;
  if (addend != nullptr && (addend_pitch < C || addend_pitch % 8 != 0 || ((uintptr_t)addend) % 16 != 0)) return XPT_ERR_ARG;
  const int blocks = head_blocks((long long)B * H * W, C);
  if (partial_floats < (size_t)blocks * (9 * C + 1)) return XPT_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const unsigned short* xp = (const unsigned short*)x;
  const unsigned short* ap = (const unsigned short*)addend;
  unsigned short* dxp = (unsigned short*)dx;
  XPT_BEGIN_LAUNCH();
  switch (C / 8) {
    case 2: hipLaunchKernelGGL(head_bwd_kernel<2>, dim3(blocks), dim3(256), 0, s, xp, xpitch, w, g, ap, addend_pitch, dxp, partials, B, H, W); break;
    case 4: hipLaunchKernelGGL(head_bwd_kernel<4>, dim3(blocks), dim3(256), 0, s, xp, xpitch, w, g, ap, addend_pitch, dxp, partials, B, H, W); break;
    case 8: hipLaunchKernelGGL(head_bwd_kernel<8>, dim3(blocks), dim3(256), 0, s, xp, xpitch, w, g, ap, addend_pitch, dxp, partials, B, H, W); break;
    default: hipLaunchKernelGGL(head_bwd_kernel<16>, dim3(blocks), dim3(256), 0, s, xp, xpitch, w, g, ap, addend_pitch, dxp, partials, B, H, W); break;
  }
  return xpt_launch_status();
}

/* g [B,H,W] fp32 -> dx [B,H,W,C] bf16 dense, partials [xpt_headconv_bwd_blocks()][9 C + 1] fp32
 * (row = dW [3][3][C] followed by dbias) */
extern "C" int xpt_headconv_bwd(const void* x, long long xpitch, const float* w, const float* g, void* dx, float* partials,
                                size_t partial_floats, int B, int H, int W, int C, void* stream) {
  return xpt_headconv_bwd_add(x, xpitch, w, g, nullptr, 0, dx, partials, partial_floats, B, H, W, C, stream);
}
