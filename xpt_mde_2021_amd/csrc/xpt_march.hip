// xpt_march.hip -- second generation of the fused view synthesis + L1 + SSIM march kernels (gfx950).
//
// Same contract as xpt_fused.hip (one pass per pyramid scale, no synthesized image materialised):
//   SynthesizeSingleScale.synthesize_batch_view   model/synthesize/synthesize_base.py:88-178
//   BilinearInterpolation.__call__                model/synthesize/bilinear_interp.py:7-147
//   photometric_loss_l1 / photometric_loss_ssim   model/loss_and_metric/loss_util.py:6-25, 52-96
// and the same mapping (one wave = a 62-column strip of one source view marching down a chunk of rows, 3x3 SSIM
// window = DPP horizontal sums + a three-row register rotation).  What changed, and why (tools/lab/valu_rates.hip,
// tools/lab/gather_rates.hip, profiles/r03_lab_*.txt):
//   * gfx950 issues v_add / v_mul / v_fma / v_mov / v_and / v_add_u32 at 2 cycles per wave-instruction, but DPP
//     operations, v_cmp, v_cndmask, v_floor, v_cvt, v_min / v_max / v_med3, shifts and integer multiply-adds only
//     every ~3.2-4.4 cycles back to back, v_pk_*_f32 at ~3.4-4 (no gain over two scalar operations) and v_rcp_f32 at
//     6-8: the row body is written for the fast class -- masks are multiplied in, clamps are output modifiers, the
//     SSIM quotient is evaluated on window SUMS scaled by the window size (no per-term division by the pixel count),
//     floor / fraction come from v_cvt_flr_i32_f32 / v_fract_f32, no packed arithmetic;
//   * bilinear taps are fetched with a scalar image base + 32-bit lane offsets (no 64-bit address arithmetic);
//   * workgroups are renumbered so that each XCD (own L2) works on a contiguous range of chunks / strips.
#include "xpt_common.h"

using namespace xpt;

#define SSIM_C1 (0.01f * 0.01f)
#define SSIM_C2 (0.03f * 0.03f)
#ifndef MARCH_DBG
#define MARCH_DBG 0       // lab diagnostics (tools/lab): 1 no tap gathers, 2 no SSIM arithmetic, 3 no loads at all, 4 taps but no arithmetic on them
#endif
#define MSTRIP 62         // forward: output columns per wave (one halo lane on each side)
#define MSTRIP_B 60       // backward: two halo lanes on each side

namespace {

__device__ inline float wave_shr1(float v) {   // lane i <- lane i-1 (0 into lane 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ inline float wave_shl1(float v) {   // lane i <- lane i+1 (0 into lane 63)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ inline float hsum3(float v) { return (wave_shr1(v) + v) + wave_shl1(v); }
__device__ inline float rcpf(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline int cvt_floor(float x) {     // (int) floor(x) in one instruction (saturating; NaN -> 0)
  int i;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(i) : "v"(x));
  return i;
}
__device__ inline float wave_sum_all(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct MDims {
  int B, N, h, w, S, CH, R;   // S strips per row, CH row chunks, R rows per chunk
  float scale;
};

struct MArgs {
  const float* src[4];
  const float* depth[4];
  const float* target[4];
  const float* g_l1[4];
  const float* g_ss[4];
  float* ddepth[4];
  long long part_off[4];       // offset (floats) of the scale's per-wave partials in the workspace
  MDims d[4];
  float inv_count[4];
  unsigned block_off[5];       // first workgroup of every scale; unused scales = the total
  int waves_per_b[4];
  int prio[4];                 // s_setprio level of the scale's waves (backward launch: longest waves first)
};

__device__ inline int scale_of(const MArgs& m, unsigned b) {
  return (int)(b >= m.block_off[1]) + (int)(b >= m.block_off[2]) + (int)(b >= m.block_off[3]);
}

// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2): give every XCD a CONTIGUOUS range of logical
// workgroups, so that neighbouring chunks / strips -- which share halo rows and the lines their taps straddle -- meet
// in one L2.  A bijection of [0, n): speed only, never correctness.
__device__ inline unsigned xcd_contiguous(unsigned bid, unsigned n) {
  const unsigned x = bid & 7u, q = n >> 3, rem = n & 7u;
  return x * q + (x < rem ? x : rem) + (bid >> 3);
}

struct WaveJob {
  int b, n, s, ck;
  bool valid;
};

__device__ inline WaveJob wave_job(const MDims& d, unsigned block) {
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long gw = (long long)block * 4 + wid;
  const long long nwaves = (long long)d.B * d.S * d.CH * d.N;
  WaveJob j;
  j.valid = gw < nwaves;
  j.n = (int)(gw % d.N);
  long long r = gw / d.N;
  j.ck = (int)(r % d.CH); r /= d.CH;
  j.s = (int)(r % d.S);
  j.b = (int)(r / d.S);
  return j;
}

// K (R (d Kinv (col, r, 1)) + t) = d (M (col, r, 1)) + K t with M = K R Kinv (wave-uniform, folded once per wave:
// the reference's chain pixel2cam / transform / cam2pixel, synthesize_base.py:106-178, re-associated)
struct Fold {
  float M[9], kt[3];
};
__device__ inline Fold fold_camera(const Cam& cam, const Pose& pose) {
  Fold f;
  float KR[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      KR[3 * i + j] = cam.k[3 * i] * pose.r[j] + cam.k[3 * i + 1] * pose.r[3 + j] + cam.k[3 * i + 2] * pose.r[6 + j];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
      f.M[3 * i + j] = KR[3 * i] * cam.ki[j] + KR[3 * i + 1] * cam.ki[3 + j] + KR[3 * i + 2] * cam.ki[6 + j];
    f.kt[i] = cam.k[3 * i] * pose.t[0] + cam.k[3 * i + 1] * pose.t[1] + cam.k[3 * i + 2] * pose.t[2];
  }
  return f;
}

// 12 window sums per pixel [x(3) y(3) x^2+y^2 (3) xy(3)]: the forward needs sigma_x + sigma_y only as their sum.
constexpr int NW = 12;

// Per-lane constants of the SSIM quotient on window SUMS (c = number of pixels in the window, 9 / 6 / 4):
//   ssim = (2 Sx Sy + C1 c^2) (2 c Sxy - 2 Sx Sy + C2 c^2) / ((Sx^2 + Sy^2 + C1 c^2) (c Sq - (Sx^2 + Sy^2) + C2 c^2))
// (numerator and denominator of loss_util.py:80-93 multiplied by c^4: no division of the twelve sums by c)
struct SsimK {
  float c, c2x, C1cc, C2cc;     // c, 2 c, C1 c^2, C2 c^2
};
// cnt_c = window columns inside the image (per lane, 2 or 3), cnt_r = window rows inside it (wave-uniform, 2 or 3): exact
__device__ inline SsimK ssim_consts(float cnt_c, float cnt_r) {
  SsimK k;
  k.c = cnt_c * cnt_r;
  k.c2x = k.c + k.c;
  const float cc = k.c * k.c;
  k.C1cc = SSIM_C1 * cc;
  k.C2cc = SSIM_C2 * cc;
  return k;
}
__device__ inline float window_rows(int r, int h) { return (float)((r > 0 ? 1 : 0) + 1 + (r < h - 1 ? 1 : 0)); }   // wave-uniform

// loss value clamp((1 - ssim) / 2, 0, 1) of one channel: 12 two-cycle operations + one reciprocal
__device__ inline float ssim_loss_sums(float Sx, float Sy, float Sq, float Sxy, const SsimK& k) {
  const float a = Sx * Sy;
  const float b = Sx * Sx + Sy * Sy;
  const float n1h = -0.5f * k.C1cc - a;                             // -(2 a + C1 c^2) / 2
  const float n2 = (Sxy * k.c2x + k.C2cc) - 2.f * a;
  const float d1 = b + k.C1cc;
  const float d2 = (Sq * k.c + k.C2cc) - b;
  const float l = (n1h * n2) * rcpf(d1 * d2) + 0.5f;
  return __builtin_fminf(__builtin_fmaxf(l, 0.f), 1.f);            // (the compiler folds this into the fma's clamp modifier)
}

// ------------------------------------------------------------------------------------------------ forward
// part[16 * wave + {0,1}] = (sum of L1 terms, sum of SSIM terms) over the wave's output pixels (3 channels each).
__device__ __forceinline__ void march_fwd_body(const float* __restrict__ src, const float* __restrict__ depth,
                                               const float* __restrict__ T, const float* __restrict__ K,
                                               const float* __restrict__ target, float* __restrict__ part,
                                               const MDims& d, unsigned block) {
  const WaveJob job = wave_job(d, block);
  if (!job.valid) return;                          // no block-level synchronisation in this kernel
  const int lane = threadIdx.x & 63;
  const int W = __builtin_amdgcn_readfirstlane(d.w), H = __builtin_amdgcn_readfirstlane(d.h);   // (kept in SGPRs: not re-loaded per row)
  const int P = H * W;
  const int col = job.s * MSTRIP - 1 + lane;
  const bool col_in = (col >= 0) && (col < W);
  const bool out_lane = (lane >= 1) && (lane <= MSTRIP) && col_in;
  const int col_c = min(max(col, 0), W - 1);
  const int r0 = job.ck * d.R, r1 = min(r0 + d.R, H);
  const Fold f = fold_camera(load_cam(K + 9 * job.b, d.scale), load_pose(T + 16 * (job.b * d.N + job.n)));
  const float* simg = src + (long long)(job.b * d.N + job.n) * P * 3;      // wave-uniform bases (SGPR pairs)
  const float* dimg = depth + (long long)job.b * P;
  const float* timg = target + (long long)job.b * P * 3;
  float m_c[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) m_c[i] = f.M[3 * i] * (float)col + f.M[3 * i + 2];
  const float colm = col_in ? 1.f : 0.f;                                   // zero padding of the SSIM window: x, y = 0 there
  const float cnt_c = (float)((col > 0 ? 1 : 0) + 1 + (col < W - 1 ? 1 : 0));
  const float u_hi = (float)(W - 1), v_hi = (float)(H - 1);           // valid: 0 <= u' < w - 1, 0 <= v' < h - 1 (floor + 1 in range)
  const unsigned row_b = 12u * (unsigned)W;                              // bytes per source row

  float acc_l1 = 0.f, acc_ss = 0.f;
  // window rows: hpair = (row r-2) + (row r-1) and hprev = row r-1 stay in registers; with the row just synthesized they
  // give the window sum (hpair + hcur, the same association as (m2 + m1) + cur) and the next step's pair
  float hA[NW], hB[NW], hC[NW], hpair[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) { hA[i] = 0.f; hB[i] = 0.f; hC[i] = 0.f; hpair[i] = 0.f; }
  bool black_prev = true;

  float nd;          // depth and target pixel of the row about to be processed (loads issued one row ahead)
  f32x3 nx;
  auto prefetch = [&](int rr) {
    const unsigned p = (unsigned)(min(max(rr, 0), H - 1) * W + col_c);
    const unsigned od = p * 4u, ot = p * 12u;
#if MARCH_DBG == 3
    nd = 10.f + (float)od * 1e-6f; nx = f32x3{0.1f, 0.2f, (float)ot * 1e-7f};
#else
    asm volatile("global_load_dword %0, %1, %2" : "=v"(nd) : "v"(od), "s"(dimg) : "memory");
    asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(nx) : "v"(ot), "s"(timg) : "memory");
#endif
  };

  // body(r, cur, m1): row r's pixel (prefetched), its taps, L1; window sums of row r; SSIM of centre row r-1
  auto body = [&](int r, float (&cur)[NW], const float (&m1)[NW]) {
    const bool row_in = (r >= 0) && (r < H);                             // wave-uniform
    const float rm = row_in ? colm : 0.f;
    const float dd = nd;
    const float x0 = nx.x * rm, x1 = nx.y * rm, x2 = nx.z * rm;
    const float fr = (float)r;
    const float q0 = (f.M[1] * fr + m_c[0]) * dd + f.kt[0];
    const float q1 = (f.M[4] * fr + m_c[1]) * dd + f.kt[1];
    const float q2 = (f.M[7] * fr + m_c[2]) * dd + f.kt[2];
    const float zinv = rcpf(q2 + 1e-10f);
    const float up = q0 * zinv, vp = q1 * zinv;
    // BilinearInterpolation (bilinear_interp.py:34-102): the clipped neighbours satisfy uf + 1 == uc exactly when
    // 0 <= floor(u') <= w - 2, i.e. 0 <= u' < w - 1 (NaN compares false: invalid, as in xpt_fused.hip)
    const bool ok = row_in && col_in && (up >= 0.f) && (up < u_hi) && (vp >= 0.f) && (vp < v_hi) && (dd != 0.f);
    const int iu = cvt_floor(up), iv = cvt_floor(vp);
    const float wuc = __builtin_amdgcn_fractf(up), wvc = __builtin_amdgcn_fractf(vp);   // u' - floor(u') (exact for u' >= 0)
    const float wuf = 1.f - wuc, wvf = 1.f - wvc;                                       // (floor + 1) - u'   (exact as well)
    const float wff = wuf * wvf, wfc = wuf * wvc, wcf = wuc * wvf, wcc = wuc * wvc;
    const unsigned off0 = ok ? (unsigned)iv * row_b + (unsigned)iu * 12u : 0u;
    const unsigned off1 = off0 + row_b;
    f32x4 a0, a1;
    f32x2 b0, b1;
#if MARCH_DBG == 1 || MARCH_DBG == 3
    a0 = f32x4{x0, x1, x2, wff}; a1 = f32x4{wfc, x1, x0, x2}; b0 = f32x2{(float)off0, x1}; b1 = f32x2{(float)off1, x2};
    prefetch(r + 1);
#else
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(a0) : "v"(off0), "s"(simg) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:16" : "=v"(b0) : "v"(off0), "s"(simg) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(a1) : "v"(off1), "s"(simg) : "memory");
    asm volatile("global_load_dwordx2 %0, %1, %2 offset:16" : "=v"(b1) : "v"(off1), "s"(simg) : "memory");
    prefetch(r + 1);                               // behind the gathers
    asm volatile("s_waitcnt vmcnt(2)" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1) : : "memory");   // gathers landed
#endif
    const float v0 = ((a0.x * wff + a1.x * wfc) + a0.w * wcf) + a1.w * wcc;
    const float v1 = ((a0.y * wff + a1.y * wfc) + b0.x * wcf) + b1.x * wcc;
    const float v2 = ((a0.z * wff + a1.z * wfc) + b0.y * wcf) + b1.y * wcc;
    const float y0 = ok ? v0 : 0.f, y1 = ok ? v1 : 0.f, y2 = ok ? v2 : 0.f;
    const bool black = ((y0 + y1) + y2) == 0.f;                            // mean_c == 0  <=>  sum_c == 0
    const float l1 = (fabsf(y0 - x0) + fabsf(y1 - x1)) + fabsf(y2 - x2);
    const bool in_chunk = (r >= r0) && (r < r1);                           // wave-uniform
    acc_l1 += (out_lane && in_chunk && !black) ? l1 : 0.f;
#if MARCH_DBG == 2 || MARCH_DBG == 4
    acc_ss += (y0 + y1) * y2 + x0 * x1 + x2 + dd + m1[0];
    black_prev = black;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(nd), "+v"(nx) : : "memory");
    return;
#endif
    const float p[NW] = {x0, x1, x2, y0, y1, y2, x0 * x0 + y0 * y0, x1 * x1 + y1 * y1, x2 * x2 + y2 * y2,
                         x0 * y0, x1 * y1, x2 * y2};
#pragma unroll
    for (int i = 0; i < NW; ++i) cur[i] = hsum3(p[i]);
    const int rc = r - 1;                          // centre row of the window (m2, m1, cur)
    if (rc >= r0 && rc < r1) {
      float W[NW];
#pragma unroll
      for (int i = 0; i < NW; ++i) W[i] = hpair[i] + cur[i];
      // window rows inside the image: 2 on the first / last image row (wave-uniform)
      const SsimK k = ssim_consts(cnt_c, window_rows(rc, H));
      const float sum = (ssim_loss_sums(W[0], W[3], W[6], W[9], k) + ssim_loss_sums(W[1], W[4], W[7], W[10], k)) +
                        ssim_loss_sums(W[2], W[5], W[8], W[11], k);
      acc_ss += (out_lane && !black_prev) ? sum : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) hpair[i] = m1[i] + cur[i];
    black_prev = black;
    // the next row's depth / target: waited for before leaving the body, so that no register with a load in flight
    // crosses a loop edge or a register copy
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(nd), "+v"(nx) : : "memory");
  };

  int r = r0 - 1;
  const int rend = r1;          // inclusive: one halo row above and below the chunk
  prefetch(r);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(nd), "+v"(nx) : : "memory");
  while (r <= rend) {           // (three names, two of them live at a time: the rotation costs no register moves)
    body(r, hA, hC); ++r;
    if (r > rend) break;
    body(r, hB, hA); ++r;
    if (r > rend) break;
    body(r, hC, hB); ++r;
  }
  acc_l1 = wave_sum_all(acc_l1);
  acc_ss = wave_sum_all(acc_ss);
  if (lane == 0) {
    const long long gw = ((long long)(job.b * d.S + job.s) * d.CH + job.ck) * d.N + job.n;
    part[16 * gw] = acc_l1;
    part[16 * gw + 1] = acc_ss;
  }
}

__global__ __launch_bounds__(256) void march_fwd_ms_kernel(MArgs m, const float* __restrict__ T, const float* __restrict__ K,
                                                           float* __restrict__ part) {
  const int s = scale_of(m, blockIdx.x);
  const unsigned local = blockIdx.x - m.block_off[s], n = m.block_off[s + 1] - m.block_off[s];
  march_fwd_body(m.src[s], m.depth[s], T, K, m.target[s], part + m.part_off[s], m.d[s], xcd_contiguous(local, n));
}

// losses[s * B + b] = L1, losses[(nscales + s) * B + b] = SSIM: fixed-order sums of the per-wave partials
__global__ void march_reduce_ms_kernel(MArgs m, const float* __restrict__ part, float* __restrict__ losses) {
  const int b = blockIdx.x, s = blockIdx.y, t = threadIdx.x, B = gridDim.x, nscales = gridDim.y;
  const int waves_per_b = m.waves_per_b[s];
  const float* q = part + m.part_off[s] + (long long)b * waves_per_b * 16;
  float s0 = 0.f, s1 = 0.f;
  for (int k = t; k < waves_per_b; k += 64) { s0 += q[16 * k]; s1 += q[16 * k + 1]; }
  s0 = wave_sum_all(s0);
  s1 = wave_sum_all(s1);
  if (t == 0) {
    losses[s * B + b] = s0 * m.inv_count[s];
    losses[(nscales + s) * B + b] = s1 * m.inv_count[s];
  }
}

// ------------------------------------------------------------------------------------------------ backward (+ losses)
// d loss / d y(q,c) = g_l1 sign(y-x) [not black(q)]  +  sum_{p in win(q)} ( A_pc + B2_pc y_qc + C_pc x_qc )
// with, for window centre p and gg = -(1/2) g_ssim [not black(p)] [|ssim| <= 1]   (loss_util.py:52-96; the clip gradient
// of tf.clip_by_value passes inside the closed interval):
//   A = gg d ssim / d Sy,  B2 = 2 gg d ssim / d Sq,  C = gg d ssim / d Sxy      (Sy, Sq = S(x^2 + y^2), Sxy: window SUMS)
// The second 3x3 box sum reuses the DPP + sliding-row scheme one row later: three stages per row step,
//   A  synthesize row r (taps, y and its derivatives w.r.t. the sampling position, horizontal window sums),
//   B  coefficients of centre row r-1 (and its SSIM loss value), their horizontal sums,
//   C  gradient of row r-2: d_depth (exchanged among the four views of a strip through LDS, ONE barrier per workgroup at
//      the end of the chunk) and the pose gradient, accumulated as G = sum dq (x) (d col, d row, d, 1) in pixel
//      coordinates; K^T G Kinv^T is applied once per wave (the per-pixel chain through K and R costs 21 more operations).
// The same launch leaves the L1 / SSIM sums of its output pixels in the workspace: with the upstream gradients known in
// advance (they are loss weights / batch size), forward and backward of a training step are this ONE pass.
constexpr int NC = 9;

struct RowSt {            // what stage C needs of a row, two row steps later
  float x[3], y[3], gu[3], gv[3];
  float d, zinv, up, vp, nb;
};

// SSIM of one channel from window sums: loss value, and the gradient coefficients of the window's centre pixel
__device__ inline float ssim_coeffs_sums(float Sx, float Sy, float Sq, float Sxy, const SsimK& k, float gg, float& A, float& B2,
                                         float& C) {
  const float a = Sx * Sy;
  const float b = Sx * Sx + Sy * Sy;
  const float n1 = (a + a) + k.C1cc;
  const float n2 = (Sxy * k.c2x + k.C2cc) - (a + a);
  const float d1 = b + k.C1cc;
  const float d2 = (Sq * k.c + k.C2cc) - b;
  const float inv = rcpf(d1 * d2);
  const float ssim = (n1 * n2) * inv;
  const float gi = (fabsf(ssim) <= 1.f ? gg : 0.f) * inv;              // clip_by_value gradient
  A = (gi + gi) * (Sx * (n2 - n1) - (Sy * ssim) * (d2 - d1));
  B2 = (ssim * d1) * (gi * (-k.c2x));
  C = n1 * (gi * k.c2x);
  const float l = 0.5f - 0.5f * ssim;
  return __builtin_fminf(__builtin_fmaxf(l, 0.f), 1.f);
}

// MODE 0: N == 4, the 4 waves of a workgroup hold the 4 views of one strip -> d_depth summed through LDS;
// MODE 1: N == 1 -> direct store.
// The state of the two rows in flight between stage A and stage C (x, y, their derivatives, the projection: 16 floats per
// lane and row) lives in a two-slot LDS ring per wave (row r overwrites row r-2 after stage C has read it): 32 registers
// less than the register rotation, i.e. 3 instead of 2 waves per SIMD -- the backward is bound by its dependent chains
// (every wave issues one vector instruction per ~5 cycles at best), not by LDS traffic (33 b32 operations per row).
constexpr int RING_FLOATS = 2 * 16 * 64;        // per wave
constexpr int DD_ROWS = 16;                     // chunk rows whose d_depth contributions wait in LDS (1 KiB per row and workgroup)
template <int MODE>
__device__ __forceinline__ void march_bwd_body(const float* __restrict__ src, const float* __restrict__ depth,
                                               const float* __restrict__ T, const float* __restrict__ K,
                                               const float* __restrict__ target, const float* __restrict__ g_l1,
                                               const float* __restrict__ g_ssim, float* __restrict__ ddepth,
                                               float* __restrict__ part, const MDims& d, float inv_count, unsigned block,
                                               float* __restrict__ lds) {
  const WaveJob job = wave_job(d, block);
  if (!job.valid) return;                   // MODE 0: nwaves % 4 == 0, whole workgroups leave together
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int W = __builtin_amdgcn_readfirstlane(d.w), H = __builtin_amdgcn_readfirstlane(d.h);   // (kept in SGPRs: not re-loaded per row)
  float* ring = lds + wid * RING_FLOATS + lane;           // ring[(slot * 16 + k) * 64]
  float* lds_dd = lds + 4 * RING_FLOATS;
  const int P = H * W;
  const int col = job.s * MSTRIP_B - 2 + lane;
  const bool col_in = (col >= 0) && (col < W);
  const bool out_lane = (lane >= 2) && (lane < 2 + MSTRIP_B) && col_in;
  const int col_c = min(max(col, 0), W - 1);
  const int r0 = job.ck * d.R, r1 = min(r0 + d.R, H);
  const Cam cam = load_cam(K + 9 * job.b, d.scale);
  const Fold f = fold_camera(cam, load_pose(T + 16 * (job.b * d.N + job.n)));
  const float* simg = src + (long long)(job.b * d.N + job.n) * P * 3;
  const float* dimg = depth + (long long)job.b * P;
  const float* timg = target + (long long)job.b * P * 3;
  float* gimg = ddepth + (long long)job.b * P;
  const float colf = (float)col;
  float m_c[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) m_c[i] = f.M[3 * i] * colf + f.M[3 * i + 2];
  const float colm = col_in ? 1.f : 0.f;
  const float outm = out_lane ? 1.f : 0.f;
  const float cnt_c = (float)((col > 0 ? 1 : 0) + 1 + (col < W - 1 ? 1 : 0));
  const float u_hi = (float)(W - 1), v_hi = (float)(H - 1);
  const unsigned row_b = 12u * (unsigned)W;
  const float gl1 = g_l1[job.b] * inv_count;
  const float gss_h = g_ssim[job.b] * inv_count * (-0.5f);

  float G[12];                              // sum of dq (x) (d col, d row, d, 1)
#pragma unroll
  for (int i = 0; i < 12; ++i) G[i] = 0.f;
  float acc_l1 = 0.f, acc_ss = 0.f;
  float hA[NW], hB[NW], hC[NW], hpair[NW], cA[NC], cB[NC], cC[NC], cpair[NC];     // pair = (row - 2) + (row - 1), as in the forward
#pragma unroll
  for (int i = 0; i < NW; ++i) { hA[i] = 0.f; hB[i] = 0.f; hC[i] = 0.f; hpair[i] = 0.f; }
#pragma unroll
  for (int i = 0; i < NC; ++i) { cA[i] = 0.f; cB[i] = 0.f; cC[i] = 0.f; cpair[i] = 0.f; }
  float nb1 = 0.f, nb2 = 0.f;               // [not black] of rows r-1 and r-2
#pragma unroll
  for (int k = 0; k < 32; ++k) ring[k * 64] = 0.f;

  float nd;
  f32x3 nx;
  auto prefetch = [&](int rr) {
    const unsigned p = (unsigned)(min(max(rr, 0), H - 1) * W + col_c);
    const unsigned od = p * 4u, ot = p * 12u;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(nd) : "v"(od), "s"(dimg) : "memory");
    asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(nx) : "v"(ot), "s"(timg) : "memory");
  };

  // d_depth: the four views' rows of the last (up to) DD_ROWS chunk rows meet in LDS; every wave finishes a quarter of them
  // (views added in a fixed order); two barriers per DD_ROWS rows, none per row
  auto flush_dd = [&](int first_row, int nrows) {
    __syncthreads();
    for (int row = wid; row < nrows; row += 4) {
      const float* v = lds_dd + row * 64 + lane;
      const float sum = ((v[0] + v[DD_ROWS * 64]) + v[2 * DD_ROWS * 64]) + v[3 * DD_ROWS * 64];
      if (out_lane) gimg[(first_row + row) * W + col] = sum;
    }
    __syncthreads();
  };

  auto body = [&](int r, float (&hcur)[NW], const float (&hm1)[NW], float (&ccur)[NC], const float (&cm1)[NC]) {
    float* slot = ring + (r & 1) * (16 * 64);                              // holds row r-2 now, row r at the end of the step
    RowSt scur;
    // ---- stage A: synthesize row r; keep y and its derivatives w.r.t. the sampling position
    {
      const bool row_in = (r >= 0) && (r < H);                           // wave-uniform
      const float rm = row_in ? colm : 0.f;
      const float dd = nd;
      scur.x[0] = nx.x * rm; scur.x[1] = nx.y * rm; scur.x[2] = nx.z * rm;
      const float fr = (float)r;
      const float q0 = (f.M[1] * fr + m_c[0]) * dd + f.kt[0];
      const float q1 = (f.M[4] * fr + m_c[1]) * dd + f.kt[1];
      const float q2 = (f.M[7] * fr + m_c[2]) * dd + f.kt[2];
      const float zinv = rcpf(q2 + 1e-10f);
      const float up = q0 * zinv, vp = q1 * zinv;
      const bool ok = row_in && col_in && (up >= 0.f) && (up < u_hi) && (vp >= 0.f) && (vp < v_hi) && (dd != 0.f);
      // invalid pixels: coordinates 0 (finite weights, tap 0) and everything derived from them multiplied by 0
      const float okf = ok ? 1.f : 0.f;
      const float us = ok ? up : 0.f, vs = ok ? vp : 0.f;
      scur.d = dd; scur.zinv = ok ? zinv : 0.f; scur.up = us; scur.vp = vs;
      const int iu = cvt_floor(us), iv = cvt_floor(vs);
      const float wuc = __builtin_amdgcn_fractf(us), wvc = __builtin_amdgcn_fractf(vs);
      const unsigned off0 = (unsigned)iv * row_b + (unsigned)iu * 12u;
      const unsigned off1 = off0 + row_b;
      f32x4 a0, a1;
      f32x2 b0, b1;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(a0) : "v"(off0), "s"(simg) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, %2 offset:16" : "=v"(b0) : "v"(off0), "s"(simg) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(a1) : "v"(off1), "s"(simg) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, %2 offset:16" : "=v"(b1) : "v"(off1), "s"(simg) : "memory");
      prefetch(r + 1);
      asm volatile("s_waitcnt vmcnt(2)" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1) : : "memory");
      // taps: row vf = (ff | cf) = (a0.xyz | a0.w b0.xy), row vf + 1 = (fc | cc) = (a1.xyz | a1.w b1.xy).  Two horizontal
      // interpolations, one vertical: y = top + (bot - top) wvc; d y / d v' = bot - top; d y / d u' interpolates the two
      // horizontal differences (the reference's sum of four weighted taps re-associated, bilinear_interp.py:88-146)
      const float ff[3] = {a0.x, a0.y, a0.z}, cf[3] = {a0.w, b0.x, b0.y};
      const float fc[3] = {a1.x, a1.y, a1.z}, cc[3] = {a1.w, b1.x, b1.y};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float ca = cf[c] - ff[c], eb = cc[c] - fc[c];
        const float top = ff[c] + ca * wuc, bot = fc[c] + eb * wuc;
        const float gv = bot - top;
        scur.y[c] = (top + gv * wvc) * okf;
        scur.gv[c] = gv * okf;
        scur.gu[c] = (ca + (eb - ca) * wvc) * okf;
      }
      scur.nb = (((scur.y[0] + scur.y[1]) + scur.y[2]) == 0.f) ? 0.f : 1.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        hcur[c] = hsum3(scur.x[c]);
        hcur[3 + c] = hsum3(scur.y[c]);
        hcur[6 + c] = hsum3(scur.x[c] * scur.x[c] + scur.y[c] * scur.y[c]);
        hcur[9 + c] = hsum3(scur.x[c] * scur.y[c]);
      }
    }
    // ---- stage B: SSIM coefficients of centre row p = r-1, then their horizontal sums
    {
      const int p = r - 1;
      if (p >= 0 && p < H && p >= r0 - 1 && p <= r1) {                   // wave-uniform
        const SsimK k = ssim_consts(cnt_c, window_rows(p, H));
        const float gg = gss_h * nb1;
        float co[NC], lsum = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          lsum += ssim_coeffs_sums(hpair[c] + hcur[c], hpair[3 + c] + hcur[3 + c], hpair[6 + c] + hcur[6 + c],
                                   hpair[9 + c] + hcur[9 + c], k, gg, co[c], co[3 + c], co[6 + c]);
        if (p >= r0 && p < r1) acc_ss += lsum * (outm * nb1);
#pragma unroll
        for (int i = 0; i < NC; ++i) ccur[i] = hsum3(co[i]);
      } else {
#pragma unroll
        for (int i = 0; i < NC; ++i) ccur[i] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < NW; ++i) hpair[i] = hm1[i] + hcur[i];
    }
    // ---- stage C: gradient of pixel (q = r-2, col) (its state comes back from the ring); window rows q-1, q, q+1
    const int q = r - 2;
    if (q >= r0 && q < r1) {                                               // wave-uniform
      const float gl1n = gl1 * nb2;
      float du = 0.f, dv = 0.f, l1 = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float x = slot[c * 64], y = slot[(3 + c) * 64], gu = slot[(6 + c) * 64], gv = slot[(9 + c) * 64];
        const float SA = cpair[c] + ccur[c];
        const float SB = cpair[3 + c] + ccur[3 + c];
        const float SC = cpair[6 + c] + ccur[6 + c];
        const float df = y - x;
        // sign(df) in {-1, 0, 1}: |df| >= 1e-30 saturates the clamp (pixel differences are >= 2^-24 or exactly 0)
        const float sg = __builtin_amdgcn_fmed3f(df * 1e30f, -1.f, 1.f);
        l1 += fabsf(df);
        const float g = ((gl1n * sg + SA) + SB * y) + SC * x;
        du += g * gu;
        dv += g * gv;
      }
      const float dq_d = slot[12 * 64], zinv = slot[13 * 64], up = slot[14 * 64], vp = slot[15 * 64];
      acc_l1 += l1 * (outm * nb2);
      du *= outm; dv *= outm;                                              // halo lanes: no contribution to dT / d_depth
      const float dq0 = du * zinv, dq1 = dv * zinv;
      const float dq2 = -(du * up + dv * vp) * zinv;
      const float fq = (float)q;
      // q = d (M pix) + K t:  d q / d d = M pix
      const float ddv = (dq0 * (f.M[1] * fq + m_c[0]) + dq1 * (f.M[4] * fq + m_c[1])) + dq2 * (f.M[7] * fq + m_c[2]);
      const float P0 = dq_d * colf, P1 = dq_d * fq, P2 = dq_d;
      G[0] += dq0 * P0; G[1] += dq0 * P1; G[2] += dq0 * P2; G[3] += dq0;
      G[4] += dq1 * P0; G[5] += dq1 * P1; G[6] += dq1 * P2; G[7] += dq1;
      G[8] += dq2 * P0; G[9] += dq2 * P1; G[10] += dq2 * P2; G[11] += dq2;
      if (MODE == 0) {
        lds_dd[(wid * DD_ROWS + ((q - r0) & (DD_ROWS - 1))) * 64 + lane] = ddv;
        if (((q - r0) & (DD_ROWS - 1)) == DD_ROWS - 1 && q != r1 - 1) flush_dd(q - (DD_ROWS - 1), DD_ROWS);   // wave-uniform, the same in all four waves
      } else if (out_lane) {
        gimg[q * W + col] = ddv;
      }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) cpair[i] = cm1[i] + ccur[i];
    // row r takes the slot of row r-2 (a wave's LDS operations complete in order)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      slot[c * 64] = scur.x[c]; slot[(3 + c) * 64] = scur.y[c]; slot[(6 + c) * 64] = scur.gu[c]; slot[(9 + c) * 64] = scur.gv[c];
    }
    slot[12 * 64] = scur.d; slot[13 * 64] = scur.zinv; slot[14 * 64] = scur.up; slot[15 * 64] = scur.vp;
    nb2 = nb1; nb1 = scur.nb;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(nd), "+v"(nx) : : "memory");
  };

  int r = r0 - 2;
  const int rend = r1 + 1;      // inclusive
  prefetch(r);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(nd), "+v"(nx) : : "memory");
  while (r <= rend) {
    body(r, hA, hC, cA, cC); ++r;
    if (r > rend) break;
    body(r, hB, hA, cB, cA); ++r;
    if (r > rend) break;
    body(r, hC, hB, cC, cB); ++r;
  }
  if (MODE == 0) {
    const int done = (r1 - r0 - 1) & ~(DD_ROWS - 1);                      // rows already flushed inside the loop
    flush_dd(r0 + done, r1 - r0 - done);
  }
  // pose gradient of the wave: dRt = K^T G [Kinv^T, 0; 0, 1]  (G in pixel coordinates, X = d Kinv pix)
  float Gs[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) Gs[i] = wave_sum_all(G[i]);
  acc_l1 = wave_sum_all(acc_l1);
  acc_ss = wave_sum_all(acc_ss);
  if (lane == 0) {
    const long long gw = ((long long)(job.b * d.S + job.s) * d.CH + job.ck) * d.N + job.n;
    float KtG[12];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int bcol = 0; bcol < 4; ++bcol)
        KtG[4 * i + bcol] = (cam.k[i] * Gs[bcol] + cam.k[3 + i] * Gs[4 + bcol]) + cam.k[6 + i] * Gs[8 + bcol];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        part[16 * gw + 4 * i + j] = (KtG[4 * i] * cam.ki[3 * j] + KtG[4 * i + 1] * cam.ki[3 * j + 1]) + KtG[4 * i + 2] * cam.ki[3 * j + 2];
      part[16 * gw + 4 * i + 3] = KtG[4 * i + 3];
    }
    part[16 * gw + 12] = acc_l1;
    part[16 * gw + 13] = acc_ss;
  }
}


// ------------------------------------------------------------------------------------------------ backward, pipelined
// Round 4.  At batch 8 the one-pass launch is ~3,100 waves on 3,072 wave slots (3 per SIMD): ONE round of waves whose row
// step is a dependent chain  depth -> projection -> tap gathers -> (wait) -> 280 vector instructions;  with three waves per
// SIMD the gather latency of every row was exposed (PMC: 36 % of the wave-cycles parked on s_waitcnt, the launch at 2.4 x
// its vector-issue time).  Here the gathers of row r+1 are issued right after stage A has consumed the taps of row r --
// into the same registers -- and fly during stages B and C (two thirds of the row step); the target pixel of row r+1 rides
// with them and the depth is fetched two rows ahead.  The loads are plain C++ loads: the compiler tracks them across the
// loop edges and places the s_waitcnt in front of the first use (top of the next row step).  Same arithmetic as
// march_bwd_body, except that the pose-gradient accumulator drops its column terms: G(., d col) = col * G(., d) per lane.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x3u __attribute__((ext_vector_type(3), aligned(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

template <int MODE>
__device__ __forceinline__ void march_bwd_body_p(const float* __restrict__ src, const float* __restrict__ depth,
                                                 const float* __restrict__ T, const float* __restrict__ K,
                                                 const float* __restrict__ target, const float* __restrict__ g_l1,
                                                 const float* __restrict__ g_ssim, float* __restrict__ ddepth,
                                                 float* __restrict__ part, const MDims& d, float inv_count, unsigned block,
                                                 float* __restrict__ lds) {
  const WaveJob job = wave_job(d, block);
  if (!job.valid) return;                   // MODE 0: nwaves % 4 == 0, whole workgroups leave together
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int W = __builtin_amdgcn_readfirstlane(d.w), H = __builtin_amdgcn_readfirstlane(d.h);
  float* ring = lds + wid * RING_FLOATS + lane;           // ring[(slot * 16 + k) * 64]
  float* lds_dd = lds + 4 * RING_FLOATS;
  const int P = H * W;
  const int col = job.s * MSTRIP_B - 2 + lane;
  const bool col_in = (col >= 0) && (col < W);
  const bool out_lane = (lane >= 2) && (lane < 2 + MSTRIP_B) && col_in;
  const int col_c = min(max(col, 0), W - 1);
  const int r0 = job.ck * d.R, r1 = min(r0 + d.R, H);
  const Cam cam = load_cam(K + 9 * job.b, d.scale);
  const Fold f = fold_camera(cam, load_pose(T + 16 * (job.b * d.N + job.n)));
  const char* simg = (const char*)(src + (long long)(job.b * d.N + job.n) * P * 3);
  const float* dimg = depth + (long long)job.b * P;
  const char* timg = (const char*)(target + (long long)job.b * P * 3);
  float* gimg = ddepth + (long long)job.b * P;
  const float colf = (float)col;
  float m_c[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) m_c[i] = f.M[3 * i] * colf + f.M[3 * i + 2];
  const float colm = col_in ? 1.f : 0.f;
  const float outm = out_lane ? 1.f : 0.f;
  const float cnt_c = (float)((col > 0 ? 1 : 0) + 1 + (col < W - 1 ? 1 : 0));
  const float u_hi = (float)(W - 1), v_hi = (float)(H - 1);
  const unsigned row_b = 12u * (unsigned)W;
  const float gl1 = g_l1[job.b] * inv_count;
  const float gss_h = g_ssim[job.b] * inv_count * (-0.5f);

  float G[9];                               // sum of dq (x) (d row, d, 1); the (d col) column is col * the (d) column
#pragma unroll
  for (int i = 0; i < 9; ++i) G[i] = 0.f;
  float acc_l1 = 0.f, acc_ss = 0.f;
  float hA[NW], hB[NW], hC[NW], hpair[NW], cA[NC], cB[NC], cC[NC], cpair[NC];
#pragma unroll
  for (int i = 0; i < NW; ++i) { hA[i] = 0.f; hB[i] = 0.f; hC[i] = 0.f; hpair[i] = 0.f; }
#pragma unroll
  for (int i = 0; i < NC; ++i) { cA[i] = 0.f; cB[i] = 0.f; cC[i] = 0.f; cpair[i] = 0.f; }
  float nb1 = 0.f, nb2 = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) ring[k * 64] = 0.f;

  // in flight between two row steps: the taps and the target pixel of the next row, the depth of the row after it
  f32x4 a0, a1;
  f32x2 b0, b1;
  f32x3 tx;
  float nd2;
  float c_d, c_zinv, c_us, c_vs, c_okf;     // the next row's projection (invalid pixels: coordinates 0, everything else times 0)
  auto issue = [&](int rn, float dd) {
    const bool row_in = (rn >= 0) && (rn < H);                           // wave-uniform
    const float fr = (float)rn;
    const float q0 = (f.M[1] * fr + m_c[0]) * dd + f.kt[0];
    const float q1 = (f.M[4] * fr + m_c[1]) * dd + f.kt[1];
    const float q2 = (f.M[7] * fr + m_c[2]) * dd + f.kt[2];
    const float zinv = rcpf(q2 + 1e-10f);
    const float up = q0 * zinv, vp = q1 * zinv;
    const bool ok = row_in && col_in && (up >= 0.f) && (up < u_hi) && (vp >= 0.f) && (vp < v_hi) && (dd != 0.f);
    c_okf = ok ? 1.f : 0.f;
    c_us = ok ? up : 0.f; c_vs = ok ? vp : 0.f;
    c_d = dd; c_zinv = ok ? zinv : 0.f;
    const int iu = cvt_floor(c_us), iv = cvt_floor(c_vs);
    const unsigned off0 = (unsigned)iv * row_b + (unsigned)iu * 12u;
    const unsigned off1 = off0 + row_b;
    a0 = *(const f32x4u*)(simg + off0);
    b0 = *(const f32x2u*)(simg + off0 + 16u);
    a1 = *(const f32x4u*)(simg + off1);
    b1 = *(const f32x2u*)(simg + off1 + 16u);
    const unsigned pt = (unsigned)(min(max(rn, 0), H - 1) * W + col_c);
    tx = *(const f32x3u*)(timg + pt * 12u);
    const unsigned pd = (unsigned)(min(max(rn + 1, 0), H - 1) * W + col_c);
    nd2 = *(const float*)((const char*)dimg + pd * 4u);
  };

  auto flush_dd = [&](int first_row, int nrows) {
    __syncthreads();
    for (int row = wid; row < nrows; row += 4) {
      const float* v = lds_dd + row * 64 + lane;
      const float sum = ((v[0] + v[DD_ROWS * 64]) + v[2 * DD_ROWS * 64]) + v[3 * DD_ROWS * 64];
      if (out_lane) gimg[(first_row + row) * W + col] = sum;
    }
    __syncthreads();
  };

  auto body = [&](int r, float (&hcur)[NW], const float (&hm1)[NW], float (&ccur)[NC], const float (&cm1)[NC]) {
    float* slot = ring + (r & 1) * (16 * 64);                              // holds row r-2 now, row r at the end of the step
    RowSt scur;
    // ---- stage A: row r from its taps (issued one row step ago)
    {
      const bool row_in = (r >= 0) && (r < H);                           // wave-uniform
      const float rm = row_in ? colm : 0.f;
      scur.x[0] = tx.x * rm; scur.x[1] = tx.y * rm; scur.x[2] = tx.z * rm;
      scur.d = c_d; scur.zinv = c_zinv; scur.up = c_us; scur.vp = c_vs;
      const float okf = c_okf;
      const float wuc = __builtin_amdgcn_fractf(c_us), wvc = __builtin_amdgcn_fractf(c_vs);
      const float ff[3] = {a0.x, a0.y, a0.z}, cf[3] = {a0.w, b0.x, b0.y};
      const float fc[3] = {a1.x, a1.y, a1.z}, cc[3] = {a1.w, b1.x, b1.y};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float ca = cf[c] - ff[c], eb = cc[c] - fc[c];
        const float top = ff[c] + ca * wuc, bot = fc[c] + eb * wuc;
        const float gv = bot - top;
        scur.y[c] = (top + gv * wvc) * okf;
        scur.gv[c] = gv * okf;
        scur.gu[c] = (ca + (eb - ca) * wvc) * okf;
      }
      scur.nb = (((scur.y[0] + scur.y[1]) + scur.y[2]) == 0.f) ? 0.f : 1.f;
    }
    // ---- the next row's gathers leave now (into the registers stage A has just read)
    issue(r + 1, nd2);
    {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        hcur[c] = hsum3(scur.x[c]);
        hcur[3 + c] = hsum3(scur.y[c]);
        hcur[6 + c] = hsum3(scur.x[c] * scur.x[c] + scur.y[c] * scur.y[c]);
        hcur[9 + c] = hsum3(scur.x[c] * scur.y[c]);
      }
    }
    // ---- stage B: SSIM coefficients of centre row p = r-1, then their horizontal sums
    {
      const int p = r - 1;
      if (p >= 0 && p < H && p >= r0 - 1 && p <= r1) {                   // wave-uniform
        const SsimK k = ssim_consts(cnt_c, window_rows(p, H));
        const float gg = gss_h * nb1;
        float co[NC], lsum = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          lsum += ssim_coeffs_sums(hpair[c] + hcur[c], hpair[3 + c] + hcur[3 + c], hpair[6 + c] + hcur[6 + c],
                                   hpair[9 + c] + hcur[9 + c], k, gg, co[c], co[3 + c], co[6 + c]);
        if (p >= r0 && p < r1) acc_ss += lsum * (outm * nb1);
#pragma unroll
        for (int i = 0; i < NC; ++i) ccur[i] = hsum3(co[i]);
      } else {
#pragma unroll
        for (int i = 0; i < NC; ++i) ccur[i] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < NW; ++i) hpair[i] = hm1[i] + hcur[i];
    }
    // ---- stage C: gradient of pixel (q = r-2, col) (its state comes back from the ring); window rows q-1, q, q+1
    const int q = r - 2;
    if (q >= r0 && q < r1) {                                               // wave-uniform
      const float gl1n = gl1 * nb2;
      float du = 0.f, dv = 0.f, l1 = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float x = slot[c * 64], y = slot[(3 + c) * 64], gu = slot[(6 + c) * 64], gv = slot[(9 + c) * 64];
        const float SA = cpair[c] + ccur[c];
        const float SB = cpair[3 + c] + ccur[3 + c];
        const float SC = cpair[6 + c] + ccur[6 + c];
        const float df = y - x;
        const float sg = __builtin_amdgcn_fmed3f(df * 1e30f, -1.f, 1.f);
        l1 += fabsf(df);
        const float g = ((gl1n * sg + SA) + SB * y) + SC * x;
        du += g * gu;
        dv += g * gv;
      }
      const float dq_d = slot[12 * 64], zinv = slot[13 * 64], up = slot[14 * 64], vp = slot[15 * 64];
      acc_l1 += l1 * (outm * nb2);
      du *= outm; dv *= outm;
      const float dq0 = du * zinv, dq1 = dv * zinv;
      const float dq2 = -(du * up + dv * vp) * zinv;
      const float fq = (float)q;
      const float ddv = (dq0 * (f.M[1] * fq + m_c[0]) + dq1 * (f.M[4] * fq + m_c[1])) + dq2 * (f.M[7] * fq + m_c[2]);
      const float P1 = dq_d * fq, P2 = dq_d;
      G[0] += dq0 * P1; G[1] += dq0 * P2; G[2] += dq0;
      G[3] += dq1 * P1; G[4] += dq1 * P2; G[5] += dq1;
      G[6] += dq2 * P1; G[7] += dq2 * P2; G[8] += dq2;
      if (MODE == 0) {
        lds_dd[(wid * DD_ROWS + ((q - r0) & (DD_ROWS - 1))) * 64 + lane] = ddv;
        if (((q - r0) & (DD_ROWS - 1)) == DD_ROWS - 1 && q != r1 - 1) flush_dd(q - (DD_ROWS - 1), DD_ROWS);
      } else if (out_lane) {
        gimg[q * W + col] = ddv;
      }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) cpair[i] = cm1[i] + ccur[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      slot[c * 64] = scur.x[c]; slot[(3 + c) * 64] = scur.y[c]; slot[(6 + c) * 64] = scur.gu[c]; slot[(9 + c) * 64] = scur.gv[c];
    }
    slot[12 * 64] = scur.d; slot[13 * 64] = scur.zinv; slot[14 * 64] = scur.up; slot[15 * 64] = scur.vp;
    nb2 = nb1; nb1 = scur.nb;
  };

  int r = r0 - 2;
  const int rend = r1 + 1;      // inclusive
  issue(r, *(const float*)((const char*)dimg + (unsigned)(min(max(r, 0), H - 1) * W + col_c) * 4u));
  while (r <= rend) {
    body(r, hA, hC, cA, cC); ++r;
    if (r > rend) break;
    body(r, hB, hA, cB, cA); ++r;
    if (r > rend) break;
    body(r, hC, hB, cC, cB); ++r;
  }
  if (MODE == 0) {
    const int done = (r1 - r0 - 1) & ~(DD_ROWS - 1);
    flush_dd(r0 + done, r1 - r0 - done);
  }
  float Gs[12];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    Gs[4 * i] = wave_sum_all(G[3 * i + 1] * colf);
    Gs[4 * i + 1] = wave_sum_all(G[3 * i]);
    Gs[4 * i + 2] = wave_sum_all(G[3 * i + 1]);
    Gs[4 * i + 3] = wave_sum_all(G[3 * i + 2]);
  }
  acc_l1 = wave_sum_all(acc_l1);
  acc_ss = wave_sum_all(acc_ss);
  if (lane == 0) {
    const long long gw = ((long long)(job.b * d.S + job.s) * d.CH + job.ck) * d.N + job.n;
    float KtG[12];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int bcol = 0; bcol < 4; ++bcol)
        KtG[4 * i + bcol] = (cam.k[i] * Gs[bcol] + cam.k[3 + i] * Gs[4 + bcol]) + cam.k[6 + i] * Gs[8 + bcol];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        part[16 * gw + 4 * i + j] = (KtG[4 * i] * cam.ki[3 * j] + KtG[4 * i + 1] * cam.ki[3 * j + 1]) + KtG[4 * i + 2] * cam.ki[3 * j + 2];
      part[16 * gw + 4 * i + 3] = KtG[4 * i + 3];
    }
    part[16 * gw + 12] = acc_l1;
    part[16 * gw + 13] = acc_ss;
  }
}

#ifndef MARCH_BWD_WAVES
#define MARCH_BWD_WAVES 3     // waves per SIMD the register allocation is asked to allow
#endif
template <int MODE>
__global__ __launch_bounds__(256, MARCH_BWD_WAVES) void march_bwd_ms_kernel(MArgs m, const float* __restrict__ T, const float* __restrict__ K,
                                                           float* __restrict__ part) {
  extern __shared__ float lds_dyn[];
  const int s = scale_of(m, blockIdx.x);
  const unsigned local = blockIdx.x - m.block_off[s], n = m.block_off[s + 1] - m.block_off[s];
  march_bwd_body<MODE>(m.src[s], m.depth[s], T, K, m.target[s], m.g_l1[s], m.g_ss[s], m.ddepth[s], part + m.part_off[s],
                       m.d[s], m.inv_count[s], xcd_contiguous(local, n), lds_dyn);
}

template <int MODE>
__global__ __launch_bounds__(256, MARCH_BWD_WAVES) void march_bwd_ms_p_kernel(MArgs m, const float* __restrict__ T, const float* __restrict__ K,
                                                                               float* __restrict__ part) {
  extern __shared__ float lds_dyn[];
  const int s = scale_of(m, blockIdx.x);
  const unsigned local = blockIdx.x - m.block_off[s], n = m.block_off[s + 1] - m.block_off[s];
  // the launch is ONE round of waves whose length is set by its longest waves (the chunks of the finest scale run more row
  // steps than those of the coarse scales): they win the issue arbitration of their SIMD, the short waves fill the gaps
  if (m.prio[s] >= 3) __builtin_amdgcn_s_setprio(3);
  else if (m.prio[s] == 2) __builtin_amdgcn_s_setprio(2);
  else if (m.prio[s] == 1) __builtin_amdgcn_s_setprio(1);
  march_bwd_body_p<MODE>(m.src[s], m.depth[s], T, K, m.target[s], m.g_l1[s], m.g_ss[s], m.ddepth[s], part + m.part_off[s],
                         m.d[s], m.inv_count[s], xcd_contiguous(local, n), lds_dyn);
}

// One finishing launch: workgroups [0, nb_dT) add the pose-gradient partials over the waves and the scales (in scale
// order) into dT [B,N,4,4] (last row 0), workgroups [nb_dT, nb_dT + B * nscales) the loss partials (slots 12 / 13).
__global__ __launch_bounds__(256) void march_bwd_finish_kernel(MArgs m, int nscales, const float* __restrict__ part,
                                                               float* __restrict__ dT, float* __restrict__ losses, int B, int N,
                                                               unsigned nb_dT) {
  if (blockIdx.x < nb_dT) {
    const int BN = B * N;
    const int e = blockIdx.x * 16 + (threadIdx.x >> 4);      // entry in [0, BN*16)
    const int t = threadIdx.x & 15;
    const int bn = e / 16, i = e % 16;
    float total = 0.f;
    for (int s = 0; s < nscales; ++s) {
      float sum = 0.f;
      if (bn < BN && i < 12) {
        const int b = bn / N, n = bn % N;
        const int waves_per_b = m.waves_per_b[s], per_view = waves_per_b / N;
        const float* ps = part + m.part_off[s];
        for (int k = t; k < per_view; k += 16) sum += ps[16 * ((long long)b * waves_per_b + (long long)k * N + n) + i];
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) sum += __shfl_down(sum, off, 16);
      total += sum;
    }
    if (t == 0 && bn < BN) dT[e] = total;
  } else if (threadIdx.x < 64) {
    const int id = blockIdx.x - nb_dT, b = id % B, s = id / B, t = threadIdx.x;
    const int waves_per_b = m.waves_per_b[s];
    const float* q = part + m.part_off[s] + (long long)b * waves_per_b * 16;
    float s0 = 0.f, s1 = 0.f;
    for (int k = t; k < waves_per_b; k += 64) { s0 += q[16 * k + 12]; s1 += q[16 * k + 13]; }
    s0 = wave_sum_all(s0);
    s1 = wave_sum_all(s1);
    if (t == 0) {
      losses[s * B + b] = s0 * m.inv_count[s];
      losses[(nscales + s) * B + b] = s1 * m.inv_count[s];
    }
  }
}

inline MDims make_dims(int B, int N, int h, int w, float scale, int rows_per_chunk, int strip) {
  MDims d;
  d.B = B; d.N = N; d.h = h; d.w = w; d.scale = scale;
  d.S = (w + strip - 1) / strip;
  d.R = rows_per_chunk < h ? rows_per_chunk : h;
  d.CH = (h + d.R - 1) / d.R;
  return d;
}

int g_fwd_min_waves = 4096, g_bwd_min_waves = 1536, g_min_rows = 8;
int g_bwd_max_rows = 32;
int g_bwd_prio = 1;                      // 1: wave priorities by chunk length in the backward launch
int g_bwd_variant = 1;                   // 1: pipelined row step (march_bwd_body_p); 0: the round-3 body
int g_bwd_rows[4] = {0, 0, 0, 0};        // rows per chunk of the backward launch per scale (0 = automatic)

// Rows per chunk: enough waves to cover 256 CUs x 4 SIMDs a few times over, as few halo rows as possible.
inline int pick_rows(int B, int N, int h, int w, long long min_waves, int max_rows = 32) {
  const long long strips = (long long)B * N * ((w + MSTRIP_B - 1) / MSTRIP_B);
  int R = max_rows;
  while (R > g_min_rows && strips * ((h + R - 1) / R) < min_waves) R >>= 1;
  return R;
}

}  // namespace

extern "C" {

int xpt_photo_march_tune(int fwd_min_waves, int bwd_min_waves, int min_rows) {
  if (fwd_min_waves < 1 || bwd_min_waves < 1 || fwd_min_waves > 16384 || bwd_min_waves > 16384) return XPT_ERR_ARG;
  if (min_rows != 2 && min_rows != 4 && min_rows != 8 && min_rows != 16 && min_rows != 32) return XPT_ERR_ARG;
  g_fwd_min_waves = fwd_min_waves;
  g_bwd_min_waves = bwd_min_waves;
  g_min_rows = min_rows;
  return XPT_OK;
}

int xpt_photo_march_plan(int bwd_variant, int rows_s0, int rows_s1, int rows_s2, int rows_s3) {
  const int rows[4] = {rows_s0, rows_s1, rows_s2, rows_s3};
  if (bwd_variant < 0 || bwd_variant > 2) return XPT_ERR_ARG;      // 2: the pipelined row step without wave priorities (lab)
  g_bwd_prio = bwd_variant != 2;
  if (bwd_variant == 2) bwd_variant = 1;
  for (int s = 0; s < 4; ++s)
    if (rows[s] < 0 || rows[s] > 4096) return XPT_ERR_ARG;
  g_bwd_variant = bwd_variant;
  for (int s = 0; s < 4; ++s) g_bwd_rows[s] = rows[s];
  return XPT_OK;
}

int xpt_photo_march_ms_fwd(int nscales, const float* const* src, const float* const* depth, const float* T, const float* K,
                           const float* const* target, float* losses, float* workspace, size_t workspace_floats,
                           int B, int N, const int* h, const int* w, const float* scale, void* stream) {
  XPT_CHECK_PTR(src); XPT_CHECK_PTR(depth); XPT_CHECK_PTR(T); XPT_CHECK_PTR(K); XPT_CHECK_PTR(target);
  XPT_CHECK_PTR(workspace); XPT_CHECK_PTR(h); XPT_CHECK_PTR(w); XPT_CHECK_PTR(scale);     // losses == NULL: partials only
  if (nscales < 1 || nscales > 4 || B <= 0 || N <= 0) return XPT_ERR_ARG;
  MArgs m{};
  size_t need = 0;
  unsigned blocks = 0;
  for (int s = 0; s < nscales; ++s) {
    if (!src[s] || !depth[s] || !target[s]) return XPT_ERR_NULL;
    // lane offsets into one image are 32-bit BYTE offsets
    if (h[s] <= 0 || w[s] <= 0 || !(scale[s] > 0.f) || (long long)h[s] * w[s] * 12 >= (1LL << 31)) return XPT_ERR_SHAPE;
    MDims d = make_dims(B, N, h[s], w[s], scale[s], pick_rows(B, N, h[s], w[s], g_fwd_min_waves), MSTRIP);
    const long long nwaves = (long long)d.B * d.S * d.CH * d.N;
    m.src[s] = src[s]; m.depth[s] = depth[s]; m.target[s] = target[s];
    m.d[s] = d;
    m.part_off[s] = (long long)need;
    m.inv_count[s] = 1.0f / ((float)N * (float)h[s] * (float)w[s] * 3.0f);
    m.waves_per_b[s] = d.S * d.CH * d.N;
    m.block_off[s] = blocks;
    blocks += (unsigned)((nwaves + 3) / 4);
    need += xpt_photo_fused_workspace_floats(B, N, h[s], w[s]);
  }
  for (int s = nscales; s <= 4; ++s) m.block_off[s] = blocks;
  if (workspace_floats < need) return XPT_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(march_fwd_ms_kernel, dim3(blocks), dim3(256), 0, st, m, T, K, workspace);
  if (losses) hipLaunchKernelGGL(march_reduce_ms_kernel, dim3(B, nscales), dim3(64), 0, st, m, workspace, losses);
  return xpt_launch_status();
}

/* backward (and, with losses != NULL, the forward values too) of all scales in one launch + one finishing launch */
int xpt_photo_march_ms_fwdbwd(int nscales, const float* const* src, const float* const* depth, const float* T, const float* K,
                              const float* const* target, const float* const* g_l1, const float* const* g_ssim,
                              float* losses, float* const* ddepth, float* dT, float* workspace, size_t workspace_floats,
                              int B, int N, const int* h, const int* w, const float* scale, void* stream) {
  XPT_CHECK_PTR(src); XPT_CHECK_PTR(depth); XPT_CHECK_PTR(T); XPT_CHECK_PTR(K); XPT_CHECK_PTR(target); XPT_CHECK_PTR(g_l1);
  XPT_CHECK_PTR(g_ssim); XPT_CHECK_PTR(ddepth); XPT_CHECK_PTR(dT); XPT_CHECK_PTR(workspace); XPT_CHECK_PTR(h);
  XPT_CHECK_PTR(w); XPT_CHECK_PTR(scale);
  if (nscales < 1 || nscales > 4 || B <= 0 || (N != 4 && N != 1)) return XPT_ERR_ARG;
  MArgs m{};
  size_t need = 0;
  unsigned blocks = 0;
  int maxR = 0;
  for (int s = 0; s < nscales; ++s) {
    if (!src[s] || !depth[s] || !target[s] || !g_l1[s] || !g_ssim[s] || !ddepth[s]) return XPT_ERR_NULL;
    if (h[s] <= 0 || w[s] <= 0 || !(scale[s] > 0.f) || (long long)h[s] * w[s] * 12 >= (1LL << 31)) return XPT_ERR_SHAPE;
    const int rows = g_bwd_rows[s] > 0 ? g_bwd_rows[s] : pick_rows(B, N, h[s], w[s], g_bwd_min_waves, g_bwd_max_rows);
    MDims d = make_dims(B, N, h[s], w[s], scale[s], rows, MSTRIP_B);
    const long long nwaves = (long long)d.B * d.S * d.CH * d.N;
    m.src[s] = src[s]; m.depth[s] = depth[s]; m.target[s] = target[s];
    m.g_l1[s] = g_l1[s]; m.g_ss[s] = g_ssim[s]; m.ddepth[s] = ddepth[s];
    m.d[s] = d;
    m.part_off[s] = (long long)need;
    m.inv_count[s] = 1.0f / ((float)N * (float)h[s] * (float)w[s] * 3.0f);
    m.waves_per_b[s] = d.S * d.CH * d.N;
    m.block_off[s] = blocks;
    blocks += (unsigned)((nwaves + 3) / 4);
    need += xpt_photo_fused_workspace_floats(B, N, h[s], w[s]);
    maxR = d.R > maxR ? d.R : maxR;
  }
  for (int s = nscales; s <= 4; ++s) m.block_off[s] = blocks;
  if (workspace_floats < need) return XPT_ERR_WORKSPACE;
  if (g_bwd_prio) {             // priority 3 for the scales with the most row steps per wave, one level less per shorter class
    for (int s = 0; s < nscales; ++s) {
      int longer = 0;
      for (int t = 0; t < nscales; ++t) longer += (m.d[t].R > m.d[s].R && (t == 0 || m.d[t].R != m.d[t - 1].R)) ? 1 : 0;
      m.prio[s] = 3 - longer < 0 ? 0 : 3 - longer;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
  const size_t ring_bytes = 4 * RING_FLOATS * sizeof(float);          // row-state rings of the four waves
  if (g_bwd_variant == 1) {
    if (N == 4) hipLaunchKernelGGL(march_bwd_ms_p_kernel<0>, dim3(blocks), dim3(256), ring_bytes + (size_t)DD_ROWS * 1024, st, m, T, K, workspace);
    else hipLaunchKernelGGL(march_bwd_ms_p_kernel<1>, dim3(blocks), dim3(256), ring_bytes, st, m, T, K, workspace);
  } else {
    if (N == 4) hipLaunchKernelGGL(march_bwd_ms_kernel<0>, dim3(blocks), dim3(256), ring_bytes + (size_t)DD_ROWS * 1024, st, m, T, K, workspace);
    else hipLaunchKernelGGL(march_bwd_ms_kernel<1>, dim3(blocks), dim3(256), ring_bytes, st, m, T, K, workspace);
  }
  const unsigned nb_dT = (unsigned)((B * N * 16 + 15) / 16);
  hipLaunchKernelGGL(march_bwd_finish_kernel, dim3(nb_dT + (losses ? (unsigned)(B * nscales) : 0u)), dim3(256), 0, st, m, nscales,
                     workspace, dT, losses, B, N, nb_dT);
  return xpt_launch_status();
}

int xpt_photo_march_ms_bwd(int nscales, const float* const* src, const float* const* depth, const float* T, const float* K,
                           const float* const* target, const float* const* g_l1, const float* const* g_ssim,
                           float* const* ddepth, float* dT, float* workspace, size_t workspace_floats, int B, int N,
                           const int* h, const int* w, const float* scale, void* stream) {
  return xpt_photo_march_ms_fwdbwd(nscales, src, depth, T, K, target, g_l1, g_ssim, nullptr, ddepth, dT, workspace,
                                   workspace_floats, B, N, h, w, scale, stream);
}

}  // extern "C"
