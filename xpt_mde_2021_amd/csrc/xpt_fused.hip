// xpt_fused.hip -- fused view synthesis + L1 + SSIM "march" kernels for gfx950 (the HBM-bound core of the hot path).
//
// Replaces, in ONE pass per scale and without materialising the synthesized views:
//   SynthesizeSingleScale.synthesize_batch_view   model/synthesize/synthesize_base.py:88-178
//   BilinearInterpolation.__call__                model/synthesize/bilinear_interp.py:7-147
//   photometric_loss_l1 / photometric_loss_ssim   model/loss_and_metric/loss_util.py:6-25, 52-96
// (the reference materialises ~430 B per warped pixel for these; algorithmic traffic is
//  P (16 + 12 N) bytes forward without the synth output, P (20 + 12 N) backward).
//
// Mapping (CDNA4, 64-wide waves):
//   * one WAVE owns a column strip of one source view (lane = column, 1 or 2 halo lanes per side) and marches down a
//     chunk of rows keeping the last two rows' horizontal window sums in registers;
//   * the 3x3 SSIM window = horizontal 3-sums through DPP wave shifts (v_add_f32_dpp wave_shr:1 / wave_shl:1, no
//     LDS traffic) + a 3-row sliding sum in registers, so every input pixel is loaded once per view;
//   * lanes are consecutive target pixels -> depth / target rows are contiguous 256 B / 768 B segments and the four
//     bilinear taps of neighbouring lanes fall into the same or adjacent 128 B lines of the source row;
//   * the 4 waves of a workgroup take the N = 4 source views of the same strip, so the shared target / depth rows
//     come from the CU's L1 after the first wave touched them; in the backward they also combine their d_depth
//     contributions through LDS so that d_depth is written exactly once;
//   * everything that is uniform per wave (camera, pose, image bases) is forced into SGPRs (readfirstlane on the
//     wave index), divisions are v_rcp_f32 (1 ulp; the bar is 1e-4), pixel offsets are 32-bit;
//   * per-wave partial sums go to a workspace and are reduced in a fixed order (deterministic, no float atomics).
#include "xpt_common.h"

using namespace xpt;

#define SSIM_C1 (0.01f * 0.01f)
#define SSIM_C2 (0.03f * 0.03f)
#define STRIP 62          // forward: output columns per wave (one halo lane on each side)
#define STRIP_B 60        // backward: two halo lanes on each side

namespace {

__device__ inline float wave_shr1(float v) {   // lane i <- lane i-1 (0 into lane 0)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ inline float wave_shl1(float v) {   // lane i <- lane i+1 (0 into lane 63)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ inline float hsum3(float v) { return (wave_shr1(v) + v) + wave_shl1(v); }
__device__ inline float rcpf(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ inline float wave_sum_all(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

struct FusedDims {
  int B, N, h, w, S, CH, R;   // S strips per row, CH row chunks, R rows per chunk
  float scale;
  int lds_ok;                 // source rows / images are 16-byte aligned: the LDS-staged variant may copy aligned lines
  int dbg;                    // diagnostic of the compiler-scheduled march: 1 = no gathers at all (taps = target pixel),
                              // i.e. what the kernel costs without its neighbour loads
};

// Which (batch, view, strip, chunk) a wave works on; all members are wave-uniform (SGPRs).
struct WaveJob {
  int b, n, s, ck;
  bool valid;
};

__device__ inline WaveJob wave_job(const FusedDims& d, unsigned block) {
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

// projection with reciprocal instead of IEEE division
__device__ inline void project_fast(const Cam& c, const Pose& p, Warp& w) {
  float Xs[3], q[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) Xs[i] = p.r[3 * i + 0] * w.X[0] + p.r[3 * i + 1] * w.X[1] + p.r[3 * i + 2] * w.X[2] + p.t[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) q[i] = c.k[3 * i + 0] * Xs[0] + c.k[3 * i + 1] * Xs[1] + c.k[3 * i + 2] * Xs[2];
  w.zinv = rcpf(q[2] + 1e-10f);
  w.up = q[0] * w.zinv;
  w.vp = q[1] * w.zinv;
}

// loads the four taps (ff, fc, cf, cc) x RGB of a pixel; 32-bit offsets from the (scalar) image base
__device__ inline void load_taps(const float* __restrict__ simg, int w, const Taps& t, float tap[12]) {
  const int off = (t.vf * w + t.uf) * 3, oc = (t.vc * w + t.uf) * 3, du = (t.uc - t.uf) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    tap[c] = simg[off + c];
    tap[3 + c] = simg[oc + c];
    tap[6 + c] = simg[off + du + c];
    tap[9 + c] = simg[oc + du + c];
  }
}

// The forward needs sigma_x and sigma_y only as their sum, so ONE window sum Sq = S(x^2 + y^2) stands in for S(x^2) and
// S(y^2): 12 window sums per pixel [x(3) y(3) x^2+y^2 (3) xy(3)], not 15.
constexpr int NW = 12;

// SSIM loss value of one channel from the 3x3 window sums (loss_util.py:80-93)
__device__ inline float ssim_loss(float Sx, float Sy, float Sq, float Sxy, float ic) {
  const float mux = Sx * ic, muy = Sy * ic;
  const float m2 = mux * mux + muy * muy;
  const float sxy = Sxy * ic - mux * muy;
  const float n = (2.f * mux * muy + SSIM_C1) * (2.f * sxy + SSIM_C2);
  const float dn = (m2 + SSIM_C1) * ((Sq * ic - m2) + SSIM_C2);
  return clampf((1.f - n * rcpf(dn)) * 0.5f, 0.f, 1.f);
}

// two channels at once on packed-fp32 VALU ops (v_pk_add/mul/fma_f32: two results per issue slot)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ inline f32x2 ssim_loss2(f32x2 Sx, f32x2 Sy, f32x2 Sq, f32x2 Sxy, float ic) {
  const f32x2 mux = Sx * ic, muy = Sy * ic;
  const f32x2 m2 = mux * mux + muy * muy;
  const f32x2 sxy = Sxy * ic - mux * muy;
  const f32x2 n = (2.f * mux * muy + SSIM_C1) * (2.f * sxy + SSIM_C2);
  const f32x2 dn = (m2 + SSIM_C1) * ((Sq * ic - m2) + SSIM_C2);
  f32x2 l;
  l.x = clampf((1.f - n.x * rcpf(dn.x)) * 0.5f, 0.f, 1.f);
  l.y = clampf((1.f - n.y * rcpf(dn.y)) * 0.5f, 0.f, 1.f);
  return l;
}

// horizontal 3-sums of the NW window terms of one row (x = target pixel, y = synthesized pixel of this lane)
__device__ inline void window_terms(const float (&x)[3], const float (&y)[3], float (&cur)[NW]) {
  const f32x2 x01{x[0], x[1]}, y01{y[0], y[1]};
  const f32x2 q01 = x01 * x01 + y01 * y01, xy01 = x01 * y01;
  const float p[NW] = {x[0], x[1], x[2], y[0], y[1], y[2], q01.x, q01.y, x[2] * x[2] + y[2] * y[2],
                       xy01.x, xy01.y, x[2] * y[2]};
#pragma unroll
  for (int i = 0; i < NW; ++i) cur[i] = hsum3(p[i]);
}

// sum over the three channels of the SSIM loss of the window (m2, m1, cur) = rows (r-2, r-1, r), centre r-1
__device__ inline float ssim_row(const float (&m2)[NW], const float (&m1)[NW], const float (&cur)[NW], float ic) {
  // channels 0 and 1 ride together in packed registers, channel 2 stays scalar (same operation order per channel)
  f32x2 W[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    W[q] = (f32x2{m2[3 * q], m2[3 * q + 1]} + f32x2{m1[3 * q], m1[3 * q + 1]}) + f32x2{cur[3 * q], cur[3 * q + 1]};
  const f32x2 l01 = ssim_loss2(W[0], W[1], W[2], W[3], ic);
  const f32x2 Wa = (f32x2{m2[2], m2[5]} + f32x2{m1[2], m1[5]}) + f32x2{cur[2], cur[5]};        // (x, y) of channel 2
  const f32x2 Wb = (f32x2{m2[8], m2[11]} + f32x2{m1[8], m1[11]}) + f32x2{cur[8], cur[11]};     // (x^2 + y^2, xy)
  const float l2 = ssim_loss(Wa.x, Wa.y, Wb.x, Wb.y, ic);
  return (l01.x + l01.y) + l2;
}

__device__ inline float window_rcp(int r, int h, float cnt_c) {
  const float cnt_r = (float)((r > 0 ? 1 : 0) + 1 + (r < h - 1 ? 1 : 0));
  return rcpf(cnt_r * cnt_c);
}

// ------------------------------------------------------------------------------------------------ forward
// part[16 * wave + {0,1}] = (sum of L1 terms, sum of SSIM terms) over the wave's output pixels (3 channels each).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));

// PIPE (only without EMIT_SYNTH): every vector load of the row loop is issued by hand (inline asm) with hand-placed
// s_waitcnt, so that the streaming loads of row r+1 (depth, target) are in flight BEHIND the gathers of row r and are
// only waited for when row r's arithmetic is done -- the compiler's own waitcnt insertion cannot express that (vector
// memory returns in order and loop-carried load destinations are waited for conservatively).
// (the march itself is a device function of the workgroup index: the per-scale kernel passes blockIdx.x, the multi-scale
//  kernel below the index inside its scale's range of workgroups)
template <bool EMIT_SYNTH, bool PIPE>
__device__ __forceinline__ void fused_fwd_body(const float* __restrict__ src, const float* __restrict__ depth,
                                               const float* __restrict__ T, const float* __restrict__ K,
                                               const float* __restrict__ target, float* __restrict__ synth,
                                               float* __restrict__ part, const FusedDims& d, unsigned block) {
  const WaveJob job = wave_job(d, block);
  if (!job.valid) return;                          // no block-level synchronisation in this kernel
  const int lane = threadIdx.x & 63;
  const int P = d.h * d.w;
  const int col = job.s * STRIP - 1 + lane;
  const bool col_in = (col >= 0) && (col < d.w);
  const bool out_lane = (lane >= 1) && (lane <= STRIP) && col_in;
  const int r0 = job.ck * d.R, r1 = min(r0 + d.R, d.h);
  const Cam cam = load_cam(K + 9 * job.b, d.scale);
  const Pose pose = load_pose(T + 16 * (job.b * d.N + job.n));
  const float* simg = src + (long long)(job.b * d.N + job.n) * P * 3;
  const float* dimg = depth + (long long)job.b * P;
  const float* timg = target + (long long)job.b * P * 3;
  float* oimg = EMIT_SYNTH ? synth + (long long)(job.b * d.N + job.n) * P * 3 : nullptr;
  const float cnt_c = (float)((col > 0 ? 1 : 0) + 1 + (col < d.w - 1 ? 1 : 0));
  // K (R (d Kinv (col, r, 1)) + t) = d (M (col, r, 1)) + K t with M = K R Kinv: the two wave-uniform 3x3 products are
  // folded once per wave, leaving 3 FMAs per row for M (col, r, 1) and 3 for the depth (the reference's chain of
  // pixel2cam / transform / cam2pixel, synthesize_base.py:106-178, re-associated: coordinates move by ~1e-7 relative)
  float M[9], kt[3], m_c[3];
  {
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
        M[3 * i + j] = KR[3 * i] * cam.ki[j] + KR[3 * i + 1] * cam.ki[3 + j] + KR[3 * i + 2] * cam.ki[6 + j];
      kt[i] = cam.k[3 * i] * pose.t[0] + cam.k[3 * i + 1] * pose.t[1] + cam.k[3 * i + 2] * pose.t[2];
      m_c[i] = M[3 * i] * (float)col + M[3 * i + 2];
    }
  }
  const float fu_max = (float)(d.w - 2), fv_max = (float)(d.h - 2);
  const int row3 = 3 * d.w;

  float acc_l1 = 0.f, acc_ss = 0.f;
  float hA[NW], hB[NW], hC[NW];                    // horizontal window sums [x(3) y(3) xx(3) yy(3) xy(3)] of 3 rows
#pragma unroll
  for (int i = 0; i < NW; ++i) { hA[i] = 0.f; hB[i] = 0.f; hC[i] = 0.f; }
  bool black_prev = true;

  // body(r, cur, m1, m2): loads row r, fills cur, emits the SSIM of row r-1 from (m2, m1, cur)
  auto body = [&](int r, float (&cur)[NW], const float (&m1)[NW], const float (&m2)[NW]) {
    float x[3] = {0.f, 0.f, 0.f}, y[3] = {0.f, 0.f, 0.f};
    if (r >= 0 && r < d.h && col_in) {             // r is wave-uniform
      const int p = r * d.w + col;
      const float dd = dimg[p];
      x[0] = timg[3 * p]; x[1] = timg[3 * p + 1]; x[2] = timg[3 * p + 2];
      const float fr = (float)r;
      const float q0 = (M[1] * fr + m_c[0]) * dd + kt[0];
      const float q1 = (M[4] * fr + m_c[1]) * dd + kt[1];
      const float q2 = (M[7] * fr + m_c[2]) * dd + kt[2];
      const float zinv = rcpf(q2 + 1e-10f);
      const float up = q0 * zinv, vp = q1 * zinv;
      // BilinearInterpolation (bilinear_interp.py:34-102): with floor(u') = fu the clipped neighbours satisfy
      // uf + 1 == uc exactly when 0 <= fu <= w - 2 (same for v), so validity is four compares and the four taps sit
      // at fixed offsets (0, 3, 3 w, 3 w + 3) from one address; invalid pixels read tap 0 and are zeroed below
      const float fu = floorf(up), fv = floorf(vp);
      const bool ok = (fu >= 0.f) && (fu <= fu_max) && (fv >= 0.f) && (fv <= fv_max) && (dd != 0.f);
      const int off = ok ? ((int)fv * d.w + (int)fu) * 3 : 0;
      const float wuf = (fu + 1.f) - up, wuc = up - fu, wvf = (fv + 1.f) - vp, wvc = vp - fv;
      const float wff = wuf * wvf, wfc = wuf * wvc, wcf = wuc * wvf, wcc = wuc * wvc;
      const float* t0 = simg + off;
      const float* t1 = t0 + row3;
      float ta[6], tb[6];
      if (d.dbg == 1) {                            // wave-uniform: diagnostic without any gather
#pragma unroll
        for (int e = 0; e < 6; ++e) { ta[e] = x[e % 3]; tb[e] = x[e % 3]; }
      } else {
#pragma unroll
        for (int e = 0; e < 6; ++e) { ta[e] = t0[e]; tb[e] = t1[e]; }
      }
      const f32x2 v01 = ((f32x2{ta[0], ta[1]} * wff + f32x2{tb[0], tb[1]} * wfc) + f32x2{ta[3], ta[4]} * wcf) +
                        f32x2{tb[3], tb[4]} * wcc;
      const float v2 = ((ta[2] * wff + tb[2] * wfc) + ta[5] * wcf) + tb[5] * wcc;
      y[0] = ok ? v01.x : 0.f;
      y[1] = ok ? v01.y : 0.f;
      y[2] = ok ? v2 : 0.f;
      if (EMIT_SYNTH && out_lane && r >= r0 && r < r1) {
        oimg[3 * p] = y[0]; oimg[3 * p + 1] = y[1]; oimg[3 * p + 2] = y[2];
      }
    }
    const bool black = ((y[0] + y[1]) + y[2]) == 0.f;       // mean_c == 0  <=>  sum_c == 0
    if (out_lane && r >= r0 && r < r1 && !black)
      acc_l1 += (fabsf(y[0] - x[0]) + fabsf(y[1] - x[1])) + fabsf(y[2] - x[2]);
    window_terms(x, y, cur);
    const int rc = r - 1;                          // centre row of the window (m2, m1, cur)
    if (rc >= r0 && rc < r1) {
      const float ic = window_rcp(rc, d.h, cnt_c);
      const float sum = ssim_row(m2, m1, cur, ic);
      if (out_lane && !black_prev) acc_ss += sum;
    }
    black_prev = black;
  };

  // rows r0-1 .. r1 (one halo row above and below the chunk), three-way rotation keeps the history in registers
  int r = r0 - 1;
  const int rend = r1;          // inclusive
  if constexpr (!PIPE) {
    while (r <= rend) {
      body(r, hA, hC, hB); ++r;
      if (r > rend) break;
      body(r, hB, hA, hC); ++r;
      if (r > rend) break;
      body(r, hC, hB, hA); ++r;
    }
  } else {
    // ---- hand-pipelined march (branch-free per row: out-of-image rows / columns read a clamped address, are zeroed)
    const int col_c = min(max(col, 0), d.w - 1);
    float nd;
    f32x3 nx;
    bool nvalid;
    auto prefetch = [&](int rr) {                  // 2 loads: depth and target pixel of row rr
      nvalid = rr >= 0 && rr < d.h && col_in;      // rr is wave-uniform
      const int p = min(max(rr, 0), d.h - 1) * d.w + col_c;
      const float* pd = dimg + p;
      const float* pt = timg + 3 * p;
      asm volatile("global_load_dword %0, %1, off" : "=v"(nd) : "v"(pd) : "memory");
      asm volatile("global_load_dwordx3 %0, %1, off" : "=v"(nx) : "v"(pt) : "memory");
    };
    auto pbody = [&](int r, float (&cur)[NW], const float (&m1)[NW], const float (&m2)[NW]) {
      // the prefetched depth / target of this row have been waited for (end of the previous body or below)
      const bool valid = nvalid;
      const float dd = valid ? nd : 0.f;
      const float x[3] = {valid ? nx.x : 0.f, valid ? nx.y : 0.f, valid ? nx.z : 0.f};
      const float fr = (float)r;
      const float q0 = (M[1] * fr + m_c[0]) * dd + kt[0];
      const float q1 = (M[4] * fr + m_c[1]) * dd + kt[1];
      const float q2 = (M[7] * fr + m_c[2]) * dd + kt[2];
      const float zinv = rcpf(q2 + 1e-10f);
      const float up = q0 * zinv, vp = q1 * zinv;
      const float fu = floorf(up), fv = floorf(vp);
      const bool ok = valid && (fu >= 0.f) && (fu <= fu_max) && (fv >= 0.f) && (fv <= fv_max) && (dd != 0.f);
      const int off = ok ? ((int)fv * d.w + (int)fu) * 3 : 0;
      const float wuf = (fu + 1.f) - up, wuc = up - fu, wvf = (fv + 1.f) - vp, wvc = vp - fv;
      const float wff = wuf * wvf, wfc = wuf * wvc, wcf = wuc * wvf, wcc = wuc * wvc;
      const float* t0 = simg + off;
      const float* t1 = t0 + row3;
      f32x4 a0, a1;
      f32x2 b0, b1;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a0) : "v"(t0) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, off offset:16" : "=v"(b0) : "v"(t0) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a1) : "v"(t1) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, off offset:16" : "=v"(b1) : "v"(t1) : "memory");
      prefetch(r + 1);                             // behind the gathers
      asm volatile("s_waitcnt vmcnt(2)" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1) : : "memory");   // gathers landed
      const f32x2 v01 = ((f32x2{a0.x, a0.y} * wff + f32x2{a1.x, a1.y} * wfc) + f32x2{a0.w, b0.x} * wcf) +
                        f32x2{a1.w, b1.x} * wcc;
      const float v2 = ((a0.z * wff + a1.z * wfc) + b0.y * wcf) + b1.y * wcc;
      const float y[3] = {ok ? v01.x : 0.f, ok ? v01.y : 0.f, ok ? v2 : 0.f};
      const bool black = ((y[0] + y[1]) + y[2]) == 0.f;
      if (out_lane && r >= r0 && r < r1 && !black)
        acc_l1 += (fabsf(y[0] - x[0]) + fabsf(y[1] - x[1])) + fabsf(y[2] - x[2]);
      window_terms(x, y, cur);
      const int rc = r - 1;
      if (rc >= r0 && rc < r1) {
        const float ic = window_rcp(rc, d.h, cnt_c);
        const float sum = ssim_row(m2, m1, cur, ic);
        if (out_lane && !black_prev) acc_ss += sum;
      }
      black_prev = black;
      // the next row's depth / target: waited for here (end of this row's arithmetic).  Always before leaving the body,
      // so that no register with a load in flight crosses a loop edge or a register copy.
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(nd), "+v"(nx) : : "memory");
    };
    prefetch(r);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(nd), "+v"(nx) : : "memory");
    while (r <= rend) {
      pbody(r, hA, hC, hB); ++r;
      if (r > rend) break;
      pbody(r, hB, hA, hC); ++r;
      if (r > rend) break;
      pbody(r, hC, hB, hA); ++r;
    }
  }
  acc_l1 = wave_sum_all(acc_l1);
  acc_ss = wave_sum_all(acc_ss);
  if (lane == 0) {
    const long long gw = ((long long)(job.b * d.S + job.s) * d.CH + job.ck) * d.N + job.n;
    part[16 * gw] = acc_l1;
    part[16 * gw + 1] = acc_ss;
  }
}

template <bool EMIT_SYNTH, bool PIPE>
__global__ __launch_bounds__(256) void fused_fwd_kernel(const float* __restrict__ src, const float* __restrict__ depth,
                                                        const float* __restrict__ T, const float* __restrict__ K,
                                                        const float* __restrict__ target, float* __restrict__ synth,
                                                        float* __restrict__ part, FusedDims d) {
  fused_fwd_body<EMIT_SYNTH, PIPE>(src, depth, T, K, target, synth, part, d, blockIdx.x);
}

// All scales of the loss pyramid in ONE launch: the workgroups of scale 0 come first, then scale 1, ... (the small scales
// are latency chains of ~10 dependent row steps on a handful of waves: alone each costs a 10 us launch, here they run
// under the shadow of scale 0).  The scale of a workgroup is wave-uniform; its arguments are scalar loads.
struct MsArgs {
  const float* src[4];
  const float* depth[4];
  const float* target[4];
  const float* g_l1[4];
  const float* g_ss[4];
  float* ddepth[4];
  long long part_off[4];       // offset (floats) of the scale's per-wave partials in the workspace
  FusedDims d[4];
  float inv_count[4];
  unsigned block_off[5];       // first workgroup of every scale; unused scales = the total
  int waves_per_b[4];
};

__device__ inline int ms_scale_of(const MsArgs& m, unsigned b) {
  return (int)(b >= m.block_off[1]) + (int)(b >= m.block_off[2]) + (int)(b >= m.block_off[3]);
}

__global__ __launch_bounds__(256) void fused_fwd_ms_kernel(MsArgs m, const float* __restrict__ T, const float* __restrict__ K,
                                                           float* __restrict__ part) {
  const int s = ms_scale_of(m, blockIdx.x);
  fused_fwd_body<false, true>(m.src[s], m.depth[s], T, K, m.target[s], nullptr, part + m.part_off[s], m.d[s],
                              blockIdx.x - m.block_off[s]);
}

// ------------------------------------------------------------------------------------------------ forward, LDS-staged taps
// Variant 2.  The gather of the four bilinear neighbours is what bounds the march above: every source texel is
// requested by two lanes and by two consecutive row steps, as unaligned 16 + 8 byte loads per lane and tap row -- the
// address path (TA / L1) is ~50 % busy at 25 % of the HBM rate (profiles/r01_d_pmc_fused.md).  Here a wave works in blocks of
// FB = 6 target rows:
//   pass 1  streams the block's depth / target rows (12 coalesced loads in flight together) and projects all pixels;
//   a wave reduction gives the bounding box [umin, umax] x [vmin, vmax] of the neighbours the block touches;
//   if it fits (<= FT rows, one 1 KiB line of floats per row), the box is copied ROW BY ROW into the wave's LDS tile
//   with ONE aligned 16-byte load per lane and row (all in flight together) -- each texel leaves L2 once per block;
//   pass 2  takes the 12 neighbour values of every pixel from LDS (4-byte reads, texel stride 3 dwords: conflict-free)
//   and runs the same L1 / SSIM arithmetic as the other variants.
// A block whose box does not fit (strong zoom / rotation, depth discontinuities spanning the strip) gathers from global
// memory exactly like variant 0; the decision is wave-uniform.  The same texels enter the same expressions as in the
// other variants (L1 sums are bit-identical; the SSIM sums agree to ~1e-7 relative: FMA contraction of a separately
// compiled body).
constexpr int FB = 6;            // target rows per block (a multiple of 3: the register rotation of the row history)
constexpr int FT = 10;           // source rows the LDS tile holds
constexpr int FROW = 256;        // floats per tile row = 64 lanes x 16 bytes

__device__ inline int wave_min_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ inline int wave_max_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}

__global__ __launch_bounds__(256) void fused_fwd_lds_kernel(const float* __restrict__ src, const float* __restrict__ depth,
                                                            const float* __restrict__ T, const float* __restrict__ K,
                                                            const float* __restrict__ target, float* __restrict__ part,
                                                            FusedDims d) {
  __shared__ __attribute__((aligned(16))) float tiles[4][FT * FROW];
  const WaveJob job = wave_job(d, blockIdx.x);
  if (!job.valid) return;                          // no block-level synchronisation in this kernel
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* tile = tiles[wid];
  const int P = d.h * d.w;
  const int col = job.s * STRIP - 1 + lane;
  const bool col_in = (col >= 0) && (col < d.w);
  const bool out_lane = (lane >= 1) && (lane <= STRIP) && col_in;
  const int col_c = min(max(col, 0), d.w - 1);
  const int r0 = job.ck * d.R, r1 = min(r0 + d.R, d.h);
  const Cam cam = load_cam(K + 9 * job.b, d.scale);
  const Pose pose = load_pose(T + 16 * (job.b * d.N + job.n));
  const float* simg = src + (long long)(job.b * d.N + job.n) * P * 3;
  const float* dimg = depth + (long long)job.b * P;
  const float* timg = target + (long long)job.b * P * 3;
  const float cnt_c = (float)((col > 0 ? 1 : 0) + 1 + (col < d.w - 1 ? 1 : 0));
  float M[9], kt[3], m_c[3];
  {
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
        M[3 * i + j] = KR[3 * i] * cam.ki[j] + KR[3 * i + 1] * cam.ki[3 + j] + KR[3 * i + 2] * cam.ki[6 + j];
      kt[i] = cam.k[3 * i] * pose.t[0] + cam.k[3 * i + 1] * pose.t[1] + cam.k[3 * i + 2] * pose.t[2];
      m_c[i] = M[3 * i] * (float)col + M[3 * i + 2];
    }
  }
  const float fu_max = (float)(d.w - 2), fv_max = (float)(d.h - 2);
  const int row3 = 3 * d.w;

  float acc_l1 = 0.f, acc_ss = 0.f;
  float hA[NW], hB[NW], hC[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) { hA[i] = 0.f; hB[i] = 0.f; hC[i] = 0.f; }
  bool black_prev = true;

  // one row of the march once x (target) and y (synthesized) are known: L1, horizontal window sums, SSIM of row r-1
  auto emit = [&](int r, const float (&x)[3], const float (&y)[3], float (&cur)[NW], const float (&m1)[NW],
                  const float (&m2)[NW]) {
    const bool black = ((y[0] + y[1]) + y[2]) == 0.f;
    if (out_lane && r >= r0 && r < r1 && !black)
      acc_l1 += (fabsf(y[0] - x[0]) + fabsf(y[1] - x[1])) + fabsf(y[2] - x[2]);
    window_terms(x, y, cur);
    const int rc = r - 1;
    if (rc >= r0 && rc < r1) {
      const float ic = window_rcp(rc, d.h, cnt_c);
      const float sum = ssim_row(m2, m1, cur, ic);
      if (out_lane && !black_prev) acc_ss += sum;
    }
    black_prev = black;
  };

  const int rend = r1;                             // inclusive: one halo row below the chunk
  int n_blocks = 0, n_staged = 0;                  // diagnostics (workspace slots 2, 3 of the wave)
  for (int rb = r0 - 1; rb <= rend; rb += FB) {
    // ---- pass 1: stream depth / target of the block's rows, project, find the box of touched source texels
    float dd[FB];
    f32x3 xx[FB];
#pragma unroll
    for (int i = 0; i < FB; ++i) {                 // unconditional loads from clamped rows (zeroed below)
      const int p = min(max(rb + i, 0), d.h - 1) * d.w + col_c;
      dd[i] = dimg[p];
      xx[i] = *(const f32x3*)(timg + 3 * p);
    }
    float up[FB], vp[FB];
    int offs[FB];                                  // (fv * w + fu) * 3 of valid pixels, -1 otherwise
    int umin = 0x7fffffff, umax = -1, vmin = 0x7fffffff, vmax = -1;
#pragma unroll
    for (int i = 0; i < FB; ++i) {
      const int r = rb + i;
      const bool rvalid = r >= 0 && r < d.h && r <= rend && col_in;
      if (!rvalid) { dd[i] = 0.f; xx[i] = f32x3{0.f, 0.f, 0.f}; }
      const float fr = (float)r;
      const float q0 = (M[1] * fr + m_c[0]) * dd[i] + kt[0];
      const float q1 = (M[4] * fr + m_c[1]) * dd[i] + kt[1];
      const float q2 = (M[7] * fr + m_c[2]) * dd[i] + kt[2];
      const float zinv = rcpf(q2 + 1e-10f);
      up[i] = q0 * zinv;
      vp[i] = q1 * zinv;
      const float fu = floorf(up[i]), fv = floorf(vp[i]);
      const bool ok = rvalid && (fu >= 0.f) && (fu <= fu_max) && (fv >= 0.f) && (fv <= fv_max) && (dd[i] != 0.f);
      const int iu = ok ? (int)fu : 0, iv = ok ? (int)fv : 0;
      offs[i] = ok ? (iv * d.w + iu) * 3 : -1;
      if (ok) {
        umin = min(umin, iu); umax = max(umax, iu + 1);
        vmin = min(vmin, iv); vmax = max(vmax, iv + 1);
      }
    }
    umin = wave_min_i(umin); umax = wave_max_i(umax);
    vmin = wave_min_i(vmin); vmax = wave_max_i(vmax);
    const bool any_ok = vmax >= 0;                                   // wave-uniform from here on
    // tile geometry: row segment [a0, a0 + len) of source row vmin (float indices inside the image), a0 16-byte aligned
    const int f0 = (vmin * d.w + umin) * 3;
    const int a0 = f0 & ~3;
    const int cu0 = a0 - vmin * row3;                                // float offset of tile column 0 inside a source row
    const int len = umax * 3 + 3 - cu0;                              // floats needed per row
    const bool staged = any_ok && d.lds_ok && (vmax - vmin + 1 <= FT) && (len <= FROW);
    n_blocks += any_ok ? 1 : 0;
    n_staged += staged ? 1 : 0;
    if (staged) {
      const int nrows = vmax - vmin + 1;
      f32x4 seg[FT];
#pragma unroll
      for (int t = 0; t < FT; ++t) {                                 // every needed row's line in flight together
        const bool need = t < nrows && 4 * lane < len;
        const float* g = simg + a0 + (need ? t * row3 + 4 * lane : 0);
        const f32x4 v = *(const f32x4*)g;
        seg[t] = need ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int t = 0; t < FT; ++t)
        if (t < nrows) *(f32x4*)(tile + t * FROW + 4 * lane) = seg[t];
    }
    // ---- pass 2: neighbours from the tile (or from global memory), loss arithmetic
#pragma unroll
    for (int i = 0; i < FB; ++i) {
      const int r = rb + i;
      if (r > rend) break;                                           // wave-uniform; nothing after the last row
      const bool ok = offs[i] >= 0;
      const float fu = floorf(up[i]), fv = floorf(vp[i]);
      const float wuf = (fu + 1.f) - up[i], wuc = up[i] - fu, wvf = (fv + 1.f) - vp[i], wvc = vp[i] - fv;
      const float wff = wuf * wvf, wfc = wuf * wvc, wcf = wuc * wvf, wcc = wuc * wvc;
      float a[6], b[6];
      if (staged) {
        const int o = ok ? offs[i] - vmin * row3 : cu0;              // (fv * w + fu) * 3 relative to source row vmin
        const int trow = ok ? (o / row3) : 0;                        // fv - vmin   (0 <= o - trow*row3 < row3)
        const float* t0 = tile + trow * FROW + (o - trow * row3) - cu0;
#pragma unroll
        for (int e = 0; e < 6; ++e) { a[e] = t0[e]; b[e] = t0[FROW + e]; }
      } else {
        const float* t0 = simg + (ok ? offs[i] : 0);
        const float* t1 = t0 + row3;
#pragma unroll
        for (int e = 0; e < 6; ++e) { a[e] = t0[e]; b[e] = t1[e]; }
      }
      const f32x2 v01 = ((f32x2{a[0], a[1]} * wff + f32x2{b[0], b[1]} * wfc) + f32x2{a[3], a[4]} * wcf) +
                        f32x2{b[3], b[4]} * wcc;
      const float v2 = ((a[2] * wff + b[2] * wfc) + a[5] * wcf) + b[5] * wcc;
      const float y[3] = {ok ? v01.x : 0.f, ok ? v01.y : 0.f, ok ? v2 : 0.f};
      const float x[3] = {xx[i].x, xx[i].y, xx[i].z};
      if (i % 3 == 0) emit(r, x, y, hA, hC, hB);
      else if (i % 3 == 1) emit(r, x, y, hB, hA, hC);
      else emit(r, x, y, hC, hB, hA);
    }
  }
  acc_l1 = wave_sum_all(acc_l1);
  acc_ss = wave_sum_all(acc_ss);
  if (lane == 0) {
    const long long gw = ((long long)(job.b * d.S + job.s) * d.CH + job.ck) * d.N + job.n;
    part[16 * gw] = acc_l1;
    part[16 * gw + 1] = acc_ss;
    part[16 * gw + 2] = (float)n_staged;
    part[16 * gw + 3] = (float)n_blocks;
  }
}

// loss[b] = inv_count * sum over the waves of batch element b (fixed order), 64 threads per b.
__global__ void fused_reduce_kernel(const float* __restrict__ part, float* __restrict__ loss_l1,
                                    float* __restrict__ loss_ssim, int waves_per_b, float inv_count) {
  const int b = blockIdx.x, t = threadIdx.x;
  const float* q = part + (long long)b * waves_per_b * 16;
  float s0 = 0.f, s1 = 0.f;
  for (int k = t; k < waves_per_b; k += 64) { s0 += q[16 * k]; s1 += q[16 * k + 1]; }
  s0 = wave_sum_all(s0);
  s1 = wave_sum_all(s1);
  if (t == 0) { loss_l1[b] = s0 * inv_count; loss_ssim[b] = s1 * inv_count; }
}

// ------------------------------------------------------------------------------------------------ backward
// d loss / d y(q,c) = g_l1 sign(y-x) [not black]  +  sum_{p in win(q)} ( A_pc + 2 B_pc y_qc + C_pc x_qc )
// with, for window centre p:  (A,B,C)_pc = g_ssim * (-1/2) * [not black(p)] * [0 <= (1-ssim)/2 <= 1] / cnt_p *
//                                          ( dssim/dmu_y, dssim/dE[yy], dssim/dE[xy] )
// The second 3x3 box sum reuses the same DPP + sliding-row scheme one row later, so the march has three stages per
// step: A = synthesize row r, B = coefficients of centre row r-1, C = pixel gradients of row r-2.
struct RowState {
  float gu[3], gv[3];     // d y_c / d u', d y_c / d v'   (from the taps and weights of stage A, already masked)
  float x[3], y[3];
  float d;
  bool black;
};

__device__ inline void ssim_coeffs(float Sx, float Sy, float Sq, float Sxy, float ic, float g, float& A, float& Bq,
                                   float& Cq) {
  const float mux = Sx * ic, muy = Sy * ic;
  const float m2 = mux * mux + muy * muy;
  const float sxy = Sxy * ic - mux * muy;
  const float n1 = 2.f * mux * muy + SSIM_C1, n2 = 2.f * sxy + SSIM_C2;
  const float d1 = m2 + SSIM_C1, d2 = (Sq * ic - m2) + SSIM_C2;            // sigma_x + sigma_y from S(x^2 + y^2)
  const float i1 = rcpf(d1), i2 = rcpf(d2);
  const float inv12 = i1 * i2;
  const float ssim = n1 * n2 * inv12;
  const float val = (1.f - ssim) * 0.5f;
  const float gg = (val >= 0.f && val <= 1.f) ? g * (-0.5f) * ic : 0.f;      // clip_by_value gradient, pooling divisor
  A = gg * (2.f * mux * (n2 - n1) * inv12 - 2.f * muy * ssim * (i1 - i2));
  Bq = gg * (-ssim * i2);
  Cq = gg * (2.f * n1 * inv12);
}

// MODE 0: N == 4, the 4 waves of a workgroup hold the 4 views of one strip -> d_depth summed through LDS;
// MODE 1: N == 1 -> direct store;  MODE 2: any other N -> atomicAdd into a zeroed d_depth.
template <int MODE>
__device__ __forceinline__ void fused_bwd_body(const float* __restrict__ src, const float* __restrict__ depth,
                                               const float* __restrict__ T, const float* __restrict__ K,
                                               const float* __restrict__ target, const float* __restrict__ g_l1,
                                               const float* __restrict__ g_ssim, float* __restrict__ ddepth,
                                               float* __restrict__ part, const FusedDims& d, float inv_count, unsigned block) {
  __shared__ float lds_dd[4][64];
  const WaveJob job = wave_job(d, block);
  if (!job.valid) return;                   // MODE 0: nwaves % 4 == 0, whole workgroups leave together
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int P = d.h * d.w;
  const int col = job.s * STRIP_B - 2 + lane;
  const bool col_in = (col >= 0) && (col < d.w);
  const bool out_lane = (lane >= 2) && (lane < 2 + STRIP_B) && col_in;
  const int r0 = job.ck * d.R, r1 = min(r0 + d.R, d.h);
  const Cam cam = load_cam(K + 9 * job.b, d.scale);
  const Pose pose = load_pose(T + 16 * (job.b * d.N + job.n));
  const float* simg = src + (long long)(job.b * d.N + job.n) * P * 3;
  const float* dimg = depth + (long long)job.b * P;
  const float* timg = target + (long long)job.b * P * 3;
  float* gimg = ddepth + (long long)job.b * P;
  const float cnt_c = (float)((col > 0 ? 1 : 0) + 1 + (col < d.w - 1 ? 1 : 0));
  const float gl1 = g_l1[job.b] * inv_count, gss = g_ssim[job.b] * inv_count;
  float ray_c[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) ray_c[i] = cam.ki[3 * i] * (float)col + cam.ki[3 * i + 2];

  float dRt[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) dRt[i] = 0.f;
  float hA[NW], hB[NW], hC[NW], cA[9], cB[9], cC[9];
  RowState sA, sB, sC;
#pragma unroll
  for (int i = 0; i < NW; ++i) { hA[i] = 0.f; hB[i] = 0.f; hC[i] = 0.f; }
#pragma unroll
  for (int i = 0; i < 9; ++i) { cA[i] = 0.f; cB[i] = 0.f; cC[i] = 0.f; }
  auto clear_state = [](RowState& st) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { st.gu[i] = 0.f; st.gv[i] = 0.f; st.x[i] = 0.f; st.y[i] = 0.f; }
    st.d = 0.f;
    st.black = true;
  };
  clear_state(sA); clear_state(sB); clear_state(sC);

  auto body = [&](int r, RowState& scur, const RowState& sm1, const RowState& sm2, float (&hcur)[NW],
                  const float (&hm1)[NW], const float (&hm2)[NW], float (&ccur)[9], const float (&cm1)[9],
                  const float (&cm2)[9]) {
    // ---- stage A: synthesize row r; keep y and its derivatives w.r.t. the sampling position
    clear_state(scur);
    if (r >= 0 && r < d.h && col_in) {
      const int p = r * d.w + col;
      scur.d = dimg[p];
      scur.x[0] = timg[3 * p]; scur.x[1] = timg[3 * p + 1]; scur.x[2] = timg[3 * p + 2];
      Warp wp;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        wp.ray[i] = ray_c[i] + cam.ki[3 * i + 1] * (float)r;
        wp.X[i] = wp.ray[i] * scur.d;
      }
      project_fast(cam, pose, wp);
      const Taps t = make_taps(wp.up, wp.vp, d.h, d.w, scur.d != 0.f);
      float tap[12];
      load_taps(simg, d.w, t, tap);
      const float wff = t.wuf * t.wvf, wfc = t.wuf * t.wvc, wcf = t.wuc * t.wvf, wcc = t.wuc * t.wvc;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        scur.y[c] = ((tap[c] * wff + tap[3 + c] * wfc) + tap[6 + c] * wcf) + tap[9 + c] * wcc;
        scur.gu[c] = ((tap[6 + c] - tap[c]) * t.wvf + (tap[9 + c] - tap[3 + c]) * t.wvc) * t.mask;
        scur.gv[c] = ((tap[3 + c] - tap[c]) * t.wuf + (tap[9 + c] - tap[6 + c]) * t.wuc) * t.mask;
      }
      scur.black = ((scur.y[0] + scur.y[1]) + scur.y[2]) == 0.f;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      hcur[c] = hsum3(scur.x[c]);
      hcur[3 + c] = hsum3(scur.y[c]);
      hcur[6 + c] = hsum3(scur.x[c] * scur.x[c] + scur.y[c] * scur.y[c]);
      hcur[9 + c] = hsum3(scur.x[c] * scur.y[c]);
    }
    // ---- stage B: SSIM coefficients of centre row p = r-1 (state sm1), then their horizontal sums
    {
      const int p = r - 1;
      float co[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) co[i] = 0.f;
      if (p >= 0 && p < d.h) {                        // wave-uniform
        const float ic = window_rcp(p, d.h, cnt_c);
        const float g = (col_in && !sm1.black) ? gss : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          ssim_coeffs(hm2[c] + hm1[c] + hcur[c], hm2[3 + c] + hm1[3 + c] + hcur[3 + c],
                      hm2[6 + c] + hm1[6 + c] + hcur[6 + c], hm2[9 + c] + hm1[9 + c] + hcur[9 + c], ic, g, co[c],
                      co[3 + c], co[6 + c]);
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) ccur[i] = hsum3(co[i]);
    }
    // ---- stage C: gradient of pixel (q = r-2, col) (state sm2); window rows q-1, q, q+1 = cm2, cm1, ccur
    const int q = r - 2;
    const bool emit = (q >= r0) && (q < r1);          // wave-uniform
    if (emit) {
      float dd = 0.f;
      if (out_lane) {
        float du = 0.f, dv = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float SA = cm2[c] + cm1[c] + ccur[c];
          const float SB = cm2[3 + c] + cm1[3 + c] + ccur[3 + c];
          const float SC = cm2[6 + c] + cm1[6 + c] + ccur[6 + c];
          const float df = sm2.y[c] - sm2.x[c];
          const float sg = sm2.black ? 0.f : ((df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f));
          const float g = gl1 * sg + SA + 2.f * SB * sm2.y[c] + SC * sm2.x[c];
          du += g * sm2.gu[c];
          dv += g * sm2.gv[c];
        }
        Warp wp;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          wp.ray[i] = ray_c[i] + cam.ki[3 * i + 1] * (float)q;
          wp.X[i] = wp.ray[i] * sm2.d;
        }
        project_fast(cam, pose, wp);
        const float dq0 = du * wp.zinv, dq1 = dv * wp.zinv;
        const float dq2 = -(du * wp.up + dv * wp.vp) * wp.zinv;
        float dXs[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) dXs[j] = cam.k[j] * dq0 + cam.k[3 + j] * dq1 + cam.k[6 + j] * dq2;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          dRt[4 * i + 0] += dXs[i] * wp.X[0];
          dRt[4 * i + 1] += dXs[i] * wp.X[1];
          dRt[4 * i + 2] += dXs[i] * wp.X[2];
          dRt[4 * i + 3] += dXs[i];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
          dd += (pose.r[j] * dXs[0] + pose.r[3 + j] * dXs[1] + pose.r[6 + j] * dXs[2]) * wp.ray[j];
      }
      if (MODE == 0) {                                // every wave of the workgroup shares (r0, r1, strip)
        lds_dd[wid][lane] = dd;
        __syncthreads();
        if (wid == 0 && out_lane)
          gimg[q * d.w + col] = ((lds_dd[0][lane] + lds_dd[1][lane]) + lds_dd[2][lane]) + lds_dd[3][lane];
        __syncthreads();
      } else if (MODE == 1) {
        if (out_lane) gimg[q * d.w + col] = dd;
      } else {
        if (out_lane) atomicAdd(gimg + q * d.w + col, dd);
      }
    }
  };

  int r = r0 - 2;
  const int rend = r1 + 1;      // inclusive
  while (r <= rend) {
    body(r, sA, sC, sB, hA, hC, hB, cA, cC, cB); ++r;
    if (r > rend) break;
    body(r, sB, sA, sC, hB, hA, hC, cB, cA, cC); ++r;
    if (r > rend) break;
    body(r, sC, sB, sA, hC, hB, hA, cC, cB, cA); ++r;
  }
  const long long gw = ((long long)(job.b * d.S + job.s) * d.CH + job.ck) * d.N + job.n;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const float v = wave_sum_all(dRt[i]);
    if (lane == 0) part[16 * gw + i] = v;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void fused_bwd_kernel(const float* __restrict__ src, const float* __restrict__ depth,
                                                        const float* __restrict__ T, const float* __restrict__ K,
                                                        const float* __restrict__ target,
                                                        const float* __restrict__ g_l1, const float* __restrict__ g_ssim,
                                                        float* __restrict__ ddepth, float* __restrict__ part,
                                                        FusedDims d, float inv_count) {
  fused_bwd_body<MODE>(src, depth, T, K, target, g_l1, g_ssim, ddepth, part, d, inv_count, blockIdx.x);
}

template <int MODE>
__global__ __launch_bounds__(256) void fused_bwd_ms_kernel(MsArgs m, const float* __restrict__ T, const float* __restrict__ K,
                                                           float* __restrict__ part) {
  const int s = ms_scale_of(m, blockIdx.x);
  fused_bwd_body<MODE>(m.src[s], m.depth[s], T, K, m.target[s], m.g_l1[s], m.g_ss[s], m.ddepth[s], part + m.part_off[s],
                       m.d[s], m.inv_count[s], blockIdx.x - m.block_off[s]);
}

// multi-scale finishers: grid.y = scale (forward); the pose gradient adds the scales in order
__global__ void fused_reduce_ms_kernel(MsArgs m, const float* __restrict__ part, float* __restrict__ losses) {
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

__global__ void fused_bwd_reduce_ms_kernel(MsArgs m, int nscales, const float* __restrict__ part, float* __restrict__ dT, int BN,
                                           int N) {
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
}

// dT[b,n] (4x4, last row 0) = sum over the (strip, chunk) waves of view (b,n), fixed order; 16 threads per entry.
__global__ void fused_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dT, int BN, int N,
                                        int waves_per_b) {
  const int e = blockIdx.x * 16 + (threadIdx.x >> 4);      // entry in [0, BN*16)
  const int t = threadIdx.x & 15;
  float sum = 0.f;
  const int bn = e / 16, i = e % 16;
  if (bn < BN && i < 12) {
    const int b = bn / N, n = bn % N;
    const int per_view = waves_per_b / N;
    for (int k = t; k < per_view; k += 16) sum += part[16 * ((long long)b * waves_per_b + (long long)k * N + n) + i];
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) sum += __shfl_down(sum, off, 16);
  if (t == 0 && bn < BN) dT[e] = sum;
}

__global__ void zero_fill_kernel(float* __restrict__ p, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

inline FusedDims make_dims(int B, int N, int h, int w, float scale, int rows_per_chunk, int strip) {
  FusedDims d;
  d.B = B; d.N = N; d.h = h; d.w = w; d.scale = scale;
  d.lds_ok = 0;
  d.dbg = 0;
  d.S = (w + strip - 1) / strip;
  d.R = rows_per_chunk < h ? rows_per_chunk : h;
  d.CH = (h + d.R - 1) / d.R;
  return d;
}

int g_fwd_min_waves = 4096, g_bwd_min_waves = 1536, g_min_rows = 8, g_fwd_pipe = 1, g_fwd_dbg = 0;

// Rows per chunk: enough waves to cover 256 CUs x 4 SIMDs a few times over, as few halo rows as possible.
inline int pick_rows(int B, int N, int h, int w, long long min_waves) {
  const long long strips = (long long)B * N * ((w + STRIP_B - 1) / STRIP_B);
  int R = 32;
  while (R > g_min_rows && strips * ((h + R - 1) / R) < min_waves) R >>= 1;
  return R;
}

}  // namespace

extern "C" {

int xpt_photo_fused_tune(int fwd_min_waves, int bwd_min_waves, int min_rows) {
  if (fwd_min_waves < 1 || bwd_min_waves < 1 || fwd_min_waves > 16384 || bwd_min_waves > 16384) return XPT_ERR_ARG;
  if (min_rows != 2 && min_rows != 4 && min_rows != 8 && min_rows != 16 && min_rows != 32) return XPT_ERR_ARG;
  g_fwd_min_waves = fwd_min_waves;
  g_bwd_min_waves = bwd_min_waves;
  g_min_rows = min_rows;
  return XPT_OK;
}

/* 2: blocks of rows with the neighbour texels staged in LDS; 1: the hand-pipelined march (asm loads, explicit
 * s_waitcnt); 0: the compiler-scheduled march.  Same results. */
int xpt_photo_fused_variant(int pipelined) {
  if (pipelined == 10) {                           // diagnostic: the compiler-scheduled march without its neighbour loads
    g_fwd_pipe = 0;
    g_fwd_dbg = 1;
    return XPT_OK;
  }
  if (pipelined < 0 || pipelined > 2) return XPT_ERR_ARG;
  g_fwd_dbg = 0;
  g_fwd_pipe = pipelined;
  return XPT_OK;
}

size_t xpt_photo_fused_workspace_floats(int B, int N, int h, int w) {
  if (B <= 0 || N <= 0 || h <= 0 || w <= 0) return 0;
  const FusedDims d = make_dims(B, N, h, w, 1.f, 2, STRIP_B);   // upper bound of both directions (smallest row chunk)
  return (size_t)d.B * d.S * d.CH * d.N * 16;       // per wave: 2 floats forward, 12 (pose gradient) backward
}

int xpt_photo_fused_fwd(const float* src, const float* depth, const float* T, const float* K, const float* target,
                        float* synth, float* loss_l1, float* loss_ssim, float* workspace, size_t workspace_floats,
                        int B, int N, int h, int w, float scale, void* stream) {
  XPT_CHECK_PTR(src); XPT_CHECK_PTR(depth); XPT_CHECK_PTR(T); XPT_CHECK_PTR(K); XPT_CHECK_PTR(target);
  XPT_CHECK_PTR(workspace);
  if ((loss_l1 == nullptr) != (loss_ssim == nullptr)) return XPT_ERR_NULL;     // both, or neither (partials only)
  if (B <= 0 || N <= 0 || h <= 0 || w <= 0 || !(scale > 0.f) || (long long)h * w * 3 >= (1LL << 31)) return XPT_ERR_SHAPE;
  if (workspace_floats < xpt_photo_fused_workspace_floats(B, N, h, w)) return XPT_ERR_WORKSPACE;
  FusedDims d = make_dims(B, N, h, w, scale, pick_rows(B, N, h, w, g_fwd_min_waves), STRIP);
  d.lds_ok = (w % 4 == 0) && (((uintptr_t)src) % 16 == 0);        // rows of 3 w floats and images of 3 h w floats stay aligned
  d.dbg = g_fwd_dbg;
  const long long nwaves = (long long)d.B * d.S * d.CH * d.N;
  const unsigned blocks = (unsigned)((nwaves + 3) / 4);
  hipStream_t s = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
  if (!synth && g_fwd_pipe == 2)
    hipLaunchKernelGGL(fused_fwd_lds_kernel, dim3(blocks), dim3(256), 0, s, src, depth, T, K, target, workspace, d);
  else if (synth)
    hipLaunchKernelGGL((fused_fwd_kernel<true, false>), dim3(blocks), dim3(256), 0, s, src, depth, T, K, target, synth,
                       workspace, d);
  else if (g_fwd_pipe)
    hipLaunchKernelGGL((fused_fwd_kernel<false, true>), dim3(blocks), dim3(256), 0, s, src, depth, T, K, target, synth,
                       workspace, d);
  else
    hipLaunchKernelGGL((fused_fwd_kernel<false, false>), dim3(blocks), dim3(256), 0, s, src, depth, T, K, target, synth,
                       workspace, d);
  if (loss_l1)
    hipLaunchKernelGGL(fused_reduce_kernel, dim3(B), dim3(64), 0, s, workspace, loss_l1, loss_ssim, d.S * d.CH * d.N,
                       1.0f / ((float)N * (float)h * (float)w * 3.0f));
  return xpt_launch_status();
}

int xpt_photo_fused_bwd(const float* src, const float* depth, const float* T, const float* K, const float* target,
                        const float* g_l1, const float* g_ssim, float* ddepth, float* dT, float* workspace,
                        size_t workspace_floats, int B, int N, int h, int w, float scale, void* stream) {
  XPT_CHECK_PTR(src); XPT_CHECK_PTR(depth); XPT_CHECK_PTR(T); XPT_CHECK_PTR(K); XPT_CHECK_PTR(target);
  XPT_CHECK_PTR(g_l1); XPT_CHECK_PTR(g_ssim); XPT_CHECK_PTR(ddepth); XPT_CHECK_PTR(dT); XPT_CHECK_PTR(workspace);
  if (B <= 0 || N <= 0 || h <= 0 || w <= 0 || !(scale > 0.f) || (long long)h * w * 3 >= (1LL << 31)) return XPT_ERR_SHAPE;
  if (workspace_floats < xpt_photo_fused_workspace_floats(B, N, h, w)) return XPT_ERR_WORKSPACE;
  const FusedDims d = make_dims(B, N, h, w, scale, pick_rows(B, N, h, w, g_bwd_min_waves), STRIP_B);
  const long long nwaves = (long long)d.B * d.S * d.CH * d.N;
  const unsigned blocks = (unsigned)((nwaves + 3) / 4);
  const float inv_count = 1.0f / ((float)N * (float)h * (float)w * 3.0f);
  hipStream_t s = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
  if (N == 4) {
    hipLaunchKernelGGL(fused_bwd_kernel<0>, dim3(blocks), dim3(256), 0, s, src, depth, T, K, target, g_l1, g_ssim, ddepth,
                       workspace, d, inv_count);
  } else if (N == 1) {
    hipLaunchKernelGGL(fused_bwd_kernel<1>, dim3(blocks), dim3(256), 0, s, src, depth, T, K, target, g_l1, g_ssim, ddepth,
                       workspace, d, inv_count);
  } else {
    // (a zero-fill KERNEL, not hipMemsetAsync: memset nodes of a captured hipGraph write garbage from the second replay
    // on with this runtime -- tools/replay_probe_memset.py)
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)(((long long)B * h * w + 255) / 256)), dim3(256), 0, s, ddepth,
                       (long long)B * h * w);
    hipLaunchKernelGGL(fused_bwd_kernel<2>, dim3(blocks), dim3(256), 0, s, src, depth, T, K, target, g_l1, g_ssim, ddepth,
                       workspace, d, inv_count);
  }
  hipLaunchKernelGGL(fused_bwd_reduce_kernel, dim3((B * N * 16 + 15) / 16), dim3(256), 0, s, workspace, dT, B * N, N,
                     d.S * d.CH * d.N);
  return xpt_launch_status();
}

/* ---- all scales of the pyramid in one launch (same arithmetic per scale as the calls above).
 * Arrays of nscales (<= 4) entries: src[s] [B,N,h_s,w_s,3], depth[s] [B,h_s,w_s], target[s] [B,h_s,w_s,3], scale[s]
 * (intrinsic divisor).  losses [2 nscales][B]: row s = photometric L1 of scale s, row nscales + s = its SSIM loss.
 * Workspace: the sum of the per-scale xpt_photo_fused_workspace_floats. */
int xpt_photo_fused_ms_fwd(int nscales, const float* const* src, const float* const* depth, const float* T, const float* K,
                           const float* const* target, float* losses, float* workspace, size_t workspace_floats,
                           int B, int N, const int* h, const int* w, const float* scale, void* stream) {
  XPT_CHECK_PTR(src); XPT_CHECK_PTR(depth); XPT_CHECK_PTR(T); XPT_CHECK_PTR(K); XPT_CHECK_PTR(target);
  XPT_CHECK_PTR(workspace); XPT_CHECK_PTR(h); XPT_CHECK_PTR(w); XPT_CHECK_PTR(scale);     // losses == NULL: partials only
  if (nscales < 1 || nscales > 4 || B <= 0 || N <= 0) return XPT_ERR_ARG;
  MsArgs m{};
  size_t need = 0;
  unsigned blocks = 0;
  for (int s = 0; s < nscales; ++s) {
    if (!src[s] || !depth[s] || !target[s]) return XPT_ERR_NULL;
    if (h[s] <= 0 || w[s] <= 0 || !(scale[s] > 0.f) || (long long)h[s] * w[s] * 3 >= (1LL << 31)) return XPT_ERR_SHAPE;
    FusedDims d = make_dims(B, N, h[s], w[s], scale[s], pick_rows(B, N, h[s], w[s], g_fwd_min_waves), STRIP);
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
  hipLaunchKernelGGL(fused_fwd_ms_kernel, dim3(blocks), dim3(256), 0, st, m, T, K, workspace);
  if (losses) hipLaunchKernelGGL(fused_reduce_ms_kernel, dim3(B, nscales), dim3(64), 0, st, m, workspace, losses);
  return xpt_launch_status();
}

/* backward of the above: g_l1[s] / g_ssim[s] [B] -> ddepth[s] [B,h_s,w_s] and dT [B,N,4,4] = the sum over the scales
 * (added in scale order).  N must be 4 or 1 (the modes that write d_depth without atomics). */
int xpt_photo_fused_ms_bwd(int nscales, const float* const* src, const float* const* depth, const float* T, const float* K,
                           const float* const* target, const float* const* g_l1, const float* const* g_ssim,
                           float* const* ddepth, float* dT, float* workspace, size_t workspace_floats, int B, int N,
                           const int* h, const int* w, const float* scale, void* stream) {
  XPT_CHECK_PTR(src); XPT_CHECK_PTR(depth); XPT_CHECK_PTR(T); XPT_CHECK_PTR(K); XPT_CHECK_PTR(target); XPT_CHECK_PTR(g_l1);
  XPT_CHECK_PTR(g_ssim); XPT_CHECK_PTR(ddepth); XPT_CHECK_PTR(dT); XPT_CHECK_PTR(workspace); XPT_CHECK_PTR(h);
  XPT_CHECK_PTR(w); XPT_CHECK_PTR(scale);
  if (nscales < 1 || nscales > 4 || B <= 0 || (N != 4 && N != 1)) return XPT_ERR_ARG;
  MsArgs m{};
  size_t need = 0;
  unsigned blocks = 0;
  for (int s = 0; s < nscales; ++s) {
    if (!src[s] || !depth[s] || !target[s] || !g_l1[s] || !g_ssim[s] || !ddepth[s]) return XPT_ERR_NULL;
    if (h[s] <= 0 || w[s] <= 0 || !(scale[s] > 0.f) || (long long)h[s] * w[s] * 3 >= (1LL << 31)) return XPT_ERR_SHAPE;
    FusedDims d = make_dims(B, N, h[s], w[s], scale[s], pick_rows(B, N, h[s], w[s], g_bwd_min_waves), STRIP_B);
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
  }
  for (int s = nscales; s <= 4; ++s) m.block_off[s] = blocks;
  if (workspace_floats < need) return XPT_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
  if (N == 4) hipLaunchKernelGGL(fused_bwd_ms_kernel<0>, dim3(blocks), dim3(256), 0, st, m, T, K, workspace);
  else hipLaunchKernelGGL(fused_bwd_ms_kernel<1>, dim3(blocks), dim3(256), 0, st, m, T, K, workspace);
  hipLaunchKernelGGL(fused_bwd_reduce_ms_kernel, dim3((B * N * 16 + 15) / 16), dim3(256), 0, st, m, nscales, workspace, dT,
                     B * N, N);
  return xpt_launch_status();
}

}  // extern "C"
