// xpt_warp.hip -- pose algebra (K0), image pyramid (K1), view synthesis (K2+K3) and the
// stand-alone bilinear sampler (K3) for gfx950.  C ABI: include/xpt_hip.h.
#include "xpt_common.h"

using namespace xpt;

// =================================================================== K0: twist -> matrix
// pose_rvec2matr_batch_tf, utils/convert_pose.py:32-71.  W = -[w]x (":56"), w = u/|u|,
// R = I + sin(t) W + (1-cos(t)) W W ; where(|t| < 1e-8, I, R).
__device__ inline void skew_neg(const float w[3], float W[9]) {
  W[0] = 0.f;   W[1] = w[2];  W[2] = -w[1];
  W[3] = -w[2]; W[4] = 0.f;   W[5] = w[0];
  W[6] = w[1];  W[7] = -w[0]; W[8] = 0.f;
}

__device__ inline void matmul3(const float* A, const float* B, float* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

__global__ void pose_fwd_kernel(const float* __restrict__ pose, float* __restrict__ T, int M) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float* p = pose + 6 * m;
  const float u[3] = {p[3], p[4], p[5]};
  const float th = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  const float w[3] = {u[0] / th, u[1] / th, u[2] / th};
  float W[9], W2[9], R[9];
  skew_neg(w, W);
  matmul3(W, W, W2);
  const float s = sinf(th), c = 1.f - cosf(th);
  const bool ident = fabsf(th) < 1e-8f;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const float id = (i % 4 == 0) ? 1.f : 0.f;
    R[i] = ident ? id : (id + W[i] * s + W2[i] * c);
  }
  float* o = T + 16 * m;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o[4 * i + 0] = R[3 * i + 0];
    o[4 * i + 1] = R[3 * i + 1];
    o[4 * i + 2] = R[3 * i + 2];
    o[4 * i + 3] = p[i];
  }
  o[12] = 0.f; o[13] = 0.f; o[14] = 0.f; o[15] = 1.f;
}

__global__ void pose_bwd_kernel(const float* __restrict__ pose, const float* __restrict__ dT,
                                float* __restrict__ dpose, int M) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float* p = pose + 6 * m;
  const float* g4 = dT + 16 * m;
  float* o = dpose + 6 * m;
  o[0] = g4[3]; o[1] = g4[7]; o[2] = g4[11];
  const float u[3] = {p[3], p[4], p[5]};
  const float th = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  if (fabsf(th) < 1e-8f) {  // where() selects the constant identity: no gradient
    o[3] = 0.f; o[4] = 0.f; o[5] = 0.f;
    return;
  }
  const float w[3] = {u[0] / th, u[1] / th, u[2] / th};
  float W[9], W2[9], G[9];
  skew_neg(w, W);
  matmul3(W, W, W2);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) G[3 * i + j] = g4[4 * i + j];
  const float s = sinf(th), cs = cosf(th), c = 1.f - cs;
  float dLds = 0.f, dLdc = 0.f;
#pragma unroll
  for (int i = 0; i < 9; ++i) { dLds += G[i] * W[i]; dLdc += G[i] * W2[i]; }
  const float dLdth = dLds * cs + dLdc * s;
  // dL/dW = s G + c (G W^T + W^T G)
  float Wt[9], GWt[9], WtG[9], dW[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Wt[3 * i + j] = W[3 * j + i];
  matmul3(G, Wt, GWt);
  matmul3(Wt, G, WtG);
#pragma unroll
  for (int i = 0; i < 9; ++i) dW[i] = s * G[i] + c * (GWt[i] + WtG[i]);
  // W01 = w3, W02 = -w2, W10 = -w3, W12 = w1, W20 = w2, W21 = -w1
  const float dw[3] = {dW[5] - dW[7], dW[6] - dW[2], dW[1] - dW[3]};
  const float wd = w[0] * dw[0] + w[1] * dw[1] + w[2] * dw[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) o[3 + i] = (dw[i] - w[i] * wd) / th + w[i] * dLdth;
}

// =================================================================== K1: half-pixel bilinear down-scale
// tf.image.resize(bilinear) at an exact integer factor s: the source coordinate of output i is
// s*i + s/2 - 0.5, i.e. taps s*i+s/2-1 and s*i+s/2 with lerp 0.5 on each axis (TF lerp order:
// top = tl + (tr-tl)*xl ; bottom = bl + (br-bl)*xl ; out = top + (bottom-top)*yl).
__global__ void resize_down_kernel(const float* __restrict__ img, float* __restrict__ out, int M, int H, int W,
                                   int C, int s) {
  const int h = H / s, w = W / s;
  const long long total = (long long)M * h * w * C;
  const long long rowC = (long long)W * C;
  const int a = s / 2 - 1;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h);
    const int m = (int)(r / h);
    if (s == 1) { out[i] = img[i]; continue; }
    const float* base = img + ((long long)m * H + (y * s + a)) * rowC + (long long)(x * s + a) * C + c;
    const float tl = base[0], tr = base[C], bl = base[rowC], br = base[rowC + C];
    const float top = tl + (tr - tl) * 0.5f;
    const float bot = bl + (br - bl) * 0.5f;
    out[i] = top + (bot - top) * 0.5f;
  }
}

// Every image the loss stage reads, in one launch: the dense copies of the source frames and of the target frame of a
// snippet batch image5d [B,S,H,W,3] (losses.py:57-66: TARGET FRAME LAST) and their resize_down pyramids.  A job =
// (frames f0 .. f0+nf-1 of every snippet, factor s) -> out [B*nf, H/s, W/s, 3]; same arithmetic as resize_down_kernel.
struct PyramidJob {
  float* out;
  long long total;     // outputs (float4 units when s == 1 and vec)
  int f0, nf, s, block_off;
};
struct PyramidArgs {
  PyramidJob job[10];
  const float* img;
  int njobs, S, H, W, vec;
};

__global__ void pyramid_kernel(PyramidArgs a) {
  int j = 0;
  while (j + 1 < a.njobs && (int)blockIdx.x >= a.job[j + 1].block_off) ++j;
  const PyramidJob job = a.job[j];
  const int nblk = (j + 1 < a.njobs ? a.job[j + 1].block_off : (int)gridDim.x) - job.block_off;
  const long long first = (long long)(blockIdx.x - job.block_off) * blockDim.x + threadIdx.x;
  const long long step = (long long)nblk * blockDim.x;
  const int H = a.H, W = a.W, s = job.s;
  if (s == 1) {
    if (a.vec) {
      const long long F4 = (long long)H * W * 3 / 4;
      const float4* in4 = (const float4*)a.img;
      float4* out4 = (float4*)job.out;
      // (flat indices below 2^31 -- checked by the launcher --: split without the 64-bit divisions, ~120 instructions each)
      for (long long i = first; i < job.total; i += step) {
        unsigned r_, f_;
        const long long m = (long long)xpt_divmod((unsigned)i, (unsigned)F4, r_), r = (long long)r_;
        const long long b = (long long)xpt_divmod((unsigned)m, (unsigned)job.nf, f_), f = (long long)f_ + job.f0;
        out4[i] = in4[(b * a.S + f) * F4 + r];
      }
    } else {
      const long long F = (long long)H * W * 3;
      for (long long i = first; i < job.total; i += step) {
        unsigned r_, f_;
        const long long m = (long long)xpt_divmod((unsigned)i, (unsigned)F, r_), r = (long long)r_;
        const long long b = (long long)xpt_divmod((unsigned)m, (unsigned)job.nf, f_), f = (long long)f_ + job.f0;
        job.out[i] = a.img[(b * a.S + f) * F + r];
      }
    }
    return;
  }
  const int h = H / s, w = W / s;
  const long long rowC = (long long)W * 3;
  const int t = s / 2 - 1;
  for (long long i = first; i < job.total; i += step) {
    unsigned c_, x_, y_, f_;
    unsigned q = xpt_divmod((unsigned)i, 3u, c_);
    q = xpt_divmod(q, (unsigned)w, x_);
    const unsigned m = xpt_divmod(q, (unsigned)h, y_);
    const long long b = (long long)xpt_divmod(m, (unsigned)job.nf, f_), f = (long long)f_ + job.f0;
    const int c = (int)c_, x = (int)x_, y = (int)y_;
    const float* base = a.img + ((b * a.S + f) * H + (y * s + t)) * rowC + (long long)(x * s + t) * 3 + c;
    const float tl = base[0], tr = base[3], bl = base[rowC], br = base[rowC + 3];
    const float top = tl + (tr - tl) * 0.5f;
    const float bot = bl + (br - bl) * 0.5f;
    job.out[i] = top + (bot - top) * 0.5f;
  }
}

// =================================================================== K2+K3: view synthesis (per-pixel form)
// One thread per target pixel, looping over the N source views (depth / ray shared).
__device__ inline void sample3(const float* __restrict__ img, int w, const Taps& t, float out[3], float tap[12]) {
  const float* pff = img + ((long long)t.vf * w + t.uf) * 3;
  const float* pfc = img + ((long long)t.vc * w + t.uf) * 3;  // (v=vc, u=uf)
  const float* pcf = img + ((long long)t.vf * w + t.uc) * 3;  // (v=vf, u=uc)
  const float* pcc = img + ((long long)t.vc * w + t.uc) * 3;
  const float wff = t.wuf * t.wvf, wfc = t.wuf * t.wvc, wcf = t.wuc * t.wvf, wcc = t.wuc * t.wvc;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    tap[c] = pff[c]; tap[3 + c] = pfc[c]; tap[6 + c] = pcf[c]; tap[9 + c] = pcc[c];
    // merge_images (bilinear_interp.py:134-146): sum over the 4 neighbours in (ff, fc, cf, cc) order
    out[c] = ((tap[c] * wff + tap[3 + c] * wfc) + tap[6 + c] * wcf) + tap[9 + c] * wcc;
  }
}

__global__ void warp_fwd_kernel(const float* __restrict__ src, const float* __restrict__ depth,
                                const float* __restrict__ T, const float* __restrict__ K,
                                float* __restrict__ synth, int N, int h, int w, float scale) {
  const int b = blockIdx.y;
  const int P = h * w;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const Cam cam = load_cam(K + 9 * b, scale);
  const float d = depth[(long long)b * P + p];
  const int v = p / w, u = p - v * w;
  Warp wp;
  backproject(cam, (float)u, (float)v, d, wp);
  for (int n = 0; n < N; ++n) {
    const Pose pose = load_pose(T + 16 * ((long long)b * N + n));
    project(cam, pose, wp);
    const Taps t = make_taps(wp.up, wp.vp, h, w, d != 0.f);
    float out[3], tap[12];
    sample3(src + ((long long)b * N + n) * P * 3, w, t, out, tap);
    float* o = synth + (((long long)b * N + n) * P + p) * 3;
    o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
  }
}

// Backward of one (pixel, view): dsynth g[3] -> d(depth), d(R,t) contributions.
__device__ inline void warp_pixel_bwd(const Cam& cam, const Pose& pose, const Warp& wp, const Taps& t,
                                      const float tap[12], const float g[3], float& dd, float dRt[12]) {
  // d out_c / du' = mask * [ (I_cf - I_ff) wvf + (I_cc - I_fc) wvc ],  d/dv' likewise
  float du = 0.f, dv = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    du += g[c] * ((tap[6 + c] - tap[c]) * t.wvf + (tap[9 + c] - tap[3 + c]) * t.wvc);
    dv += g[c] * ((tap[3 + c] - tap[c]) * t.wuf + (tap[9 + c] - tap[6 + c]) * t.wuc);
  }
  du *= t.mask;
  dv *= t.mask;
  // (u',v') = (q0,q1)/(q2+eps)
  const float dq0 = du * wp.zinv, dq1 = dv * wp.zinv;
  const float dq2 = -(du * wp.up + dv * wp.vp) * wp.zinv;
  // q = K X'  ->  dX' = K^T dq
  float dXs[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) dXs[j] = cam.k[j] * dq0 + cam.k[3 + j] * dq1 + cam.k[6 + j] * dq2;
  // X' = R X + t
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    dRt[4 * i + 0] = dXs[i] * wp.X[0];
    dRt[4 * i + 1] = dXs[i] * wp.X[1];
    dRt[4 * i + 2] = dXs[i] * wp.X[2];
    dRt[4 * i + 3] = dXs[i];
  }
  // dX = R^T dX' ; X = d * ray -> dd = dX . ray
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float dXj = pose.r[j] * dXs[0] + pose.r[3 + j] * dXs[1] + pose.r[6 + j] * dXs[2];
    acc += dXj * wp.ray[j];
  }
  dd = acc;
}

// workspace layout: part[b][n][blk][12]
__global__ void warp_bwd_kernel(const float* __restrict__ src, const float* __restrict__ depth,
                                const float* __restrict__ T, const float* __restrict__ K,
                                const float* __restrict__ dsynth, float* __restrict__ ddepth,
                                float* __restrict__ part, int N, int h, int w, float scale) {
  __shared__ float red[4 * 12];
  const int b = blockIdx.y;
  const int P = h * w;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = p < P;
  const Cam cam = load_cam(K + 9 * b, scale);
  const float d = live ? depth[(long long)b * P + p] : 0.f;
  const int v = live ? p / w : 0, u = live ? p - v * w : 0;
  Warp wp;
  backproject(cam, (float)u, (float)v, d, wp);
  float dd_total = 0.f;
  for (int n = 0; n < N; ++n) {
    const Pose pose = load_pose(T + 16 * ((long long)b * N + n));
    float dRt[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) dRt[i] = 0.f;
    if (live) {
      project(cam, pose, wp);
      const Taps t = make_taps(wp.up, wp.vp, h, w, d != 0.f);
      float out[3], tap[12];
      sample3(src + ((long long)b * N + n) * P * 3, w, t, out, tap);
      const float* gp = dsynth + (((long long)b * N + n) * P + p) * 3;
      const float g[3] = {gp[0], gp[1], gp[2]};
      float dd;
      warp_pixel_bwd(cam, pose, wp, t, tap, g, dd, dRt);
      dd_total += dd;
    }
    block_sum_n<12>(dRt, red);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < 12; ++i) part[(((long long)b * N + n) * gridDim.x + blockIdx.x) * 12 + i] = dRt[i];
    }
  }
  if (live) ddepth[(long long)b * P + p] = dd_total;
}

// dT[bn][4x4] = sum_blk part[bn][blk][12] in fixed order; last row 0.
__global__ void warp_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dT, int BN, int nblk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BN * 16) return;
  const int bn = i / 16, e = i % 16;
  float s = 0.f;
  if (e < 12) {
    const float* q = part + (long long)bn * nblk * 12 + e;
    for (int k = 0; k < nblk; ++k) s += q[(long long)k * 12];
  }
  dT[i] = s;
}

// =================================================================== K3 alone: sampler with explicit coords
#define XPT_MAX_C 16
__global__ void bilinear_fwd_kernel(const float* __restrict__ image, const float* __restrict__ coords,
                                    const float* __restrict__ vmask, float* __restrict__ out, int N, int h, int w,
                                    int C, int ncoord) {
  const int bn = blockIdx.y;
  const int b = bn / N;
  const int P = h * w;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float* cb = coords + (long long)bn * ncoord * P;
  const float u = cb[p], v = cb[P + p];
  const bool ok = vmask ? (vmask[(long long)b * P + p] != 0.f) : true;
  const Taps t = make_taps(u, v, h, w, ok);
  const float* img = image + (long long)bn * P * C;
  const float* pff = img + ((long long)t.vf * w + t.uf) * C;
  const float* pfc = img + ((long long)t.vc * w + t.uf) * C;
  const float* pcf = img + ((long long)t.vf * w + t.uc) * C;
  const float* pcc = img + ((long long)t.vc * w + t.uc) * C;
  const float wff = t.wuf * t.wvf, wfc = t.wuf * t.wvc, wcf = t.wuc * t.wvf, wcc = t.wuc * t.wvc;
  float* o = out + ((long long)bn * P + p) * C;
  for (int c = 0; c < C; ++c) o[c] = ((pff[c] * wff + pfc[c] * wfc) + pcf[c] * wcf) + pcc[c] * wcc;
}

__global__ void bilinear_bwd_kernel(const float* __restrict__ image, const float* __restrict__ coords,
                                    const float* __restrict__ vmask, const float* __restrict__ dout,
                                    float* __restrict__ dcoords, int N, int h, int w, int C, int ncoord) {
  const int bn = blockIdx.y;
  const int b = bn / N;
  const int P = h * w;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float* cb = coords + (long long)bn * ncoord * P;
  const float u = cb[p], v = cb[P + p];
  const bool ok = vmask ? (vmask[(long long)b * P + p] != 0.f) : true;
  const Taps t = make_taps(u, v, h, w, ok);
  const float* img = image + (long long)bn * P * C;
  const float* pff = img + ((long long)t.vf * w + t.uf) * C;
  const float* pfc = img + ((long long)t.vc * w + t.uf) * C;
  const float* pcf = img + ((long long)t.vf * w + t.uc) * C;
  const float* pcc = img + ((long long)t.vc * w + t.uc) * C;
  const float* g = dout + ((long long)bn * P + p) * C;
  float du = 0.f, dv = 0.f;
  for (int c = 0; c < C; ++c) {
    du += g[c] * ((pcf[c] - pff[c]) * t.wvf + (pcc[c] - pfc[c]) * t.wvc);
    dv += g[c] * ((pfc[c] - pff[c]) * t.wuf + (pcc[c] - pcf[c]) * t.wuc);
  }
  float* dc = dcoords + (long long)bn * ncoord * P;
  dc[p] = du * t.mask;
  dc[P + p] = dv * t.mask;
  if (ncoord == 3) dc[2 * (long long)P + p] = 0.f;
}

// =================================================================== C ABI
extern "C" {

int xpt_abi_version(void) { return 1; }
const char* xpt_build_arch(void) { return "gfx950"; }

int xpt_pose_rvec2matr_fwd(const float* pose, float* T, int M, void* stream) {
  XPT_CHECK_PTR(pose); XPT_CHECK_PTR(T);
  if (M <= 0) return XPT_ERR_SHAPE;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(pose_fwd_kernel, dim3((M + 63) / 64), dim3(64), 0, (hipStream_t)stream, pose, T, M);
  return xpt_launch_status();
}

int xpt_pose_rvec2matr_bwd(const float* pose, const float* dT, float* dpose, int M, void* stream) {
  XPT_CHECK_PTR(pose); XPT_CHECK_PTR(dT); XPT_CHECK_PTR(dpose);
  if (M <= 0) return XPT_ERR_SHAPE;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(pose_bwd_kernel, dim3((M + 63) / 64), dim3(64), 0, (hipStream_t)stream, pose, dT, dpose, M);
  return xpt_launch_status();
}

int xpt_resize_down_fwd(const float* img, float* out, int M, int H, int W, int C, int scale, void* stream) {
  XPT_CHECK_PTR(img); XPT_CHECK_PTR(out);
  if (M <= 0 || H <= 0 || W <= 0 || C <= 0 || scale <= 0) return XPT_ERR_SHAPE;
  if ((scale != 1 && (scale & 1)) || H % scale || W % scale) return XPT_ERR_SHAPE;
  const long long total = (long long)M * (H / scale) * (W / scale) * C;
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(resize_down_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, img, out, M, H, W,
                     C, scale);
  return xpt_launch_status();
}

int xpt_image_pyramids(const float* image5d, int B, int S, int H, int W, int njobs, const int* first_frame,
                       const int* nframes, const int* scale, float* const* out, void* stream) {
  XPT_CHECK_PTR(image5d); XPT_CHECK_PTR(first_frame); XPT_CHECK_PTR(nframes); XPT_CHECK_PTR(scale); XPT_CHECK_PTR(out);
  if (B <= 0 || S <= 0 || H <= 0 || W <= 0 || njobs < 1 || njobs > 10) return XPT_ERR_SHAPE;
  PyramidArgs a = {};
  a.img = image5d; a.njobs = njobs; a.S = S; a.H = H; a.W = W;
  a.vec = ((long long)H * W * 3) % 4 == 0 && ((uintptr_t)image5d) % 16 == 0;
  for (int j = 0; j < njobs; ++j) {
    if (!out[j]) return XPT_ERR_NULL;
    if (scale[j] == 1 && ((uintptr_t)out[j]) % 16 != 0) a.vec = 0;
  }
  long long blocks = 0;
  for (int j = 0; j < njobs; ++j) {
    const int s = scale[j];
    if (s <= 0 || (s != 1 && (s & 1)) || H % s || W % s) return XPT_ERR_SHAPE;
    if (first_frame[j] < 0 || nframes[j] <= 0 || first_frame[j] + nframes[j] > S) return XPT_ERR_SHAPE;
    PyramidJob& job = a.job[j];
    job.out = out[j]; job.f0 = first_frame[j]; job.nf = nframes[j]; job.s = s;
    job.total = (long long)B * nframes[j] * (H / s) * (W / s) * 3;
    if (job.total >= (1LL << 31)) return XPT_ERR_SHAPE;                 // (the kernel splits flat indices in 32 bits)
    if (s == 1 && a.vec) job.total /= 4;
    long long nb = (job.total + 255) / 256;
    if (nb > 2048) nb = 2048;
    job.block_off = (int)blocks;
    blocks += nb;
  }
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(pyramid_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return xpt_launch_status();
}

int xpt_warp_fwd(const float* src, const float* depth, const float* T, const float* K, float* synth, int B, int N,
                 int h, int w, float scale, void* stream) {
  XPT_CHECK_PTR(src); XPT_CHECK_PTR(depth); XPT_CHECK_PTR(T); XPT_CHECK_PTR(K); XPT_CHECK_PTR(synth);
  if (B <= 0 || N <= 0 || h <= 0 || w <= 0 || B > 65535 || !(scale > 0.f)) return XPT_ERR_SHAPE;
  const int P = h * w;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(warp_fwd_kernel, dim3((P + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, src, depth, T, K,
                     synth, N, h, w, scale);
  return xpt_launch_status();
}

size_t xpt_warp_bwd_workspace_floats(int B, int N, int h, int w) {
  if (B <= 0 || N <= 0 || h <= 0 || w <= 0) return 0;
  const size_t nblk = ((size_t)h * w + 255) / 256;
  return (size_t)B * N * nblk * 12;
}

int xpt_warp_bwd(const float* src, const float* depth, const float* T, const float* K, const float* dsynth,
                 float* ddepth, float* dT, float* workspace, size_t workspace_floats, int B, int N, int h, int w,
                 float scale, void* stream) {
  XPT_CHECK_PTR(src); XPT_CHECK_PTR(depth); XPT_CHECK_PTR(T); XPT_CHECK_PTR(K); XPT_CHECK_PTR(dsynth);
  XPT_CHECK_PTR(ddepth); XPT_CHECK_PTR(dT); XPT_CHECK_PTR(workspace);
  if (B <= 0 || N <= 0 || h <= 0 || w <= 0 || B > 65535 || !(scale > 0.f)) return XPT_ERR_SHAPE;
  if (workspace_floats < xpt_warp_bwd_workspace_floats(B, N, h, w)) return XPT_ERR_WORKSPACE;
  const int P = h * w;
  const int nblk = (P + 255) / 256;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(warp_bwd_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, src, depth, T, K, dsynth,
                     ddepth, workspace, N, h, w, scale);
  hipLaunchKernelGGL(warp_bwd_reduce_kernel, dim3((B * N * 16 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     workspace, dT, B * N, nblk);
  return xpt_launch_status();
}

int xpt_bilinear_fwd(const float* image, const float* coords, const float* valid_mask, float* out, int B, int N,
                     int h, int w, int C, int ncoord, void* stream) {
  XPT_CHECK_PTR(image); XPT_CHECK_PTR(coords); XPT_CHECK_PTR(out);
  if (B <= 0 || N <= 0 || h <= 0 || w <= 0 || C <= 0 || (long long)B * N > 65535) return XPT_ERR_SHAPE;
  if (ncoord != 2 && ncoord != 3) return XPT_ERR_ARG;
  const int P = h * w;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3((P + 255) / 256, B * N), dim3(256), 0, (hipStream_t)stream, image,
                     coords, valid_mask, out, N, h, w, C, ncoord);
  return xpt_launch_status();
}

int xpt_bilinear_bwd(const float* image, const float* coords, const float* valid_mask, const float* dout,
                     float* dcoords, int B, int N, int h, int w, int C, int ncoord, void* stream) {
  XPT_CHECK_PTR(image); XPT_CHECK_PTR(coords); XPT_CHECK_PTR(dout); XPT_CHECK_PTR(dcoords);
  if (B <= 0 || N <= 0 || h <= 0 || w <= 0 || C <= 0 || (long long)B * N > 65535) return XPT_ERR_SHAPE;
  if (ncoord != 2 && ncoord != 3) return XPT_ERR_ARG;
  const int P = h * w;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3((P + 255) / 256, B * N), dim3(256), 0, (hipStream_t)stream, image,
                     coords, valid_mask, dout, dcoords, N, h, w, C, ncoord);
  return xpt_launch_status();
}

}  // extern "C"
