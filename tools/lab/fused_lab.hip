// tools/lab/fused_lab.hip -- standalone C-ABI client of the fused warp + L1 + SSIM kernels (no torch): generates
// KITTI-shaped synthetic inputs (the statistics of hip/roofline.py's leg: smooth depth U(1,80), poses
// N(0, diag(0.3, 0.05, 1.0 m, 0.01, 0.02, 0.01 rad))), runs the reference launch (xpt_photo_fused_ms_*) and the
// candidate (xpt_photo_march_ms_*), compares losses / gradients and times both with HIP events.
// Build (both kernel files compiled into the binary):
//   hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -std=c++17 -o tools/lab/bin/fused_lab tools/lab/fused_lab.hip \
//         xpt_mde_2021_amd/csrc/xpt_fused.hip xpt_mde_2021_amd/csrc/xpt_march.hip
// Usage: fused_lab [B=8] [H=128] [W=416] [reps=20] [mode: both|old|new] [scales=4]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../include/xpt_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static std::vector<float> smooth_field(int B, int H, int W, int C, int cutoff, std::mt19937& g, float lo, float hi, float detail) {
  const int gh = std::max(H / cutoff, 2), gw = std::max(W / cutoff, 2);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<float> out((size_t)B * H * W * C);
  std::vector<float> coarse((size_t)gh * gw);
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      for (auto& v : coarse) v = U(g);
      for (int y = 0; y < H; ++y) {
        const float fy = (float)y * (gh - 1) / (H - 1);
        const int y0 = std::min((int)fy, gh - 2);
        const float wy = fy - y0;
        for (int x = 0; x < W; ++x) {
          const float fx = (float)x * (gw - 1) / (W - 1);
          const int x0 = std::min((int)fx, gw - 2);
          const float wx = fx - x0;
          // smoothstep weights: C1-continuous like the bicubic field of synthetic_data.smooth_noise
          const float sx = wx * wx * (3 - 2 * wx), sy = wy * wy * (3 - 2 * wy);
          float v = (coarse[y0 * gw + x0] * (1 - sx) + coarse[y0 * gw + x0 + 1] * sx) * (1 - sy) +
                    (coarse[(y0 + 1) * gw + x0] * (1 - sx) + coarse[(y0 + 1) * gw + x0 + 1] * sx) * sy;
          v = v * 0.9f + detail * U(g);
          v = std::min(std::max(v, -1.f), 1.f);
          out[(((size_t)b * H + y) * W + x) * C + c] = lo + (v + 1.f) * 0.5f * (hi - lo);
        }
      }
    }
  return out;
}

static void rodrigues(const float* p, float* T) {   // twist (tx,ty,tz,rx,ry,rz) -> 4x4, negated-skew convention of convert_pose.py:56
  const double rx = p[3], ry = p[4], rz = p[5];
  const double th = std::sqrt(rx * rx + ry * ry + rz * rz);
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (th > 1e-8) {
    const double ux = rx / th, uy = ry / th, uz = rz / th;
    const double Wm[9] = {0, uz, -uy, -uz, 0, ux, uy, -ux, 0};          // -[u]x
    double W2[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { W2[3 * i + j] = 0; for (int k = 0; k < 3; ++k) W2[3 * i + j] += Wm[3 * i + k] * Wm[3 * k + j]; }
    for (int i = 0; i < 9; ++i) R[i] += std::sin(th) * Wm[i] + (1 - std::cos(th)) * W2[i];
  }
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T[4 * i + j] = (float)R[3 * i + j]; T[4 * i + 3] = p[i]; }
  T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
}

template <class T> static T* to_dev(const std::vector<T>& v) { T* d; CK(hipMalloc(&d, v.size() * sizeof(T))); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

__global__ void downsample2(const float* __restrict__ in, float* __restrict__ out, int nimg, int h, int w, int C) {   // TF2 half-pixel bilinear 2x = 2x2 mean
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int ho = h / 2, wo = w / 2;
  if (i >= (long long)nimg * ho * wo * C) return;
  const int c = i % C; long long r = i / C; const int x = r % wo; r /= wo; const int y = r % ho; const int n = r / ho;
  const float* p = in + (((long long)n * h + 2 * y) * w + 2 * x) * C + c;
  out[i] = 0.25f * ((p[0] + p[C]) + (p[(long long)w * C] + p[(long long)w * C + C]));
}
__global__ void subsample(const float* __restrict__ in, float* __restrict__ out, int nimg, int h, int w, int s) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int ho = h / s, wo = w / s;
  if (i >= (long long)nimg * ho * wo) return;
  const int x = i % wo; long long r = i / wo; const int y = r % ho; const int n = r / ho;
  out[i] = in[((long long)n * h + (long long)y * s) * w + (long long)x * s];
}

// calibration launches for PMC passes (FETCH_SIZE / WRITE_SIZE): known byte counts in this kernel family's access widths
__global__ void calib_copy16(const float4* __restrict__ in, float4* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i];
}
__global__ void calib_read12(const float* __restrict__ in, float* __restrict__ out, long long npix) {   // 12-byte pixels, one per lane
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < npix) {
    const float a = in[3 * i], b = in[3 * i + 1], c = in[3 * i + 2];
    if (a + b + c == 12345.678f) out[i] = a;
  }
}

struct Run { std::vector<float> losses, dT; std::vector<std::vector<float>> dd; float fwd_us, bwd_us; };

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8, H = argc > 2 ? atoi(argv[2]) : 128, W = argc > 3 ? atoi(argv[3]) : 416;
  const int reps = argc > 4 ? atoi(argv[4]) : 20;
  const char* mode = argc > 5 ? argv[5] : "both";
  const int NS = argc > 6 ? atoi(argv[6]) : 4;
  const int N = 4;
  std::mt19937 g(5);
  // base: 8 distinct snippets, repeated to B (as roofline.py does)
  const int B0 = std::min(B, 8);
  std::vector<float> tgt0 = smooth_field(B0, H, W, 3, 8, g, -1.f, 1.f, 0.05f);
  std::vector<float> src0((size_t)B0 * N * H * W * 3);
  for (int b = 0; b < B0; ++b)
    for (int n = 0; n < N; ++n) {
      const int dx = (const int[]){-2, -1, 1, 2}[n], dy = (const int[]){1, 0, 0, -1}[n];
      for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x)
        memcpy(&src0[((((size_t)b * N + n) * H + (y + dy + H) % H) * W + (x + dx + W) % W) * 3], &tgt0[(((size_t)b * H + y) * W + x) * 3], 12);
    }
  std::vector<float> dep0 = smooth_field(B0, H, W, 1, 16, g, 1.f, 80.f, 0.05f);
  std::normal_distribution<float> Nrm(0.f, 1.f);
  const float sd[6] = {0.3f, 0.05f, 1.0f, 0.01f, 0.02f, 0.01f};
  std::vector<float> T0((size_t)B0 * N * 16), K0((size_t)B0 * 9);
  for (int i = 0; i < B0 * N; ++i) { float p[6]; for (int k = 0; k < 6; ++k) p[k] = Nrm(g) * sd[k] + (k >= 3 ? 1e-4f : 0.f); rodrigues(p, &T0[16 * i]); }
  for (int b = 0; b < B0; ++b) { const float k[9] = {0.58f * W, 0, 0.5f * W, 0, 1.92f * H, 0.5f * H, 0, 0, 1}; memcpy(&K0[9 * b], k, 36); }
  auto rep = [&](const std::vector<float>& v, size_t per) { std::vector<float> o((size_t)B * per); for (int b = 0; b < B; ++b) memcpy(&o[b * per], &v[(b % B0) * per], per * 4); return o; };
  std::vector<float> src = rep(src0, (size_t)N * H * W * 3), tgt = rep(tgt0, (size_t)H * W * 3), dep = rep(dep0, (size_t)H * W), Tm = rep(T0, N * 16), Km = rep(K0, 9);
  float* d_src[4]; float* d_tgt[4]; float* d_dep[4]; float* d_dd[4];
  int hs[4], ws[4]; float sc[4];
  d_src[0] = to_dev(src); d_tgt[0] = to_dev(tgt); d_dep[0] = to_dev(dep);
  float* d_T = to_dev(Tm); float* d_K = to_dev(Km);
  size_t pixels = 0;
  for (int s = 0; s < NS; ++s) {
    hs[s] = H >> s; ws[s] = W >> s; sc[s] = (float)(1 << s);
    pixels += (size_t)hs[s] * ws[s];
    if (s > 0) {
      CK(hipMalloc(&d_src[s], (size_t)B * N * hs[s] * ws[s] * 12)); CK(hipMalloc(&d_tgt[s], (size_t)B * hs[s] * ws[s] * 12)); CK(hipMalloc(&d_dep[s], (size_t)B * hs[s] * ws[s] * 4));
      long long n = (long long)B * N * hs[s] * ws[s] * 3;
      hipLaunchKernelGGL(downsample2, dim3((n + 255) / 256), dim3(256), 0, 0, d_src[s - 1], d_src[s], B * N, hs[s - 1], ws[s - 1], 3);
      n = (long long)B * hs[s] * ws[s] * 3;
      hipLaunchKernelGGL(downsample2, dim3((n + 255) / 256), dim3(256), 0, 0, d_tgt[s - 1], d_tgt[s], B, hs[s - 1], ws[s - 1], 3);
      n = (long long)B * hs[s] * ws[s];
      hipLaunchKernelGGL(subsample, dim3((n + 255) / 256), dim3(256), 0, 0, d_dep[0], d_dep[s], B, H, W, 1 << s);
    }
    CK(hipMalloc(&d_dd[s], (size_t)B * hs[s] * ws[s] * 4));
  }
  CK(hipDeviceSynchronize());
  size_t nws = 0;
  for (int s = 0; s < NS; ++s) nws += xpt_photo_fused_workspace_floats(B, N, hs[s], ws[s]);
  nws *= 2;
  float *d_ws, *d_loss, *d_dT, *d_g;
  CK(hipMalloc(&d_ws, nws * 4)); CK(hipMalloc(&d_loss, 2 * NS * B * 4)); CK(hipMalloc(&d_dT, B * N * 16 * 4));
  std::vector<float> ones(B, 1.f); d_g = to_dev(ones);
  const float* gs[4] = {d_g, d_g, d_g, d_g};
  const double fbytes = (double)B * pixels * (16 + 12 * N), bbytes = (double)B * pixels * (20 + 12 * N);

  auto run = [&](bool use_new) {
    Run r;
    auto fwd = [&](float* losses) {
      return use_new ? xpt_photo_march_ms_fwd(NS, d_src, d_dep, d_T, d_K, d_tgt, losses, d_ws, nws, B, N, hs, ws, sc, nullptr)
                     : xpt_photo_fused_ms_fwd(NS, d_src, d_dep, d_T, d_K, d_tgt, losses, d_ws, nws, B, N, hs, ws, sc, nullptr);
    };
    auto bwd = [&]() {
      return use_new ? xpt_photo_march_ms_bwd(NS, d_src, d_dep, d_T, d_K, d_tgt, gs, gs, d_dd, d_dT, d_ws, nws, B, N, hs, ws, sc, nullptr)
                     : xpt_photo_fused_ms_bwd(NS, d_src, d_dep, d_T, d_K, d_tgt, gs, gs, d_dd, d_dT, d_ws, nws, B, N, hs, ws, sc, nullptr);
    };
    CK(hipMemset(d_loss, 0, 2 * NS * B * 4)); CK(hipMemset(d_dT, 0, B * N * 64));
    for (int s = 0; s < NS; ++s) CK(hipMemset(d_dd[s], 0, (size_t)B * hs[s] * ws[s] * 4));
    int rc = fwd(d_loss); if (rc) { printf("fwd rc %d\n", rc); exit(1); }
    rc = bwd(); if (rc) { printf("bwd rc %d\n", rc); exit(1); }
    CK(hipDeviceSynchronize());
    r.losses.resize(2 * NS * B); CK(hipMemcpy(r.losses.data(), d_loss, r.losses.size() * 4, hipMemcpyDeviceToHost));
    r.dT.resize(B * N * 16); CK(hipMemcpy(r.dT.data(), d_dT, r.dT.size() * 4, hipMemcpyDeviceToHost));
    for (int s = 0; s < NS; ++s) { r.dd.emplace_back((size_t)B * hs[s] * ws[s]); CK(hipMemcpy(r.dd[s].data(), d_dd[s], r.dd[s].size() * 4, hipMemcpyDeviceToHost)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best_f = 1e30f, best_b = 1e30f;
    for (int round = 0; round < 3; ++round) {
      float ms;
      CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) fwd(nullptr); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1)); best_f = std::min(best_f, ms * 1e3f / reps);
      CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) bwd(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1)); best_b = std::min(best_b, ms * 1e3f / reps);
    }
    r.fwd_us = best_f; r.bwd_us = best_b;
    printf("%s B=%d %dx%d scales=%d: fwd %8.2f us = %6.0f GB/s (%.3f of 8 TB/s) | bwd %8.2f us = %6.0f GB/s (%.3f)\n", use_new ? "NEW" : "OLD", B, H, W, NS,
           best_f, fbytes / best_f * 1e-3, fbytes / best_f * 1e-3 / 8000, best_b, bbytes / best_b * 1e-3, bbytes / best_b * 1e-3 / 8000);
    fflush(stdout);
    return r;
  };
  if (getenv("LAB_CALIB")) {
    const long long n = 16LL << 20;                       // 256 MiB of float4
    float4 *ci, *co;
    CK(hipMalloc(&ci, n * 16)); CK(hipMalloc(&co, n * 16));
    CK(hipMemset(ci, 0, n * 16));
    for (int i = 0; i < 3; ++i) {
      hipLaunchKernelGGL(calib_copy16, dim3((unsigned)(n / 256)), dim3(256), 0, 0, ci, co, n);
      hipLaunchKernelGGL(calib_read12, dim3((unsigned)((n * 4 / 3 + 255) / 256)), dim3(256), 0, 0, (const float*)ci, (float*)co, n * 4 / 3);
    }
    CK(hipDeviceSynchronize());
    printf("calibration: calib_copy16 reads %lld B and writes %lld B; calib_read12 reads %lld B\n", n * 16, n * 16, (n * 4 / 3) * 12);
  }
  if (const char* plan = getenv("LAB_PLAN")) {             // "variant,rows0,rows1,rows2,rows3" -> xpt_photo_march_plan
    int v = 1, r[4] = {0, 0, 0, 0};
    sscanf(plan, "%d,%d,%d,%d,%d", &v, &r[0], &r[1], &r[2], &r[3]);
    const int rc = xpt_photo_march_plan(v, r[0], r[1], r[2], r[3]);
    printf("plan %s -> rc %d\n", plan, rc);
  }
  if (const char* tune = getenv("LAB_TUNE")) {             // "fwd_min_waves,bwd_min_waves,min_rows" -> xpt_photo_march_tune
    int f = 4096, bw = 1536, mr = 8;
    sscanf(tune, "%d,%d,%d", &f, &bw, &mr);
    printf("tune %s -> rc %d\n", tune, xpt_photo_march_tune(f, bw, mr));
  }
  Run a, b;
  const bool do_old = strcmp(mode, "new") != 0, do_new = strcmp(mode, "old") != 0;
  if (do_old) a = run(false);
  if (do_new) b = run(true);
  if (do_old && do_new) {
    auto cmp = [](const std::vector<float>& x, const std::vector<float>& y, const char* name) {
      double mx = 0, md = 0; size_t bad = 0, nan = 0;
      for (size_t i = 0; i < x.size(); ++i) mx = std::max(mx, (double)std::fabs(x[i]));
      for (size_t i = 0; i < x.size(); ++i) { const double d = std::fabs((double)x[i] - y[i]); if (!(d == d)) { ++nan; continue; } md = std::max(md, d); if (d > 1e-3 * mx) ++bad; }
      printf("  %-10s max|ref| %.4e  max diff %.3e (%.2e rel)  elements off by > 1e-3 of max: %zu / %zu, nan %zu\n", name, mx, md, md / (mx + 1e-30), bad, x.size(), nan);
    };
    cmp(a.losses, b.losses, "losses");
    cmp(a.dT, b.dT, "dT");
    for (int s = 0; s < NS; ++s) { char nm[32]; snprintf(nm, 32, "ddepth[%d]", s); cmp(a.dd[s], b.dd[s], nm); }
    printf("  losses[0..3] old %.6f %.6f %.6f %.6f | new %.6f %.6f %.6f %.6f\n", a.losses[0], a.losses[1], a.losses[B], a.losses[NS * B], b.losses[0], b.losses[1], b.losses[B], b.losses[NS * B]);
  }
  return 0;
}
