// Lab: does a consumer kernel find its producer's output in the L2 of ITS XCD when it runs on the XCD that wrote it?
// (The per-XCD L2s of gfx950 are not coherent with each other: if the runtime invalidates / writes them back wholesale at
// every kernel boundary, an image-to-XCD affinity of consecutive kernels buys nothing.)
//
// A chain producer -> consumer -> producer -> ... of small kernels inside ONE captured hipGraph, each over 8 "images" of
// S bytes (8 x S = the working set of a layer).  Workgroup i of a launch runs on XCD i % 8 (round-robin dispatch).
//   aligned : in both kernels the workgroups of image b are the ones with blockIdx % 8 == b  -> same XCD writes and reads
//   shifted : the consumer reads image (b + 3) % 8 from the XCD that wrote image b              -> always another XCD
// Time per kernel of the chain, HIP events around graph replays.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/lab/bin/xcd_handoff tools/lab/xcd_handoff.hip && tools/lab/bin/xcd_handoff
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                        \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

// y[image][j] = x[image'][j] * 1.0009765625 + 1 (bf16-sized traffic does not matter here: plain 16-byte vectors);
// image = blockIdx % 8, image' = (image + shift) % 8; wgs_per_image workgroups walk an image
__global__ __launch_bounds__(256) void hop(const float4* __restrict__ x, float4* __restrict__ y, int vec_per_image, int wgs_per_image,
                                           int shift) {
  const int image = blockIdx.x & 7, part = blockIdx.x >> 3;
  const int src = (image + shift) & 7;
  for (int j = part * 256 + threadIdx.x; j < vec_per_image; j += wgs_per_image * 256) {
    float4 v = x[(size_t)src * vec_per_image + j];
    v.x = v.x * 1.0009765625f + 1.f;
    y[(size_t)image * vec_per_image + j] = v;
  }
}

int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  const int hops = 200;
  printf("bytes per image | workgroups per image | us per kernel: aligned  shifted by 3 XCDs\n");
  for (int kib : {64, 256, 1024}) {
    for (int wpi : {4, 16, 64}) {
      const int vec = kib * 1024 / 16;
      float4 *a, *b;
      CK(hipMalloc(&a, (size_t)8 * vec * 16));
      CK(hipMalloc(&b, (size_t)8 * vec * 16));
      CK(hipMemset(a, 0, (size_t)8 * vec * 16));
      float us[2];
      for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int h = 0; h < hops; ++h) {
          hipLaunchKernelGGL(hop, dim3(8 * wpi), dim3(256), 0, s, (h & 1) ? b : a, (h & 1) ? a : b, vec, wpi, mode ? 3 : 0);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        us[mode] = ms * 1e3f / (5 * hops);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
      }
      printf("%6d KiB | %3d | %7.2f  %7.2f\n", kib, wpi, us[0], us[1]);
      CK(hipFree(a));
      CK(hipFree(b));
    }
  }
  return 0;
}
