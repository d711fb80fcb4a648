// tools/lab/lds_rates.hip -- LDS read shapes for bilinear taps staged in LDS (texel = 3 floats, lane stride 3 dwords):
// correctness of misaligned wide reads, and cycles per wave-instruction group at 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/lab/bin/lds_rates tools/lab/lds_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void misaligned_check(float* out) {
  __shared__ float lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (float)i;
  __syncthreads();
  const unsigned addr = (unsigned)(size_t)(lds) + (threadIdx.x * 3 + 1) * 4;   // 4-byte aligned only
  f2 a; f3 b; f4 c;
  asm volatile("ds_read_b64 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr) : "memory");
  asm volatile("ds_read_b96 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(addr) : "memory");
  asm volatile("ds_read_b128 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(c) : "v"(addr) : "memory");
  float* o = out + threadIdx.x * 12;
  o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; o[4] = b.z; o[5] = c.x; o[6] = c.y; o[7] = c.z; o[8] = c.w;
  o[9] = (float)(threadIdx.x * 3 + 1);
}

enum { L_B32x12, L_READ2x6, L_B64x6, L_B96x4, L_B128_B64x4, L_COUNT };
static const char* lNames[] = {"12 x ds_read_b32", "6 x ds_read2_b32 (adjacent dwords)", "6 x ds_read_b64 (4-byte aligned)",
                               "4 x ds_read_b96 (4-byte aligned)", "2 x (ds_read_b128 + ds_read_b64) (4-byte aligned)"};

template <int KIND>
__global__ __launch_bounds__(256) void lds_kernel(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[4][2048];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = lane; i < 2048; i += 64) lds[wid][i] = (float)i;
  __syncthreads();
  // lane's tap row 0 at dword 3*lane + jitter, row 1 one tile row (256 dwords... use 259 to mimic a pitch) further
  const unsigned a0 = (unsigned)(size_t)(&lds[wid][0]) + (3 * lane + (lane >> 4)) * 4;
  const unsigned a1 = a0 + 259 * 4;
  float r[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) r[i] = 0.f;
  f2 p[6]; f3 q[4]; f4 w[2];
  float acc = 0.f;
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\ns_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == L_B32x12) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[i]) : "v"(a0), "n"(i * 4) : "memory");
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[6 + i]) : "v"(a1), "n"(i * 4) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11])::"memory");
      acc += r[0] + r[11];
    } else if constexpr (KIND == L_READ2x6) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(p[i]) : "v"(a0), "n"(2 * i), "n"(2 * i + 1) : "memory");
        asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(p[3 + i]) : "v"(a1), "n"(2 * i), "n"(2 * i + 1) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5])::"memory");
      acc += p[0].x + p[5].y;
    } else if constexpr (KIND == L_B64x6) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(p[i]) : "v"(a0), "n"(8 * i) : "memory");
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(p[3 + i]) : "v"(a1), "n"(8 * i) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5])::"memory");
      acc += p[0].x + p[5].y;
    } else if constexpr (KIND == L_B96x4) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        asm volatile("ds_read_b96 %0, %1 offset:%2" : "=v"(q[i]) : "v"(a0), "n"(12 * i) : "memory");
        asm volatile("ds_read_b96 %0, %1 offset:%2" : "=v"(q[2 + i]) : "v"(a1), "n"(12 * i) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3])::"memory");
      acc += q[0].x + q[3].z;
    } else {
      asm volatile("ds_read_b128 %0, %1" : "=v"(w[0]) : "v"(a0) : "memory");
      asm volatile("ds_read_b64 %0, %1 offset:16" : "=v"(p[0]) : "v"(a0) : "memory");
      asm volatile("ds_read_b128 %0, %1" : "=v"(w[1]) : "v"(a1) : "memory");
      asm volatile("ds_read_b64 %0, %1 offset:16" : "=v"(p[1]) : "v"(a1) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(p[0]), "+v"(p[1])::"memory");
      acc += w[0].x + p[1].y;
    }
  }
  asm volatile("s_memtime %0\ns_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (lane == 0) cyc[gw] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int KIND>
void report(float* out, unsigned long long* cyc) {
  const int iters = 2000;
  printf("%-52s", lNames[KIND]);
  for (int wps : {1, 2, 4, 5}) {      // 32 KB per workgroup: at most 5 per CU
    const int blocks = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(lds_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(lds_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf(" | w=%d: %7.1f tick/pixel/wave %7.2f ns/pixel-row/CU", wps, (double)h[h.size() / 2] / iters, ms * 1e6 / ((double)iters * wps * 4));
  }
  printf("\n"); fflush(stdout);
}

template <int K> void all(float* out, unsigned long long* cyc) { report<K>(out, cyc); if constexpr (K + 1 < L_COUNT) all<K + 1>(out, cyc); }

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 8 * 256 * 4 + 4096);
  hipMalloc(&cyc, 256 * 8 * 4 * 8);
  hipLaunchKernelGGL(misaligned_check, dim3(1), dim3(64), 0, 0, out);
  std::vector<float> h(64 * 12);
  hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
  int bad64 = 0, bad96 = 0, bad128 = 0;
  for (int t = 0; t < 64; ++t) {
    const float* o = &h[t * 12]; const float e = o[9];
    if (o[0] != e || o[1] != e + 1) ++bad64;
    if (o[2] != e || o[3] != e + 1 || o[4] != e + 2) ++bad96;
    if (o[5] != e || o[6] != e + 1 || o[7] != e + 2 || o[8] != e + 3) ++bad128;
  }
  printf("misaligned (4-byte aligned) LDS reads: wrong lanes of 64: b64 %d, b96 %d, b128 %d   (lane 1: %g %g | %g %g %g | %g %g %g %g, expect from %g)\n",
         bad64, bad96, bad128, h[12], h[13], h[14], h[15], h[16], h[17], h[18], h[19], h[20], h[21]);
  all<0>(out, cyc);
  return 0;
}
