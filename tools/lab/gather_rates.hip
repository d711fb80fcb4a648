// tools/lab/gather_rates.hip -- what the vector-memory path of gfx950 charges for the fused march's load shapes.
// One wave = 64 consecutive target columns of one source view, marching down R rows; per row it issues the loads of one
// KIND and nothing else (no arithmetic on the data), so the time is the address / L1 / L2 / HBM path alone.
// Reported: ns per wave-row per CU (the fused forward at batch 128 spends ~96 ns there; HBM floor at 6.3 TB/s: 40 ns).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/lab/bin/gather_rates tools/lab/gather_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f2 __attribute__((ext_vector_type(2)));

enum { G_TODAY6, G_TAPS4, G_TAPS_X3X3, G_RGBA4, G_HALF_X3, G_TGT_DEPTH, G_TAPS4_SADDR, G_RGBA2, G_PLANAR6, G_ROWCOPY, G_LDS_TILE, G_LDS_FULL, G_COUNT };
static const char* gNames[] = {
  "today: x4+x2 two rows, +x3 target +x1 depth (6 loads)", "taps only: x4+x2 two rows (4 loads)", "taps as x3+x3 two rows (4 loads)",
  "RGBA texels: 4 aligned x4 (4 loads)", "half taps: one x3 per row (2 loads)", "target x3 + depth x1 only (2 loads)",
  "taps x4+x2 two rows, saddr + 32-bit voffset", "RGBA texels: one x4 per row (2 loads; pair shared)",
  "planar CHW: x2 per channel and row (6 loads)", "aligned row copy: x4 per lane, 3 loads per 64 texels (coalesced)",
  "LDS tile: per 8 rows 12 LDS-DMA rows of 1 KiB + 8 x 12 ds_read_b32 (no target / depth)",
  "LDS tile + target x3 + depth x1 per row"};

struct Args {
  const float* src;      // [nimg][h][w][3]  (RGBA kinds: [nimg][h][w][4])
  const float* tgt;      // [nimg/4][h][w][3]
  const float* depth;    // [nimg/4][h][w]
  float* out;
  int nimg, h, w, R, S, CH;
  float zoom;
};

template <int KIND>
__global__ __launch_bounds__(256) void gather_kernel(Args a) {
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long gw = (long long)blockIdx.x * 4 + wid;
  const long long nwaves = (long long)a.nimg * a.S * a.CH;
  if (gw >= nwaves) return;
  const int lane = threadIdx.x & 63;
  const int n = (int)(gw % 4);
  long long r_ = gw / 4;
  const int ck = (int)(r_ % a.CH); r_ /= a.CH;
  const int s = (int)(r_ % a.S);
  const int b = (int)(r_ / a.S);
  const int img = b * 4 + n;
  const int col = min(s * 64 + lane, a.w - 1);
  // smooth flow: source column = zoom * col + shift, source row = row + (1 + lane / 32): neighbouring lanes mostly touch neighbouring texels
  const int uf = min(max((int)(a.zoom * (float)col) + (n - 2), 0), a.w - 2);
  const int texel = (KIND == G_RGBA4 || KIND == G_RGBA2) ? 4 : 3;
  const float* simg = a.src + (long long)img * a.h * a.w * texel;
  const float* timg = a.tgt + (long long)b * a.h * a.w * 3;
  const float* dimg = a.depth + (long long)b * a.h * a.w;
  const int r0 = ck * a.R;
  f4 A0 = {0, 0, 0, 0}, A1 = A0, A2 = A0, A3 = A0;
  f2 B0 = {0, 0}, B1 = B0, B2 = B0, B3 = B0, B4 = B0, B5 = B0;
  f3 C0 = {0, 0, 0}, C1 = C0, C2 = C0, C3 = C0;
  float D0 = 0.f;
  const int rowB = a.w * texel * 4;     // bytes per source row
  for (int i = 0; i < a.R; ++i) {
    const int r = r0 + i;
    const int vf = min(r + 1 + (lane >> 5), a.h - 2);
    const int p = r * a.w + col;
    if constexpr (KIND == G_TODAY6 || KIND == G_TAPS4) {
      const float* t0 = simg + (vf * a.w + uf) * 3;
      const float* t1 = t0 + a.w * 3;
      asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(A0) : "v"(t0) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, off offset:16" : "+v"(B0) : "v"(t0) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(A1) : "v"(t1) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, off offset:16" : "+v"(B1) : "v"(t1) : "memory");
      if constexpr (KIND == G_TODAY6) {
        const float* pt = timg + 3 * p;
        const float* pd = dimg + p;
        asm volatile("global_load_dwordx3 %0, %1, off" : "+v"(C0) : "v"(pt) : "memory");
        asm volatile("global_load_dword %0, %1, off" : "+v"(D0) : "v"(pd) : "memory");
      }
    } else if constexpr (KIND == G_TAPS4_SADDR) {
      const int off0 = (vf * a.w + uf) * 12, off1 = off0 + rowB;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(A0) : "v"(off0), "s"(simg) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, %2 offset:16" : "+v"(B0) : "v"(off0), "s"(simg) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(A1) : "v"(off1), "s"(simg) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, %2 offset:16" : "+v"(B1) : "v"(off1), "s"(simg) : "memory");
    } else if constexpr (KIND == G_TAPS_X3X3) {
      const int off0 = (vf * a.w + uf) * 12, off1 = off0 + rowB;
      asm volatile("global_load_dwordx3 %0, %1, %2" : "+v"(C0) : "v"(off0), "s"(simg) : "memory");
      asm volatile("global_load_dwordx3 %0, %1, %2 offset:12" : "+v"(C1) : "v"(off0), "s"(simg) : "memory");
      asm volatile("global_load_dwordx3 %0, %1, %2" : "+v"(C2) : "v"(off1), "s"(simg) : "memory");
      asm volatile("global_load_dwordx3 %0, %1, %2 offset:12" : "+v"(C3) : "v"(off1), "s"(simg) : "memory");
    } else if constexpr (KIND == G_RGBA4) {
      const int off0 = (vf * a.w + uf) * 16, off1 = off0 + rowB;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(A0) : "v"(off0), "s"(simg) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "+v"(A1) : "v"(off0), "s"(simg) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(A2) : "v"(off1), "s"(simg) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "+v"(A3) : "v"(off1), "s"(simg) : "memory");
    } else if constexpr (KIND == G_RGBA2) {
      const int off0 = (vf * a.w + uf) * 16, off1 = off0 + rowB;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(A0) : "v"(off0), "s"(simg) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(A2) : "v"(off1), "s"(simg) : "memory");
    } else if constexpr (KIND == G_HALF_X3) {
      const int off0 = (vf * a.w + uf) * 12, off1 = off0 + rowB;
      asm volatile("global_load_dwordx3 %0, %1, %2" : "+v"(C0) : "v"(off0), "s"(simg) : "memory");
      asm volatile("global_load_dwordx3 %0, %1, %2" : "+v"(C2) : "v"(off1), "s"(simg) : "memory");
    } else if constexpr (KIND == G_TGT_DEPTH) {
      const int offt = p * 12, offd = p * 4;
      asm volatile("global_load_dwordx3 %0, %1, %2" : "+v"(C0) : "v"(offt), "s"(timg) : "memory");
      asm volatile("global_load_dword %0, %1, %2" : "+v"(D0) : "v"(offd), "s"(dimg) : "memory");
    } else if constexpr (KIND == G_PLANAR6) {
      // planes of h*w floats: [img][3][h][w]; same total bytes as HWC
      const int plane = a.h * a.w * 4;
      const int off0 = (vf * a.w + uf) * 4, off1 = off0 + a.w * 4;
      asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(B0) : "v"(off0), "s"(simg) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(B1) : "v"(off1), "s"(simg) : "memory");
      const int o2 = off0 + plane, o3 = off1 + plane, o4 = o2 + plane, o5 = o3 + plane;
      asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(B2) : "v"(o2), "s"(simg) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(B3) : "v"(o3), "s"(simg) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(B4) : "v"(o4), "s"(simg) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, %2" : "+v"(B5) : "v"(o5), "s"(simg) : "memory");
    } else if constexpr (KIND == G_ROWCOPY) {
      // 64 texels x 12 B = 768 B of one source row = 48 lanes x 16 B: one x4 load per row with 48 active lanes
      const int base = (min(r + 1, a.h - 1) * a.w + min(s * 64, a.w - 64)) * 12;
      const int off0 = base + min(lane, 47) * 16;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(A0) : "v"(off0), "s"(simg) : "memory");
    }
    // at most two rows of loads in flight per wave (the march waits for every row before its arithmetic)
    if (i & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(B0), "+v"(B1), "+v"(C0), "+v"(C1) : : "memory");
  asm volatile("" : "+v"(C2), "+v"(C3), "+v"(D0), "+v"(B2), "+v"(B3), "+v"(B4), "+v"(B5));
  const float sum = A0.x + A1.y + A2.z + A3.w + B0.x + B1.y + B2.x + B3.x + B4.x + B5.x + C0.x + C1.y + C2.z + C3.x + D0;
  if (sum == 12345.678f) a.out[gw] = sum;     // keep the loads alive, (almost) never store
}

template <int KIND>
__global__ __launch_bounds__(256) void lds_tile_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) float tiles[4][16 * 256];
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long gw = (long long)blockIdx.x * 4 + wid;
  const long long nwaves = (long long)a.nimg * a.S * a.CH;
  if (gw >= nwaves) return;
  const int lane = threadIdx.x & 63;
  const int n = (int)(gw % 4);
  long long r_ = gw / 4;
  const int ck = (int)(r_ % a.CH); r_ /= a.CH;
  const int s = (int)(r_ % a.S);
  const int b = (int)(r_ / a.S);
  const int img = b * 4 + n;
  const int col = min(s * 64 + lane, a.w - 1);
  const float* simg = a.src + (long long)img * a.h * a.w * 3;
  const float* timg = a.tgt + (long long)b * a.h * a.w * 3;
  const float* dimg = a.depth + (long long)b * a.h * a.w;
  float* tile = tiles[wid];
  const int r0 = ck * a.R;
  float acc = 0.f;
  f3 C0 = {0, 0, 0};
  float D0 = 0.f;
  for (int rb = 0; rb < a.R; rb += 8) {
    // bounding box of the block: source rows r0+rb .. +11, floats from the 16-byte aligned start of the strip
    const int v0 = min(r0 + rb, a.h - 12);
    const int f0 = (min(s * 64, a.w - 86) * 3) & ~3;
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      const float* g = simg + (long long)(v0 + t) * a.w * 3 + f0 + 4 * lane;
      __builtin_amdgcn_global_load_lds(g, tile + t * 256, 16, 0, 0);
    }
    if constexpr (KIND == G_LDS_FULL) {
      // (the real kernel would stream these per row; here all 8 rows' loads are issued behind the tile)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = r0 + rb + i;
      if constexpr (KIND == G_LDS_FULL) {
        const int p = r * a.w + col;
        const int offt = p * 12, offd = p * 4;
        asm volatile("global_load_dwordx3 %0, %1, %2" : "+v"(C0) : "v"(offt), "s"(timg) : "memory");
        asm volatile("global_load_dword %0, %1, %2" : "+v"(D0) : "v"(offd), "s"(dimg) : "memory");
      }
      const int trow = min(i + (lane >> 5), 10);
      const float* t0 = tile + trow * 256 + 3 * lane + (lane >> 4);
      float v[12];
#pragma unroll
      for (int e = 0; e < 6; ++e) { v[e] = t0[e]; v[6 + e] = t0[256 + e]; }
#pragma unroll
      for (int e = 0; e < 12; ++e) acc += v[e];
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(C0), "+v"(D0) : : "memory");
    acc += C0.x + D0;
  }
  if (acc == 12345.678f) a.out[gw] = acc;
}

template <int KIND>
void bench(const Args& a0, int batch, const char* tag) {
  Args a = a0;
  a.nimg = batch * 4;
  const long long nwaves = (long long)a.nimg * a.S * a.CH;
  const unsigned blocks = (unsigned)((nwaves + 3) / 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() {
    if constexpr (KIND == G_LDS_TILE || KIND == G_LDS_FULL) hipLaunchKernelGGL(lds_tile_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, a);
    else hipLaunchKernelGGL(gather_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, a);
  };
  for (int i = 0; i < 3; ++i) launch();
  const int reps = batch >= 64 ? 10 : 50;
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  const double waverows_per_cu = (double)nwaves * a.R / 256.0;
  printf("  %-8s B=%3d: %8.1f us  %6.1f ns per wave-row per CU\n", tag, batch, us, us * 1e3 / waverows_per_cu);
  fflush(stdout);
}

template <int K>
void all(const Args& a) {
  printf("%s\n", gNames[K]);
  bench<K>(a, 8, "L2/IC");
  bench<K>(a, 128, "HBM");
  if constexpr (K + 1 < G_COUNT) all<K + 1>(a);
}

int main() {
  Args a{};
  a.h = 128; a.w = 416; a.R = 32; a.S = 7; a.CH = 4; a.zoom = 1.0f;
  const size_t maximg = 512;
  float *src, *tgt, *depth, *out;
  hipMalloc(&src, maximg * a.h * a.w * 16 + 65536);
  hipMalloc(&tgt, maximg / 4 * a.h * a.w * 12 + 65536);
  hipMalloc(&depth, maximg / 4 * a.h * a.w * 4 + 65536);
  hipMalloc(&out, 1 << 20);
  hipMemset(src, 0, maximg * a.h * a.w * 16);
  hipMemset(tgt, 0, maximg / 4 * a.h * a.w * 12);
  hipMemset(depth, 0, maximg / 4 * a.h * a.w * 4);
  a.src = src; a.tgt = tgt; a.depth = depth; a.out = out;
  for (float zoom : {1.0f}) {
    a.zoom = zoom;
    printf("==== zoom %.1f (source columns per target column)\n", zoom);
    all<0>(a);
  }
  return 0;
}
