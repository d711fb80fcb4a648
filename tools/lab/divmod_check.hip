#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../xpt_mde_2021_amd/csrc/xpt_common.h"
__global__ void k(unsigned* bad, unsigned d0) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  // sample n over the whole 31-bit range, d over small and large values
  for (unsigned rep = 0; rep < 64; ++rep) {
    const unsigned n = (tid * 2654435761u + rep * 40503u) & 0x7fffffffu;
    const unsigned ds[8] = {1u, 2u, 3u, 11u, d0, 1056u, 65535u, (1u << 24) - 1u};
    for (int j = 0; j < 8; ++j) {
      unsigned r;
      const unsigned q = xpt_divmod(n, ds[j], r);
      if (q != n / ds[j] || r != n % ds[j]) atomicAdd(bad, 1u);
    }
    // edge values
    const unsigned e = rep < 32 ? 0x7fffffffu - tid % 4096u : tid % 4096u;
    for (int j = 0; j < 8; ++j) {
      unsigned r;
      const unsigned q = xpt_divmod(e, ds[j], r);
      if (q != e / ds[j] || r != e % ds[j]) atomicAdd(bad, 1u);
    }
  }
}
int main() {
  unsigned* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
  for (unsigned d0 = 5; d0 < 3000; d0 += 37) hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, bad, d0);
  unsigned h = 1; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("mismatches: %u\n", h);
  return h != 0;
}
