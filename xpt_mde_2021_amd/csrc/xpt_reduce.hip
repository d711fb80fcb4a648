// xpt_reduce.hip -- one launch that finishes EVERY deferred parameter gradient of a training step.
//
// The backward kernels of the BatchNorm / bias epilogues, the depthwise convolutions and the pointwise convolutions
// leave per-workgroup (split-K) partial sums in persistent HBM workspaces instead of each launching its own small
// finishing kernel (~570 launches of a few microseconds per step).  At the end of the backward pass this kernel adds
// them all, writing straight into the flat gradient buffer the fused Adam kernel and the RCCL all-reduce read:
//
//     dst_j[i] = sum_{seg} sum_{s < nsplit_seg} src_seg[s * stride_seg + i]        (fixed order: deterministic)
//
// Lanes always run along the outputs (coalesced rows of the partial matrices).  A job with few splits (<= 8) gives every
// thread one output; a job with more gives a 64-output group to the 4 waves of a workgroup, wave w adding the splits
// w, w + 4, ... and the four sums being combined through LDS in wave order.  Every lane issues its loads in batches of 8
// before adding, so a job costs a few memory round trips.  The host splits jobs with more than 256 splits into two
// passes (pass 1 writes <= 64-split group sums to a scratch matrix that pass 2 consumes).
#include "xpt_common.h"

namespace {

__global__ __launch_bounds__(256) void reduce_partials_kernel(const xpt_reduce_job* __restrict__ jobs,
                                                               const int2* __restrict__ blockmap) {
  __shared__ float red[3][64];
  const int2 bm = blockmap[blockIdx.x];
  const xpt_reduce_job job = jobs[bm.x];
  const int SW = job.split_waves;              // 1 or 4 waves share one 64-output group
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long i = (long long)bm.y + (SW == 1 ? threadIdx.x : lane);
  const int first = SW == 1 ? 0 : wave;
  const bool live = i < job.n;
  float sum = 0.f;
  for (int g = 0; g < job.nseg; ++g) {
    const float* src = job.src[g] + (live ? i : 0);
    const long long stride = job.stride[g];
    const int ns = job.nsplit[g];
    for (int s0 = first; s0 < ns; s0 += 8 * SW) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {   // unconditional loads from a clamped split, zeroed by a select: a guarded load
        const int s = s0 + u * SW;    // would get its own branch and s_waitcnt, i.e. one memory round trip per split
        const float x = src[(long long)min(s, ns - 1) * stride];
        v[u] = s < ns ? x : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
  }
  if (SW == 1) {
    if (live) job.dst[i] = sum;
    return;
  }
  if (wave > 0) red[wave - 1][lane] = sum;
  __syncthreads();
  if (wave == 0 && live) job.dst[i] = ((sum + red[0][lane]) + red[1][lane]) + red[2][lane];
}

}  // namespace

extern "C" int xpt_reduce_job_bytes(void) { return (int)sizeof(xpt_reduce_job); }

extern "C" int xpt_reduce_partials(const void* jobs, const void* blockmap, int nblocks, void* stream) {
  XPT_CHECK_PTR(jobs);
  XPT_CHECK_PTR(blockmap);
  if (nblocks <= 0) return XPT_ERR_SHAPE;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream,
                     (const xpt_reduce_job*)jobs, (const int2*)blockmap);
  return xpt_launch_status();
}
