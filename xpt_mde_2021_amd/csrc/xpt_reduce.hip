// xpt_reduce.hip -- one launch that finishes EVERY deferred parameter gradient of a training step.
//
// The backward kernels of the BatchNorm / bias epilogues, the depthwise convolutions and the pointwise convolutions
// leave per-workgroup (split-K) partial sums in persistent HBM workspaces instead of each launching its own small
// finishing kernel (~570 launches of a few microseconds per step).  At the end of the backward pass this kernel adds
// them all, writing straight into the flat gradient buffer the fused Adam kernel and the RCCL all-reduce read:
//
//     dst_j[i] = sum_{seg} sum_{s < nsplit_seg} src_seg[s * stride_seg + i]        (fixed order: deterministic)
//
// A job with many splits spreads them over `split_lanes` (1, 4, 16 or 64) lanes per output, combined by a fixed
// shuffle tree; every lane issues its loads in batches of 8 before adding, so a job costs a few memory round trips.
#include "xpt_common.h"

namespace {

__global__ __launch_bounds__(256) void reduce_partials_kernel(const xpt_reduce_job* __restrict__ jobs,
                                                               const int2* __restrict__ blockmap) {
  const int2 bm = blockmap[blockIdx.x];
  const xpt_reduce_job job = jobs[bm.x];
  const int SL = job.split_lanes;              // lanes per output
  const int OUTS = 64 / SL;                    // outputs per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sl = lane / OUTS, o = lane - sl * OUTS;
  const long long i = (long long)bm.y + wave * OUTS + o;
  const bool live = i < job.n;
  float sum = 0.f;
  for (int g = 0; g < job.nseg; ++g) {
    const float* src = job.src[g] + (live ? i : 0);
    const long long stride = job.stride[g];
    const int ns = job.nsplit[g];
    // this lane's splits: sl, sl + SL, ... ; 8 loads in flight, added in order
    for (int s0 = sl; s0 < ns; s0 += 8 * SL) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = s0 + u * SL;
        v[u] = s < ns ? src[(long long)s * stride] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += v[u];
    }
  }
  for (int off = 32; off >= OUTS; off >>= 1) sum += __shfl_down(sum, off, 64);
  if (live && sl == 0) job.dst[i] = sum;
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
