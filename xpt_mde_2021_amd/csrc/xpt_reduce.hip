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

// Wide mode (jobs whose rows are 16-byte aligned, n >= 256): a workgroup owns 256 consecutive outputs; every wave reads
// WHOLE 1 KiB rows of the partial matrix (one float4 per lane: 4x the bytes in flight of the scalar modes and DRAM bursts
// of 1 KiB instead of 256 B), the 4 waves take the splits round-robin and are combined through LDS in wave order.
__device__ inline void reduce_wide(const xpt_reduce_job& job, int first) {
  __shared__ float4 redw[3][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long i = (long long)first + 4 * lane;
  const bool live = i + 3 < job.n;                                     // n % 4 == 0 in this mode
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int g = 0; g < job.nseg; ++g) {
    const float* src = job.src[g] + (live ? i : 0);
    const long long stride = job.stride[g];
    const int ns = job.nsplit[g];
    for (int s0 = wave; s0 < ns; s0 += 32) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = s0 + 4 * u;
        const float4 x = *(const float4*)(src + (long long)min(s, ns - 1) * stride);
        const float keep = s < ns ? 1.f : 0.f;
        v[u] = make_float4(x.x * keep, x.y * keep, x.z * keep, x.w * keep);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
    }
  }
  if (wave > 0) redw[wave - 1][lane] = sum;
  __syncthreads();
  if (wave == 0 && live) {
    float4 o;
    o.x = ((sum.x + redw[0][lane].x) + redw[1][lane].x) + redw[2][lane].x;
    o.y = ((sum.y + redw[0][lane].y) + redw[1][lane].y) + redw[2][lane].y;
    o.z = ((sum.z + redw[0][lane].z) + redw[1][lane].z) + redw[2][lane].z;
    o.w = ((sum.w + redw[0][lane].w) + redw[1][lane].w) + redw[2][lane].w;
    *(float4*)(job.dst + i) = o;
  }
}

// Flat mode (wide-eligible jobs whose segments all have few splits: the large weights): the launch is bound by the
// workgroup dispatch rate (~6 ns per workgroup measured, tools/reduce_breakdown.py), not by HBM, when a workgroup only
// moves a few KiB, so here a workgroup owns 2048 consecutive outputs: every lane adds the splits of two float4 outputs
// itself, in split order (all loads of a segment in flight together), no LDS.
__device__ inline void reduce_flat(const xpt_reduce_job& job, int first) {
  long long i[2];
  bool live[2];
  float4 sum[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    i[u] = (long long)first + 1024 * u + 4 * threadIdx.x;
    live[u] = i[u] + 3 < job.n;                                        // n % 4 == 0 in this mode
    sum[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int g = 0; g < job.nseg; ++g) {
    const long long stride = job.stride[g];
    const int ns = job.nsplit[g];
    for (int s0 = 0; s0 < ns; s0 += 8) {
      float4 v[2][8];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float* src = job.src[g] + (live[u] ? i[u] : 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float4 x = *(const float4*)(src + (long long)min(s0 + k, ns - 1) * stride);
          const float keep = s0 + k < ns ? 1.f : 0.f;
          v[u][k] = make_float4(x.x * keep, x.y * keep, x.z * keep, x.w * keep);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          sum[u].x += v[u][k].x; sum[u].y += v[u][k].y; sum[u].z += v[u][k].z; sum[u].w += v[u][k].w;
        }
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (live[u]) *(float4*)(job.dst + i[u]) = sum[u];
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const xpt_reduce_job* __restrict__ jobs,
                                                               const int2* __restrict__ blockmap) {
  __shared__ float red[3][64];
  const int2 bm = blockmap[blockIdx.x];
  // by reference: the segment arrays are indexed with a loop variable, a by-value copy of the job would live in scratch
  const xpt_reduce_job& job = jobs[bm.x];
  if (job.split_waves == 16) {                 // wide mode: 256 outputs per workgroup, 16-byte loads (below)
    reduce_wide(job, bm.y);
    return;
  }
  if (job.split_waves == 32) {                 // flat mode: 2048 outputs per workgroup, few splits per segment
    reduce_flat(job, bm.y);
    return;
  }
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
