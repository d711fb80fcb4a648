// xpt_celltail.hip -- the elementwise "tail" of a NASNet-A cell in ONE launch forward and ONE launch backward (gfx950).
//
// keras builds the end of every cell from small layers (tensorflow.keras.applications.nasnet as instantiated by the
// reference's PretrainedModel, model/build_model/pretrained_nets.py:11-44):
//   _normal_a_cell:     x3 = add([AveragePooling2D(3,1,'same')(h), p]);  x4 = add([avgpool(p), avgpool(p)]);
//                       x  = concatenate([p, x1, x2, x3, x4, x5])
//   _reduction_a_cell:  x4 = add([x2, avgpool(x1)]);  x = concatenate([x2, x3, x4, x5])
// and the ONLY consumers of a cell output are Activation('relu') layers (the next cell's 1x1 "squeeze" convolution, the
// adjust block of the cell after, the final activation / the decoder taps).  So the tail is one function
//   out[:, s F:(s+1) F] = relu( sum_k scale_k * (pooled_k ? avgpool3same(in_k) : in_k) )         s = 0 .. nslices-1
// of a handful of F-channel NHWC tensors: 5 launches (2 pools, add, concat, relu) become 1, and in the backward the
// gradient accumulation over the consumers, the ReLU mask, the two pool adjoints and the concat split become 1.
//
// Layout: activations are NHWC rows [M = B H W][channels] (bf16 or fp32), inputs may be channel slices (row pitch).
// grid.y = slice (forward) / job (backward), so the term table of a workgroup is wave-uniform (SGPR loads).

#include "xpt_common.h"

namespace {

template <typename T> __device__ inline float ldf(const T* p);
template <> __device__ inline float ldf<float>(const float* p) { return *p; }
template <> __device__ inline float ldf<xpt_half_t>(const xpt_half_t* p) { return xpt_half2float(*p); }
template <typename T> __device__ inline void stf(T* p, float v);
template <> __device__ inline void stf<float>(float* p, float v) { *p = v; }
template <> __device__ inline void stf<xpt_half_t>(xpt_half_t* p, float v) { *p = xpt_float2half(v); }

template <typename T, int V> struct RowVec;
template <> struct RowVec<float, 4> { typedef float4 type; };
template <> struct RowVec<float, 2> { typedef float2 type; };
template <> struct RowVec<float, 1> { typedef float type; };
template <> struct RowVec<xpt_half_t, 8> { typedef uint4 type; };
template <> struct RowVec<xpt_half_t, 4> { typedef uint2 type; };
template <> struct RowVec<xpt_half_t, 2> { typedef unsigned type; };
template <> struct RowVec<xpt_half_t, 1> { typedef unsigned short type; };

template <typename T, int V>
__device__ inline void load_row(const T* p, float (&out)[V]) {
  typename RowVec<T, V>::type raw = *(const typename RowVec<T, V>::type*)p;
  const T* e = (const T*)&raw;
#pragma unroll
  for (int i = 0; i < V; ++i) out[i] = ldf<T>(e + i);
}

template <typename T, int V>
__device__ inline void store_row(T* p, const float (&v)[V]) {
  typename RowVec<T, V>::type raw;
  T* e = (T*)&raw;
#pragma unroll
  for (int i = 0; i < V; ++i) stf<T>(e + i, v[i]);
  *(typename RowVec<T, V>::type*)p = raw;
}

constexpr int MAX_SLICES = 8, MAX_TERMS = 2, MAX_GRADS = 4, MAX_DENSE = 4, MAX_BTERMS = 3;

struct FwdTerm {
  const void* src;
  long long pitch;
  int pooled;
  float scale;
};
struct TailFwd {
  FwdTerm t[MAX_SLICES][MAX_TERMS];
  int nterms[MAX_SLICES];
};

__device__ inline int window_count(int y, int x, int H, int W) {
  return ((y > 0) + 1 + (y < H - 1)) * ((x > 0) + 1 + (x < W - 1));
}

template <typename T, int V>
__global__ __launch_bounds__(256) void cell_tail_fwd_kernel(TailFwd a, T* __restrict__ out, long long out_pitch, int B, int H,
                                                            int W, int F, XcdSweep sw) {
  const int s = blockIdx.y;
  const int groups = F / V;
  const long long total = (long long)B * H * W * groups;
  const int nt = a.nterms[s];
  long long idx, end;
  if (!sw.range(blockIdx.x, total, idx, end)) return;
  for (idx += threadIdx.x; idx < end; idx += 256) {
    unsigned c0_;
    long long p = (long long)xpt_divmod((unsigned)idx, (unsigned)groups, c0_);
    const int c0 = (int)c0_ * V;
    const long long pix = p;
    unsigned r1_, r2_;
    const unsigned q1_ = xpt_divmod((unsigned)p, (unsigned)W, r1_);
    const int b = (int)xpt_divmod(q1_, (unsigned)H, r2_);
    const int x = (int)r1_, y = (int)r2_;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < MAX_TERMS; ++k) {
      if (k >= nt) break;                                              // uniform per workgroup
      const FwdTerm t = a.t[s][k];
      const T* src = (const T*)t.src;
      if (!t.pooled) {
        float v[V];
        load_row<T, V>(src + pix * t.pitch + c0, v);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += v[j] * t.scale;
      } else {
        float sum[V];
#pragma unroll
        for (int j = 0; j < V; ++j) sum[j] = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            const int ny = y + dy, nx = x + dx;
            const bool ok = ny >= 0 && ny < H && nx >= 0 && nx < W;
            const int cy = min(max(ny, 0), H - 1), cx = min(max(nx, 0), W - 1);
            float v[V];
            load_row<T, V>(src + (((long long)b * H + cy) * W + cx) * t.pitch + c0, v);
            const float wgt = ok ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < V; ++j) sum[j] += v[j] * wgt;
          }
        }
        const float norm = t.scale / (float)window_count(y, x, H, W);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += sum[j] * norm;
      }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = fmaxf(acc[j], 0.f);
    store_row<T, V>(out + pix * out_pitch + (long long)s * F + c0, acc);
  }
}

// Backward.  gm = [out > 0] * (sum of the consumers' gradients), written for every slice (the inputs that enter the tail
// through ONE identity term take their gradient as a channel-slice view of gm, times nothing: such terms have scale 1);
// the inputs with a pooled term or several terms get a dense gradient of their own:
//   d_in[m] = sum_terms scale * (pooled ? sum_{n in window(m)} gm_slice[n] / cnt(n) : gm_slice[m])
// evaluated from the consumers' gradients directly (gm of the neighbours is recomputed, not read back: one launch).
struct BwdTerm {
  int slice, pooled;
  float scale;
};
struct BwdDense {
  void* out;                   // [M, F] dense
  int nterms;
  BwdTerm t[MAX_BTERMS];
};
struct TailBwd {
  const void* g[MAX_GRADS];    // consumers' gradients [M, nslices F] (row pitch each)
  long long gpitch[MAX_GRADS];
  int ngrads;
  const void* out;             // the forward output (its sign is the ReLU mask)
  long long out_pitch;
  void* gm;                    // [M, nslices F] dense
  int nslices, ndense;
  BwdDense d[MAX_DENSE];
};

template <typename T, int V>
__device__ inline void masked_grad(const TailBwd& a, long long pix, long long ch, float (&gmv)[V]) {
  float o[V];
  load_row<T, V>((const T*)a.out + pix * a.out_pitch + ch, o);
#pragma unroll
  for (int j = 0; j < V; ++j) gmv[j] = 0.f;
#pragma unroll
  for (int k = 0; k < MAX_GRADS; ++k) {
    if (k >= a.ngrads) break;
    float v[V];
    load_row<T, V>((const T*)a.g[k] + pix * a.gpitch[k] + ch, v);
#pragma unroll
    for (int j = 0; j < V; ++j) gmv[j] += v[j];
  }
#pragma unroll
  for (int j = 0; j < V; ++j) gmv[j] = o[j] > 0.f ? gmv[j] : 0.f;
}

template <typename T, int V>
__global__ __launch_bounds__(256) void cell_tail_bwd_kernel(TailBwd a, int B, int H, int W, int F, XcdSweep sw) {
  const int job = blockIdx.y;
  const int groups = F / V;
  const long long total = (long long)B * H * W * groups;
  const long long gm_pitch = (long long)a.nslices * F;
  long long idx, end;
  if (!sw.range(blockIdx.x, total, idx, end)) return;      // (pixel-major indices: image-to-XCD numbering, xpt_common.h)
  for (idx += threadIdx.x; idx < end; idx += 256) {
    unsigned c0_;
    long long p = (long long)xpt_divmod((unsigned)idx, (unsigned)groups, c0_);
    const int c0 = (int)c0_ * V;
    const long long pix = p;
    if (job < a.nslices) {                                             // uniform per workgroup
      float gmv[V];
      const long long ch = (long long)job * F + c0;
      masked_grad<T, V>(a, pix, ch, gmv);
      store_row<T, V>((T*)a.gm + pix * gm_pitch + ch, gmv);
      continue;
    }
    const BwdDense& d = a.d[job - a.nslices];
    unsigned r1_, r2_;
    const unsigned q1_ = xpt_divmod((unsigned)p, (unsigned)W, r1_);
    const int b = (int)xpt_divmod(q1_, (unsigned)H, r2_);
    const int x = (int)r1_, y = (int)r2_;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < MAX_BTERMS; ++k) {
      if (k >= d.nterms) break;
      const BwdTerm t = d.t[k];
      const long long ch = (long long)t.slice * F + c0;
      if (!t.pooled) {
        float v[V];
        masked_grad<T, V>(a, pix, ch, v);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += v[j] * t.scale;
      } else {
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            const int ny = y + dy, nx = x + dx;
            const bool ok = ny >= 0 && ny < H && nx >= 0 && nx < W;
            const int cy = min(max(ny, 0), H - 1), cx = min(max(nx, 0), W - 1);
            float v[V];
            masked_grad<T, V>(a, ((long long)b * H + cy) * W + cx, ch, v);
            const float wgt = ok ? t.scale / (float)window_count(cy, cx, H, W) : 0.f;
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += v[j] * wgt;
          }
        }
      }
    }
    store_row<T, V>((T*)d.out + pix * F + c0, acc);
  }
}

// ------------------------------------------------------------------------------------------------ adjust block gather
// _adjust_block of keras NASNet when p has twice the resolution of the cell ("spatial" mode):
//   p1 = AveragePooling2D((1,1), strides 2)(relu p)            = p[2i, 2j]
//   p2 = the same of Cropping2D(((1,0),(1,0)))(ZeroPadding2D(((0,1),(0,1)))(relu p))     = p[2i+1, 2j+1] (0 outside)
// followed by one 1x1 convolution each.  Forward: ONE gather into out [M/4][2C] (the two halves feed the two GEMMs as
// channel slices); backward: ONE scatter of the two data gradients into a dense d_p (the other positions are zero).
// PyTorch's formulation costs pad (2 launches) + two strided copies forward and four slice_backward (zeros + copy) pairs,
// a pad backward and an accumulation in the backward.
template <typename T, int V>
__global__ __launch_bounds__(256) void adjust_gather_kernel(const T* __restrict__ in, long long in_pitch, T* __restrict__ out,
                                                            int B, int H, int W, int C, int H2, int W2, XcdSweep sw) {
  const int groups = C / V;
  const long long total = (long long)B * H2 * W2 * 2 * groups;
  long long idx, end;
  if (!sw.range(blockIdx.x, total, idx, end)) return;      // (pixel-major indices: image-to-XCD numbering, xpt_common.h)
  for (idx += threadIdx.x; idx < end; idx += 256) {
    unsigned c0_;
    long long p = (long long)xpt_divmod((unsigned)idx, (unsigned)groups, c0_);
    const int c0 = (int)c0_ * V;
    const int half = (int)(p & 1); p >>= 1;
    const long long opix = p;
    unsigned r1_, r2_;
    const unsigned q1_ = xpt_divmod((unsigned)p, (unsigned)W2, r1_);
    const int b = (int)xpt_divmod(q1_, (unsigned)H2, r2_);
    const int j = (int)r1_, i = (int)r2_;
    const int y = 2 * i + half, x = 2 * j + half;
    float v[V];
    const bool ok = y < H && x < W;
    load_row<T, V>(in + (((long long)b * H + (ok ? y : 0)) * W + (ok ? x : 0)) * in_pitch + c0, v);
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = ok ? v[e] : 0.f;
    store_row<T, V>(out + opix * (2LL * C) + (long long)half * C + c0, v);
  }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void adjust_scatter_kernel(const T* __restrict__ d1, long long pitch1,
                                                             const T* __restrict__ d2, long long pitch2, T* __restrict__ out,
                                                             int B, int H, int W, int C, int H2, int W2, XcdSweep sw) {
  const int groups = C / V;
  const long long total = (long long)B * H * W * groups;
  long long idx, end;
  if (!sw.range(blockIdx.x, total, idx, end)) return;      // (pixel-major indices: image-to-XCD numbering, xpt_common.h)
  for (idx += threadIdx.x; idx < end; idx += 256) {
    unsigned c0_;
    long long p = (long long)xpt_divmod((unsigned)idx, (unsigned)groups, c0_);
    const int c0 = (int)c0_ * V;
    const long long pix = p;
    unsigned r1_, r2_;
    const unsigned q1_ = xpt_divmod((unsigned)p, (unsigned)W, r1_);
    const int b = (int)xpt_divmod(q1_, (unsigned)H, r2_);
    const int x = (int)r1_, y = (int)r2_;
    float v[V];
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = 0.f;
    const long long src = ((long long)b * H2 + (y >> 1)) * W2 + (x >> 1);
    if (((y | x) & 1) == 0) {
      if (d1 != nullptr) load_row<T, V>(d1 + src * pitch1 + c0, v);
    } else if ((y & x & 1) == 1) {
      if (d2 != nullptr) load_row<T, V>(d2 + src * pitch2 + c0, v);
    }
    store_row<T, V>(out + pix * C + c0, v);
  }
}

// ------------------------------------------------------------------------------------------------ reduction-cell pools
// _reduction_a_cell pools h twice with one geometry: ZeroPadding2D(correct_pad(h, 3)) -> MaxPooling2D(3, strides 2,
// 'valid') and -> AveragePooling2D(3, strides 2, 'valid') (padding zeros take part in the max and in the divisor 9).
// Forward: one launch for both (+ the arg-max tap for the backward); backward: one gather per input pixel over the <= 4
// windows that contain it (deterministic, no atomics).
template <typename T, int V>
__global__ __launch_bounds__(256) void pool_pair_fwd_kernel(const T* __restrict__ in, long long in_pitch, T* __restrict__ mp,
                                                            T* __restrict__ ap, unsigned char* __restrict__ arg, int B, int H,
                                                            int W, int C, int OH, int OW, int pad_t, int pad_l, XcdSweep sw) {
  const int groups = C / V;
  const long long total = (long long)B * OH * OW * groups;
  long long idx, end;
  if (!sw.range(blockIdx.x, total, idx, end)) return;      // (pixel-major indices: image-to-XCD numbering, xpt_common.h)
  for (idx += threadIdx.x; idx < end; idx += 256) {
    unsigned c0_;
    long long p = (long long)xpt_divmod((unsigned)idx, (unsigned)groups, c0_);
    const int c0 = (int)c0_ * V;
    const long long opix = p;
    unsigned r1_, r2_;
    const unsigned q1_ = xpt_divmod((unsigned)p, (unsigned)OW, r1_);
    const int b = (int)xpt_divmod(q1_, (unsigned)OH, r2_);
    const int j = (int)r1_, i = (int)r2_;
    float best[V], sum[V];
    int at[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { best[e] = -INFINITY; sum[e] = 0.f; at[e] = 0; }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int y = 2 * i - pad_t + t / 3, x = 2 * j - pad_l + t % 3;
      const bool ok = y >= 0 && y < H && x >= 0 && x < W;
      float v[V];
      load_row<T, V>(in + (((long long)b * H + (ok ? y : 0)) * W + (ok ? x : 0)) * in_pitch + c0, v);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const float u = ok ? v[e] : 0.f;
        sum[e] += u;
        if (u > best[e]) { best[e] = u; at[e] = t; }
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) sum[e] *= (1.f / 9.f);
    store_row<T, V>(mp + opix * C + c0, best);
    store_row<T, V>(ap + opix * C + c0, sum);
#pragma unroll
    for (int e = 0; e < V; ++e) arg[opix * C + c0 + e] = (unsigned char)at[e];
  }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void pool_pair_bwd_kernel(const T* __restrict__ gmp, long long pitch_m,
                                                            const T* __restrict__ gmp2, long long pitch_m2,
                                                            const T* __restrict__ gap, long long pitch_a,
                                                            const unsigned char* __restrict__ arg, T* __restrict__ dh, int B,
                                                            int H, int W, int C, int OH, int OW, int pad_t, int pad_l, XcdSweep sw) {
  const int groups = C / V;
  const long long total = (long long)B * H * W * groups;
  long long idx, end;
  if (!sw.range(blockIdx.x, total, idx, end)) return;      // (pixel-major indices: image-to-XCD numbering, xpt_common.h)
  for (idx += threadIdx.x; idx < end; idx += 256) {
    unsigned c0_;
    long long p = (long long)xpt_divmod((unsigned)idx, (unsigned)groups, c0_);
    const int c0 = (int)c0_ * V;
    const long long pix = p;
    unsigned r1_, r2_;
    const unsigned q1_ = xpt_divmod((unsigned)p, (unsigned)W, r1_);
    const int b = (int)xpt_divmod(q1_, (unsigned)H, r2_);
    const int x = (int)r1_, y = (int)r2_;
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    // windows (i, j) with 2 i - pad_t <= y <= 2 i - pad_t + 2
    const int yy = y + pad_t, xx = x + pad_l;
#pragma unroll
    for (int di = 0; di < 2; ++di) {
#pragma unroll
      for (int dj = 0; dj < 2; ++dj) {
        const int i = (yy >> 1) - di, j = (xx >> 1) - dj;
        const int ty = yy - 2 * i, tx = xx - 2 * j;                   // tap of this pixel inside window (i, j)
        const bool ok = i >= 0 && i < OH && j >= 0 && j < OW && ty <= 2 && tx <= 2;
        const long long o = ((long long)b * OH + (ok ? i : 0)) * OW + (ok ? j : 0);
        float gm[V], ga[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { gm[e] = 0.f; ga[e] = 0.f; }
        if (gmp != nullptr) load_row<T, V>(gmp + o * pitch_m + c0, gm);
        if (gmp2 != nullptr) {                         // the max-pooled tensor had two consumers: their gradients added here
          float g2[V];
          load_row<T, V>(gmp2 + o * pitch_m2 + c0, g2);
#pragma unroll
          for (int e = 0; e < V; ++e) gm[e] += g2[e];
        }
        if (gap != nullptr) load_row<T, V>(gap + o * pitch_a + c0, ga);
        const int tap = ty * 3 + tx;
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const bool mine = arg[o * C + c0 + e] == (unsigned char)tap;
          const float g = (mine ? gm[e] : 0.f) + ga[e] * (1.f / 9.f);
          acc[e] += ok ? g : 0.f;
        }
      }
    }
    store_row<T, V>(dh + pix * C + c0, acc);
  }
}

inline bool aligned_for(const void* p, long long pitch, int v, int esz) {
  return ((uintptr_t)p) % (size_t)(v * esz) == 0 && pitch % v == 0;
}

}  // namespace

extern "C" {

/* out [M, nslices F] (row pitch out_pitch) = relu(sum of <= 2 terms per slice); term k of slice s: src[s*2+k] ([M, F] rows
 * with pitch[s*2+k]), pooled[s*2+k] (3x3 SAME average, divisor without the padding), scale[s*2+k].  dtype 0 fp32, 1 bf16. */
int xpt_cell_tail_fwd(int nslices, const int* nterms, const void* const* src, const long long* pitch, const int* pooled,
                      const float* scale, void* out, long long out_pitch, int B, int H, int W, int F, int dtype,
                      void* stream) {
  XPT_CHECK_PTR(nterms); XPT_CHECK_PTR(src); XPT_CHECK_PTR(pitch); XPT_CHECK_PTR(pooled); XPT_CHECK_PTR(scale);
  XPT_CHECK_PTR(out);
  if (nslices < 1 || nslices > MAX_SLICES || (dtype != 0 && dtype != 1)) return XPT_ERR_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || F <= 0 || out_pitch < (long long)nslices * F) return XPT_ERR_SHAPE;
  const int esz = dtype == 0 ? 4 : 2;
  int v = dtype == 0 ? 4 : 8;
  TailFwd a;
  for (;;) {
    bool all = F % v == 0 && aligned_for(out, out_pitch, v, esz);
    for (int s = 0; s < nslices && all; ++s)
      for (int k = 0; k < nterms[s] && k < MAX_TERMS && all; ++k) all = aligned_for(src[s * 2 + k], pitch[s * 2 + k], v, esz);
    if (all || v == 1) break;
    v >>= 1;
  }
  for (int s = 0; s < nslices; ++s) {
    if (nterms[s] < 1 || nterms[s] > MAX_TERMS) return XPT_ERR_ARG;
    a.nterms[s] = nterms[s];
    for (int k = 0; k < nterms[s]; ++k) {
      if (src[s * 2 + k] == nullptr) return XPT_ERR_NULL;
      if (pitch[s * 2 + k] < F) return XPT_ERR_SHAPE;
      a.t[s][k] = FwdTerm{src[s * 2 + k], pitch[s * 2 + k], pooled[s * 2 + k], scale[s * 2 + k]};
    }
  }
  const long long total = (long long)B * H * W * (F / v);
  if (total >= (1LL << 31)) return XPT_ERR_SHAPE;      // (the kernels split flat indices in 32 bits)
  const XcdSweep sw = xpt_xcd_sweep(total, 4096);
  const dim3 grid(sw.grid, nslices);
  hipStream_t st = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
#define XPT_TAIL(T, V) \
  hipLaunchKernelGGL((cell_tail_fwd_kernel<T, V>), grid, dim3(256), 0, st, a, (T*)out, out_pitch, B, H, W, F, sw)
  if (dtype == 0) {
    if (v == 4) XPT_TAIL(float, 4);
    else if (v == 2) XPT_TAIL(float, 2);
    else XPT_TAIL(float, 1);
  } else {
    if (v == 8) XPT_TAIL(xpt_half_t, 8);
    else if (v == 4) XPT_TAIL(xpt_half_t, 4);
    else if (v == 2) XPT_TAIL(xpt_half_t, 2);
    else XPT_TAIL(xpt_half_t, 1);
  }
#undef XPT_TAIL
  return xpt_launch_status();
}

/* Backward of xpt_cell_tail_fwd.  grads[ngrads] ([M, nslices F] rows, pitch gpitch[]) are the consumers' gradients of out;
 * gm [M, nslices F] (dense) receives [out > 0] * their sum.  Dense input gradient j (0 .. ndense-1): dense_out[j] [M, F] =
 * sum over its bterms[j] terms (slice, pooled, scale at index j*3 + k) of the (pooled-adjoint of the) masked gradient. */
int xpt_cell_tail_bwd(int ngrads, const void* const* grads, const long long* gpitch, const void* out, long long out_pitch,
                      void* gm, int nslices, int ndense, void* const* dense_out, const int* bterms, const int* slice,
                      const int* pooled, const float* scale, int B, int H, int W, int F, int dtype, void* stream) {
  XPT_CHECK_PTR(grads); XPT_CHECK_PTR(gpitch); XPT_CHECK_PTR(out); XPT_CHECK_PTR(gm);
  if (ngrads < 1 || ngrads > MAX_GRADS || nslices < 1 || nslices > MAX_SLICES || ndense < 0 || ndense > MAX_DENSE ||
      (dtype != 0 && dtype != 1))
    return XPT_ERR_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || F <= 0 || out_pitch < (long long)nslices * F) return XPT_ERR_SHAPE;
  if (ndense > 0 && (dense_out == nullptr || bterms == nullptr || slice == nullptr || pooled == nullptr || scale == nullptr))
    return XPT_ERR_NULL;
  const int esz = dtype == 0 ? 4 : 2;
  int v = dtype == 0 ? 4 : 8;
  TailBwd a;
  a.ngrads = ngrads; a.out = out; a.out_pitch = out_pitch; a.gm = gm; a.nslices = nslices; a.ndense = ndense;
  for (int k = 0; k < ngrads; ++k) {
    if (grads[k] == nullptr) return XPT_ERR_NULL;
    if (gpitch[k] < (long long)nslices * F) return XPT_ERR_SHAPE;
    a.g[k] = grads[k];
    a.gpitch[k] = gpitch[k];
  }
  for (int j = 0; j < ndense; ++j) {
    if (dense_out[j] == nullptr) return XPT_ERR_NULL;
    if (bterms[j] < 1 || bterms[j] > MAX_BTERMS) return XPT_ERR_ARG;
    a.d[j].out = dense_out[j];
    a.d[j].nterms = bterms[j];
    for (int k = 0; k < bterms[j]; ++k) {
      if (slice[j * 3 + k] < 0 || slice[j * 3 + k] >= nslices) return XPT_ERR_ARG;
      a.d[j].t[k] = BwdTerm{slice[j * 3 + k], pooled[j * 3 + k], scale[j * 3 + k]};
    }
  }
  for (;;) {
    bool all = F % v == 0 && aligned_for(out, out_pitch, v, esz) && aligned_for(gm, (long long)nslices * F, v, esz);
    for (int k = 0; k < ngrads && all; ++k) all = aligned_for(grads[k], gpitch[k], v, esz);
    for (int j = 0; j < ndense && all; ++j) all = aligned_for(dense_out[j], F, v, esz);
    if (all || v == 1) break;
    v >>= 1;
  }
  const long long total = (long long)B * H * W * (F / v);
  if (total >= (1LL << 31)) return XPT_ERR_SHAPE;      // (the kernels split flat indices in 32 bits)
  const XcdSweep sw = xpt_xcd_sweep(total, 4096);
  const dim3 grid(sw.grid, nslices + ndense);
  hipStream_t st = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
#define XPT_TAIL(T, V) hipLaunchKernelGGL((cell_tail_bwd_kernel<T, V>), grid, dim3(256), 0, st, a, B, H, W, F, sw)
  if (dtype == 0) {
    if (v == 4) XPT_TAIL(float, 4);
    else if (v == 2) XPT_TAIL(float, 2);
    else XPT_TAIL(float, 1);
  } else {
    if (v == 8) XPT_TAIL(xpt_half_t, 8);
    else if (v == 4) XPT_TAIL(xpt_half_t, 4);
    else if (v == 2) XPT_TAIL(xpt_half_t, 2);
    else XPT_TAIL(xpt_half_t, 1);
  }
#undef XPT_TAIL
  return xpt_launch_status();
}

/* "spatial" adjust block: out [B, H2, W2, 2C] (dense) = [p[2i, 2j] | p[2i+1, 2j+1] or 0], H2 = ceil(H/2), W2 = ceil(W/2) */
int xpt_adjust_gather(const void* in, long long in_pitch, void* out, int B, int H, int W, int C, int dtype, void* stream) {
  XPT_CHECK_PTR(in); XPT_CHECK_PTR(out);
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || in_pitch < C) return XPT_ERR_SHAPE;
  if (dtype != 0 && dtype != 1) return XPT_ERR_ARG;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2, esz = dtype == 0 ? 4 : 2;
  int v = dtype == 0 ? 4 : 8;
  while (v > 1 && !(C % v == 0 && aligned_for(in, in_pitch, v, esz) && aligned_for(out, C, v, esz))) v >>= 1;
  const long long total = (long long)B * H2 * W2 * 2 * (C / v);
  if (total >= (1LL << 31)) return XPT_ERR_SHAPE;      // (the kernels split flat indices in 32 bits)
  const XcdSweep sw = xpt_xcd_sweep(total, 8192);
  hipStream_t st = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
#define XPT_K(T, V) \
  hipLaunchKernelGGL((adjust_gather_kernel<T, V>), dim3(sw.grid), dim3(256), 0, st, (const T*)in, in_pitch, (T*)out, B, H, W, C, H2, W2, sw)
  if (dtype == 0) { if (v == 4) XPT_K(float, 4); else if (v == 2) XPT_K(float, 2); else XPT_K(float, 1); }
  else { if (v == 8) XPT_K(xpt_half_t, 8); else if (v == 4) XPT_K(xpt_half_t, 4); else if (v == 2) XPT_K(xpt_half_t, 2); else XPT_K(xpt_half_t, 1); }
#undef XPT_K
  return xpt_launch_status();
}

/* its backward: dp [B, H, W, C] (dense) <- d1 [B, H2, W2, C] at the (even, even) pixels, d2 at the (odd, odd) ones, 0
 * elsewhere; d1 / d2 may be NULL (no gradient through that half) and may have a row pitch */
int xpt_adjust_scatter(const void* d1, long long pitch1, const void* d2, long long pitch2, void* out, int B, int H, int W,
                       int C, int dtype, void* stream) {
  XPT_CHECK_PTR(out);
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (d1 && pitch1 < C) || (d2 && pitch2 < C)) return XPT_ERR_SHAPE;
  if (dtype != 0 && dtype != 1) return XPT_ERR_ARG;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2, esz = dtype == 0 ? 4 : 2;
  int v = dtype == 0 ? 4 : 8;
  while (v > 1 && !(C % v == 0 && aligned_for(out, C, v, esz) && (!d1 || aligned_for(d1, pitch1, v, esz)) &&
                    (!d2 || aligned_for(d2, pitch2, v, esz))))
    v >>= 1;
  const long long total = (long long)B * H * W * (C / v);
  if (total >= (1LL << 31)) return XPT_ERR_SHAPE;      // (the kernels split flat indices in 32 bits)
  const XcdSweep sw = xpt_xcd_sweep(total, 8192);
  hipStream_t st = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
#define XPT_K(T, V)                                                                                                  \
  hipLaunchKernelGGL((adjust_scatter_kernel<T, V>), dim3(sw.grid), dim3(256), 0, st, (const T*)d1, pitch1, (const T*)d2, \
                     pitch2, (T*)out, B, H, W, C, H2, W2, sw)
  if (dtype == 0) { if (v == 4) XPT_K(float, 4); else if (v == 2) XPT_K(float, 2); else XPT_K(float, 1); }
  else { if (v == 8) XPT_K(xpt_half_t, 8); else if (v == 4) XPT_K(xpt_half_t, 4); else if (v == 2) XPT_K(xpt_half_t, 2); else XPT_K(xpt_half_t, 1); }
#undef XPT_K
  return xpt_launch_status();
}

/* max and average 3x3 / stride-2 pooling of the zero-padded h in one launch: mp, ap [B, OH, OW, C] dense, arg (uint8,
 * same shape) = the tap (0..8, row-major) holding the maximum (first one on ties, zero padding included) */
int xpt_pool_pair_fwd(const void* in, long long in_pitch, void* mp, void* ap, void* arg, int B, int H, int W, int C, int OH,
                      int OW, int pad_t, int pad_l, int dtype, void* stream) {
  XPT_CHECK_PTR(in); XPT_CHECK_PTR(mp); XPT_CHECK_PTR(ap); XPT_CHECK_PTR(arg);
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || in_pitch < C) return XPT_ERR_SHAPE;
  if ((dtype != 0 && dtype != 1) || pad_t < 0 || pad_l < 0 || pad_t > 2 || pad_l > 2) return XPT_ERR_ARG;
  if (2 * (OH - 1) - pad_t >= H || 2 * (OW - 1) - pad_l >= W) return XPT_ERR_SHAPE;      // last window starts inside
  const int esz = dtype == 0 ? 4 : 2;
  int v = dtype == 0 ? 4 : 8;
  while (v > 1 && !(C % v == 0 && aligned_for(in, in_pitch, v, esz) && aligned_for(mp, C, v, esz) && aligned_for(ap, C, v, esz)))
    v >>= 1;
  const long long total = (long long)B * OH * OW * (C / v);
  if (total >= (1LL << 31)) return XPT_ERR_SHAPE;      // (the kernels split flat indices in 32 bits)
  const XcdSweep sw = xpt_xcd_sweep(total, 8192);
  hipStream_t st = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
#define XPT_K(T, V)                                                                                                 \
  hipLaunchKernelGGL((pool_pair_fwd_kernel<T, V>), dim3(sw.grid), dim3(256), 0, st, (const T*)in, in_pitch, (T*)mp, \
                     (T*)ap, (unsigned char*)arg, B, H, W, C, OH, OW, pad_t, pad_l, sw)
  if (dtype == 0) { if (v == 4) XPT_K(float, 4); else if (v == 2) XPT_K(float, 2); else XPT_K(float, 1); }
  else { if (v == 8) XPT_K(xpt_half_t, 8); else if (v == 4) XPT_K(xpt_half_t, 4); else if (v == 2) XPT_K(xpt_half_t, 2); else XPT_K(xpt_half_t, 1); }
#undef XPT_K
  return xpt_launch_status();
}

/* backward: dh [B, H, W, C] dense = max-pool gradient routed to the arg-max taps + average-pool gradient / 9; gmp / gap may
 * be NULL and may have a row pitch */
int xpt_pool_pair_bwd(const void* gmp, long long pitch_m, const void* gap, long long pitch_a, const void* arg, void* dh, int B,
                      int H, int W, int C, int OH, int OW, int pad_t, int pad_l, int dtype, void* stream) {
  return xpt_pool_pair_bwd2(gmp, pitch_m, nullptr, 0, gap, pitch_a, arg, dh, B, H, W, C, OH, OW, pad_t, pad_l, dtype, stream);
}

int xpt_pool_pair_bwd2(const void* gmp, long long pitch_m, const void* gmp2, long long pitch_m2, const void* gap, long long pitch_a,
                       const void* arg, void* dh, int B, int H, int W, int C, int OH, int OW, int pad_t, int pad_l, int dtype,
                       void* stream) {
  XPT_CHECK_PTR(arg); XPT_CHECK_PTR(dh);
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || (gmp && pitch_m < C) || (gap && pitch_a < C) ||
      (gmp2 && (pitch_m2 < C || !gmp)))
    return XPT_ERR_SHAPE;
  if ((dtype != 0 && dtype != 1) || pad_t < 0 || pad_l < 0 || pad_t > 2 || pad_l > 2) return XPT_ERR_ARG;
  const int esz = dtype == 0 ? 4 : 2;
  int v = dtype == 0 ? 4 : 8;
  while (v > 1 && !(C % v == 0 && aligned_for(dh, C, v, esz) && (!gmp || aligned_for(gmp, pitch_m, v, esz)) &&
                    (!gmp2 || aligned_for(gmp2, pitch_m2, v, esz)) && (!gap || aligned_for(gap, pitch_a, v, esz))))
    v >>= 1;
  const long long total = (long long)B * H * W * (C / v);
  if (total >= (1LL << 31)) return XPT_ERR_SHAPE;      // (the kernels split flat indices in 32 bits)
  const XcdSweep sw = xpt_xcd_sweep(total, 8192);
  hipStream_t st = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
#define XPT_K(T, V)                                                                                                    \
  hipLaunchKernelGGL((pool_pair_bwd_kernel<T, V>), dim3(sw.grid), dim3(256), 0, st, (const T*)gmp, pitch_m, (const T*)gmp2, \
                     pitch_m2, (const T*)gap, pitch_a, (const unsigned char*)arg, (T*)dh, B, H, W, C, OH, OW, pad_t, pad_l, sw)
  if (dtype == 0) { if (v == 4) XPT_K(float, 4); else if (v == 2) XPT_K(float, 2); else XPT_K(float, 1); }
  else { if (v == 8) XPT_K(xpt_half_t, 8); else if (v == 4) XPT_K(xpt_half_t, 4); else if (v == 2) XPT_K(xpt_half_t, 2); else XPT_K(xpt_half_t, 1); }
#undef XPT_K
  return xpt_launch_status();
}

}  // extern "C"
