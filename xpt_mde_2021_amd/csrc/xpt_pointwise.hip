// xpt_pointwise.hip -- per-channel epilogues of the convolution layers on NHWC activations for gfx950:
//   bias + LeakyReLU / linear      (CustomConv2D layers of DepthNet decoder and PoseNet: model/model_util/layer_ops.py:31-35,
//                                   activation LeakyReLU(0.1) per config-example.py:56-63)
//   inference-mode BatchNorm affine (+ ReLU on the way in) of the NASNet cells (keras BatchNormalization called without
//                                   training=True, model/train_val.py:82; reference call site pretrained_nets.py:36-44)
// Both are one streaming pass forward and one streaming pass backward (dx plus the per-channel parameter gradients
// reduced deterministically: per-workgroup partials, then a fixed-order sum) instead of the 2-5 library launches per
// layer (bias add, activation, their backward kernels and a separate reduction) of the generic framework path.

#include "xpt_common.h"

namespace {

template <typename T> __device__ inline float ldf(const T* p);
template <> __device__ inline float ldf<float>(const float* p) { return *p; }
template <> __device__ inline float ldf<xpt_half_t>(const xpt_half_t* p) { return xpt_half2float(*p); }
template <typename T> __device__ inline void stf(T* p, float v);
template <> __device__ inline void stf<float>(float* p, float v) { *p = v; }
template <> __device__ inline void stf<xpt_half_t>(xpt_half_t* p, float v) { *p = xpt_float2half(v); }

// ---------------------------------------------------------------- y = act(f(x) * scale[c] + shift[c])
// scale = gamma * rsqrt(var + eps), shift = beta - mean * scale (gamma == nullptr: scale = 1, shift = beta).
// slope: LeakyReLU negative slope (1 = linear, 0 = ReLU).  relu_in: f(x) = max(x, 0), else f(x) = x.
struct Affine {
  const float* gamma;
  const float* beta;
  const float* mean;
  const float* var;
  float eps;
};

__device__ inline void affine_coeffs(const Affine& a, int c, float& sc, float& sh, float& rstd, float& mu) {
  if (a.gamma) {
    rstd = rsqrtf(a.var[c] + a.eps);
    mu = a.mean[c];
    sc = a.gamma[c] * rstd;
    sh = a.beta[c] - mu * sc;
  } else {
    rstd = 1.f; mu = 0.f; sc = 1.f; sh = a.beta[c];
  }
}

template <typename T>
__global__ void affine_act_fwd_kernel(const T* __restrict__ x, Affine a, const T* __restrict__ residual,
                                      T* __restrict__ y, long long n, int C, float slope, int relu_in, XcdSweep sw) {
  long long i, i_end;
  if (!sw.range(blockIdx.x, n, i, i_end)) return;      // (pixel-major indices: image-to-XCD numbering, xpt_common.h)
  for (i += threadIdx.x; i < i_end; i += 256) {
    unsigned c_;
    (void)xpt_divmod((unsigned)i, (unsigned)C, c_);      // (n < 2^31: checked by the launcher)
    const int c = (int)c_;
    float sc, sh, rstd, mu;
    affine_coeffs(a, c, sc, sh, rstd, mu);
    float v = ldf<T>(x + i);
    if (relu_in) v = fmaxf(v, 0.f);
    v = v * sc + sh;
    v = v > 0.f ? v : v * slope;
    if (residual) v += ldf<T>(residual + i);     // the cell's branch sum (keras layers.add), linear epilogues only
    stf<T>(y + i, v);
  }
}

// V consecutive channels of one pixel row as one load / store
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

// Backward.  A thread owns V consecutive channels (one 16-byte load per tensor and pixel when the layout allows) and
// walks the pixel rows of its workgroup's slice, keeping its 2 V running sums in registers; the row slots of a
// workgroup are then folded through LDS in slot order (deterministic).
//   g  = dy * act'(y)            (act' from the OUTPUT sign: valid for slope > 0, and for slope == 0 where y == 0 => 0)
//   dx = g * scale [* (x > 0) if relu_in]
//   part[blk][0][c] = sum g  (dbeta);  part[blk][1][c] = sum g f(x)  or, with final_partials, the block's share of
//   dgamma = rstd * (sum g f(x) - mean * sum g).
// grid (channel tiles of 64 V-groups, row slices); dy may have a row pitch (a channel slice of a wider tensor).
#define PW_MAX_GROUPS 64
template <typename T, int V>
__global__ __launch_bounds__(256) void affine_act_bwd_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                              const T* __restrict__ dy, long long dy_pitch, Affine a,
                                                              T* __restrict__ dx, float* __restrict__ part,
                                                              long long rows, int C, long long rows_per_block,
                                                              float slope, int relu_in, int need_dscale,
                                                              int final_partials, int nblocks, int xcd) {
  __shared__ float red[2][256 * V];
  unsigned rb;                                                // grid: x = row slices (image-to-XCD numbering), y = channel tiles
  if (!xpt_xcd_unit(xcd != 0, blockIdx.x, (unsigned)nblocks, rb)) return;
  const int groups = C / V;                                   // V divides C
  const int g0 = blockIdx.y * PW_MAX_GROUPS;
  const int gt = min(groups - g0, PW_MAX_GROUPS);             // V-groups of this channel tile
  const int slots = 256 / gt;                                 // pixel rows in flight per iteration
  const int grp = threadIdx.x % gt, slot = threadIdx.x / gt;
  const int c0 = (g0 + grp) * V;
  const long long r_begin = (long long)rb * rows_per_block;
  const long long r_end = min(rows, r_begin + rows_per_block);
  float s_shift[V], s_scale[V], sc[V];
#pragma unroll
  for (int i = 0; i < V; ++i) s_shift[i] = s_scale[i] = 0.f;
  if (slot < slots) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
      float sh, rstd, mu;
      affine_coeffs(a, c0 + i, sc[i], sh, rstd, mu);
    }
#pragma unroll 2
    for (long long r = r_begin + slot; r < r_end; r += slots) {
      const long long o = r * C + c0;
      float g[V], xv[V], gx[V];
      load_row<T, V>(dy + r * dy_pitch + c0, g);
      if (slope != 1.f) {                                     // y is not read for a linear epilogue
        float yv[V];
        load_row<T, V>(y + o, yv);
#pragma unroll
        for (int i = 0; i < V; ++i) g[i] = yv[i] > 0.f ? g[i] : g[i] * slope;
      }
      if (need_dscale || relu_in) {
        load_row<T, V>(x + o, xv);
      } else {
#pragma unroll
        for (int i = 0; i < V; ++i) xv[i] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < V; ++i) {
        gx[i] = g[i] * sc[i];
        if (relu_in) {
          if (!(xv[i] > 0.f)) gx[i] = 0.f;
          xv[i] = fmaxf(xv[i], 0.f);
        }
        s_shift[i] += g[i];
        s_scale[i] += g[i] * xv[i];
      }
      if (dx) store_row<T, V>(dx + o, gx);
    }
  }
  // fold the row slots: red[q][slot][grp * V + i]
  const int width = gt * V;
  if (slot < slots) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
      red[0][slot * width + grp * V + i] = s_shift[i];
      red[1][slot * width + grp * V + i] = s_scale[i];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < width; e += 256) {
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < slots; ++k) {
      s0 += red[0][k * width + e];
      s1 += red[1][k * width + e];
    }
    const int c = g0 * V + e;
    if (final_partials && need_dscale) {   // this block's share of dgamma itself (the finishing pass only adds)
      float scc, sh, rstd, mu;
      affine_coeffs(a, c, scc, sh, rstd, mu);
      s1 = rstd * (s1 - mu * s0);
    }
    float* p = part + (long long)rb * 2 * C;
    p[c] = s0;
    p[C + c] = s1;
  }
}

// out[r, c] = sum_i in_i[r * pitch_i + c]: the gradients that reach a tensor feeding several branches of a NASNet cell
// (autograd would add them pairwise, one launch per extra consumer); inputs may be channel slices (row pitch).
struct SumInputs {
  const void* ptr[8];
  long long pitch[8];
  int n;
};

template <typename T, int V>
__global__ __launch_bounds__(256) void sum_rows_kernel(SumInputs in, T* __restrict__ out, long long rows, int C, XcdSweep sw) {
  const int groups = C / V;
  const long long total = rows * groups;
  long long idx, idx_end;
  if (!sw.range(blockIdx.x, total, idx, idx_end)) return;      // (pixel-major indices: image-to-XCD numbering, xpt_common.h)
  for (idx += threadIdx.x; idx < idx_end; idx += 256) {
    unsigned c0_;
    const long long r = (long long)xpt_divmod((unsigned)idx, (unsigned)groups, c0_);
    const int c0 = (int)c0_ * V;
    // all (up to 8) operand rows are requested before the first add: unused slots re-read operand 0 and add zero
    float acc[V], v[7][V];
    load_row<T, V>((const T*)in.ptr[0] + r * in.pitch[0] + c0, acc);
#pragma unroll
    for (int i = 1; i < 8; ++i) {
      const int k = i < in.n ? i : 0;
      load_row<T, V>((const T*)in.ptr[k] + r * in.pitch[k] + c0, v[i - 1]);
    }
#pragma unroll
    for (int i = 1; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] += i < in.n ? v[i - 1][j] : 0.f;
    store_row<T, V>(out + r * C + c0, acc);
  }
}

// ---------------------------------------------------------------- several device-to-device copies in one launch
// The captured training step reads its batch from static buffers (model/train_val.py: _StepGraph); a new batch is copied
// into them in front of every replay: the snippet tensor (16 MB at batch 8) and half a dozen small ones.  One launch whose
// workgroups walk the concatenation of all byte ranges in 16-byte vectors (every range 16-byte aligned and a multiple of
// 16 bytes long, or copied byte-wise by its first workgroup otherwise); torch's multi-tensor copy took 25 us for them.
struct CopyJobs {
  const unsigned char* src[8];
  unsigned char* dst[8];
  long long first[9];        // first 16-byte vector of job i in the concatenated index space; first[n] = total
  long long bytes[8];
  int n;
};

__global__ __launch_bounds__(256) void multi_copy_kernel(CopyJobs jobs) {
  const long long total = jobs.first[jobs.n];
  for (long long v = blockIdx.x * 256LL + threadIdx.x; v < total; v += gridDim.x * 256LL) {
    int j = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
      if (i < jobs.n && v >= jobs.first[i]) j = i;
    const long long off = (v - jobs.first[j]) * 16;
    if (off + 16 <= jobs.bytes[j] && ((((uintptr_t)jobs.src[j]) | ((uintptr_t)jobs.dst[j])) & 15) == 0) {
      *(uint4*)(jobs.dst[j] + off) = *(const uint4*)(jobs.src[j] + off);
    } else {
      for (long long b = off; b < off + 16 && b < jobs.bytes[j]; ++b) jobs.dst[j][b] = jobs.src[j][b];
    }
  }
}

// ---------------------------------------------------------------- channel concatenation (bf16)
// out [rows, Ct] (dense, Ct a multiple of 8) = [in_0 | in_1 | ... | zeros]: the decoder's concat([up-convolution, skip,
// up-sampled previous prediction]) (model/build_model/depth_net.py:104-107) with the zero channels that pad it to the
// 8-channel groups the matrix-core convolution reads.  A thread moves one 16-byte group of the output: one 16-byte load
// when the group lies inside one input on a 16-byte boundary, element-wise otherwise (the one-channel prediction, the
// pad).  torch.cat's batched copy moved these 5 - 20 MB at 0.5 - 0.8 TB/s.
struct CatInputs {
  const unsigned short* ptr[4];
  long long pitch[4];
  int first[5];              // first[i] = output channel where input i starts; first[n] = total real channels
  int n;
};

__global__ __launch_bounds__(256) void concat_channels_kernel(CatInputs in, unsigned short* __restrict__ out, unsigned total,
                                                              int groups, XcdSweep sw) {
  long long i64, e64;
  if (!sw.range(blockIdx.x, (long long)total, i64, e64)) return;      // (pixel-major indices: image-to-XCD numbering, xpt_common.h)
  for (unsigned idx = (unsigned)i64 + threadIdx.x; idx < (unsigned)e64; idx += 256u) {
    unsigned g_;
    const long long row = (long long)xpt_divmod(idx, (unsigned)groups, g_);
    const int c0 = (int)g_ * 8;
    uint4 val = make_uint4(0u, 0u, 0u, 0u);
    bool done = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < in.n && c0 >= in.first[i] && c0 + 8 <= in.first[i + 1]) {
        const unsigned short* src = in.ptr[i] + row * in.pitch[i] + (c0 - in.first[i]);
        if (((uintptr_t)src & 15) == 0) {
          val = *(const uint4*)src;
          done = true;
        }
      }
    }
    if (!done) {
      unsigned short e[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = c0 + k;
        unsigned short v = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < in.n && c >= in.first[i] && c < in.first[i + 1]) v = in.ptr[i][row * in.pitch[i] + (c - in.first[i])];
        e[k] = v;
      }
      val.x = e[0] | ((unsigned)e[1] << 16); val.y = e[2] | ((unsigned)e[3] << 16);
      val.z = e[4] | ((unsigned)e[5] << 16); val.w = e[6] | ((unsigned)e[7] << 16);
    }
    *(uint4*)(out + (row * groups + (long long)g_) * 8) = val;
  }
}

// 3x3 stride-1 SAME average pooling whose divisor excludes the padding (keras AveragePooling2D((3,3), strides 1,
// padding='same') of the NASNet cells), times `scale`.  adjoint == 0: out = scale / cnt(out px) * sum of the valid
// neighbours; adjoint == 1 (the backward): out = sum over the valid neighbours n of in[n] * scale / cnt(n).
// The input may be a channel slice of a wider tensor (row pitch); loads are unconditional on clamped coordinates.
template <typename T, int V>
__global__ __launch_bounds__(256) void avgpool3_kernel(const T* __restrict__ in, long long in_pitch, T* __restrict__ out,
                                                       int B, int H, int W, int C, float scale, int adjoint) {
  const int groups = C / V;
  const long long total = (long long)B * H * W * groups;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    unsigned c0_;
    long long p = (long long)xpt_divmod((unsigned)idx, (unsigned)groups, c0_);
    const int c0 = (int)c0_ * V;
    unsigned r1_, r2_;
    const unsigned q1_ = xpt_divmod((unsigned)p, (unsigned)W, r1_);
    const int b = (int)xpt_divmod(q1_, (unsigned)H, r2_);
    const int x = (int)r1_, y = (int)r2_;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int ny = y + dy, nx = x + dx;
        const bool ok = ny >= 0 && ny < H && nx >= 0 && nx < W;
        const int cy = min(max(ny, 0), H - 1), cx = min(max(nx, 0), W - 1);
        float v[V];
        load_row<T, V>(in + (((long long)b * H + cy) * W + cx) * in_pitch + c0, v);
        float wgt = ok ? 1.f : 0.f;
        if (adjoint) {      // weight of neighbour n: 1 / (valid taps of n's own window)
          const int cnt = ((cy > 0) + 1 + (cy < H - 1)) * ((cx > 0) + 1 + (cx < W - 1));
          wgt = ok ? 1.f / (float)cnt : 0.f;
        }
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += v[j] * wgt;
      }
    }
    float norm = scale;
    if (!adjoint) norm = scale / (float)(((y > 0) + 1 + (y < H - 1)) * ((x > 0) + 1 + (x < W - 1)));
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] *= norm;
    store_row<T, V>(out + (((long long)b * H + y) * W + x) * C + c0, acc);
  }
}

// Depth head activation (model_factory.py:134-138 InverseSigmoid + util_funcs.py:157-160 safe_reciprocal_number):
//   depth = safe_rcp(sigmoid(x) + 0.01),  disp = safe_rcp(depth),  safe_rcp(v) = (1 / v) * [v > 1e-5]
// and its backward gx = d depth/dx * (g_depth + d disp/d depth * g_disp): one launch each instead of ~10 / ~7
// elementwise launches per scale.
__global__ void depth_head_fwd_kernel(const float* __restrict__ x, float* __restrict__ depth, float* __restrict__ disp,
                                      long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float sg = 1.f / (1.f + expf(-x[i]));
    const float u = sg + 0.01f;
    const float d = u > 1e-5f ? 1.f / u : 0.f;
    depth[i] = d;
    disp[i] = d > 1e-5f ? 1.f / d : 0.f;
  }
}

__global__ void depth_head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g_depth,
                                      const float* __restrict__ g_disp, float* __restrict__ gx, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float sg = 1.f / (1.f + expf(-x[i]));
    const float u = sg + 0.01f;
    const float d = u > 1e-5f ? 1.f / u : 0.f;
    float gd = g_depth ? g_depth[i] : 0.f;
    if (g_disp && d > 1e-5f) gd -= g_disp[i] / (d * d);
    const float gu = u > 1e-5f ? -gd / (u * u) : 0.f;
    gx[i] = gu * sg * (1.f - sg);
  }
}

// ---------------------------------------------------------------- GlobalAveragePooling2D of an NHWC map (PoseNet's tail)
template <typename T>
__global__ void gap_fwd_kernel(const T* __restrict__ x, float* __restrict__ y, int total, int HW, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (b, c)
  if (i >= total) return;
  const int b = i / C, c = i - b * C;
  const T* p = x + (long long)b * HW * C + c;
  float s = 0.f;
  for (int k = 0; k < HW; ++k) s += ldf<T>(p + (long long)k * C);
  y[i] = s / (float)HW;
}

template <typename T>
__global__ void gap_bwd_kernel(const float* __restrict__ g, T* __restrict__ dx, long long total, int HW, int C) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    unsigned c_;
    (void)xpt_divmod((unsigned)i, (unsigned)C, c_);      // (n < 2^31: checked by the launcher)
    const int c = (int)c_;
    const long long b = i / ((long long)HW * C);
    stf<T>(dx + i, g[b * C + c] / (float)HW);
  }
}

// ---------------------------------------------------------------- exact 2x bilinear up-sampling of one-channel maps
// tf.image.resize(bilinear, half-pixel centres) at an exact factor 2 = torch upsample_bilinear2d(align_corners=False):
// source coordinate of output y is max(y / 2 - 0.25, 0): taps (i0, min(i0 + 1, h - 1)) with weights (1 - l, l).
__device__ inline void up2_taps(int y, int h, int& i0, int& i1, float& l) {
  const float src = fmaxf(0.5f * (float)y - 0.25f, 0.f);
  i0 = (int)src;
  i1 = min(i0 + 1, h - 1);
  l = src - (float)i0;
}

template <typename T>
__global__ void upsample2x_fwd_kernel(const float* __restrict__ src, T* __restrict__ out, long long total, int h, int w,
                                      XcdSweep sw) {
  const int H = 2 * h, W = 2 * w;
  long long i, i_end;
  if (!sw.range(blockIdx.x, total, i, i_end)) return;      // (pixel-major indices: image-to-XCD numbering, xpt_common.h)
  for (i += threadIdx.x; i < i_end; i += 256) {
    unsigned xu, yu;                                // (total < 2^31: the launcher checks; 64-bit divisions are ~120 instructions each)
    const long long m = (long long)xpt_divmod(xpt_divmod((unsigned)i, (unsigned)W, xu), (unsigned)H, yu);
    const int x = (int)xu, y = (int)yu;
    int y0, y1, x0, x1;
    float ly, lx;
    up2_taps(y, h, y0, y1, ly);
    up2_taps(x, w, x0, x1, lx);
    const float* p = src + m * h * w;
    const float top = (1.f - lx) * p[(long long)y0 * w + x0] + lx * p[(long long)y0 * w + x1];
    const float bot = (1.f - lx) * p[(long long)y1 * w + x0] + lx * p[(long long)y1 * w + x1];
    stf<T>(out + i, (1.f - ly) * top + ly * bot);
  }
}

// weight of input i in output row y (0 when y is outside or i is not one of its taps); the adjoint of up2_taps
__device__ inline float up2_weight(int y, int i, int h) {
  if (y < 0 || y >= 2 * h) return 0.f;
  int i0, i1;
  float l;
  up2_taps(y, h, i0, i1, l);
  return (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
}

// gather form of the backward: input (i, j) collects its (at most) 4 x 4 outputs; g has a pixel pitch (a channel slice
// of an NHWC concatenation gradient is read in place)
template <typename T>
__global__ void upsample2x_bwd_kernel(const T* __restrict__ g, long long g_pitch, const float* __restrict__ addend,
                                      float* __restrict__ dsrc, long long total, int h, int w, XcdSweep sw) {
  const int W = 2 * w, H = 2 * h;
  long long idx, idx_end;
  if (!sw.range(blockIdx.x, total, idx, idx_end)) return;      // (pixel-major indices: image-to-XCD numbering, xpt_common.h)
  for (idx += threadIdx.x; idx < idx_end; idx += 256) {
    unsigned ju, iu;
    const long long m = (long long)xpt_divmod(xpt_divmod((unsigned)idx, (unsigned)w, ju), (unsigned)h, iu);
    const int j = (int)ju, i = (int)iu;
    const T* gm = g + m * H * W * g_pitch;
    float acc = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 2; ++dy) {
      const int y = 2 * i + dy;
      const float wy = up2_weight(y, i, h);
      if (wy == 0.f) continue;
      float row = 0.f;
#pragma unroll
      for (int dx = -1; dx <= 2; ++dx) {
        const int x = 2 * j + dx;
        const float wx = up2_weight(x, j, w);
        if (wx != 0.f) row += wx * ldf<T>(gm + ((long long)y * W + x) * g_pitch);
      }
      acc += wy * row;
    }
    dsrc[idx] = addend != nullptr ? addend[idx] + acc : acc;      // (the gradient the other consumer of src left: one fan-in launch less)
  }
}

// The four prediction scales of the decoder in one launch (blockIdx.y = scale; a grid-stride loop inside the scale).
struct DepthHeadMs {
  const float* x[4];
  float* depth[4];       // backward: gx
  float* disp[4];
  const float* g_depth[4];
  const float* g_disp[4];
  long long n[4];
};

__global__ void depth_head_ms_fwd_kernel(DepthHeadMs a) {
  const int s = blockIdx.y;
  const float* __restrict__ x = a.x[s];
  float* __restrict__ depth = a.depth[s];
  float* __restrict__ disp = a.disp[s];
  const long long n = a.n[s];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float sg = 1.f / (1.f + expf(-x[i]));
    const float u = sg + 0.01f;
    const float d = u > 1e-5f ? 1.f / u : 0.f;
    depth[i] = d;
    disp[i] = d > 1e-5f ? 1.f / d : 0.f;
  }
}

__global__ void depth_head_ms_bwd_kernel(DepthHeadMs a) {
  const int s = blockIdx.y;
  const float* __restrict__ x = a.x[s];
  const float* __restrict__ g_depth = a.g_depth[s];
  const float* __restrict__ g_disp = a.g_disp[s];
  float* __restrict__ gx = a.depth[s];
  const long long n = a.n[s];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float sg = 1.f / (1.f + expf(-x[i]));
    const float u = sg + 0.01f;
    const float d = u > 1e-5f ? 1.f / u : 0.f;
    float gd = g_depth ? g_depth[i] : 0.f;
    if (g_disp && d > 1e-5f) gd -= g_disp[i] / (d * d);
    const float gu = u > 1e-5f ? -gd / (u * u) : 0.f;
    gx[i] = gu * sg * (1.f - sg);
  }
}

// per channel c: S0 = sum_k part[k][0][c], S1 = sum_k part[k][1][c]; one wave (64 threads) per channel so that the
// partial rows are fetched with few dependent round trips; fixed tree -> deterministic.
__global__ void affine_finish_kernel(const float* __restrict__ part, Affine a, float* __restrict__ dbeta,
                                     float* __restrict__ dgamma, int C, int nblk) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int t = threadIdx.x & 63;
  float s0 = 0.f, s1 = 0.f;
  if (c < C)
    for (int k = t; k < nblk; k += 64) {
      s0 += part[(long long)k * 2 * C + c];
      s1 += part[(long long)k * 2 * C + C + c];
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s0 += __shfl_down(s0, off, 64);
    s1 += __shfl_down(s1, off, 64);
  }
  if (t == 0 && c < C) {
    dbeta[c] = s0;
    if (dgamma) {
      float sc, sh, rstd, mu;
      affine_coeffs(a, c, sc, sh, rstd, mu);
      dgamma[c] = rstd * (s1 - mu * s0);
    }
  }
}

inline unsigned grid_for(long long total) {
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

// pixel rows per workgroup: ~512 row slices per launch (small maps: at least 8 rows each, so that even a 4x13 map
// spreads over ~50 workgroups instead of walking its rows in a few long dependent loops)
static long long affine_bwd_rows_per_block(long long rows) {
  long long rpb = (rows + 511) / 512;
  if (rpb < 8) rpb = 8;
  return (rpb + 7) / 8 * 8;
}

static int affine_bwd_blocks(long long rows, int C) {
  const long long rpb = affine_bwd_rows_per_block(rows);
  return (int)((rows + rpb - 1) / rpb);
}

template <typename T, int V>
static void affine_bwd_launch_v(const void* x, const void* y, const void* dy, long long dy_pitch, const Affine& a,
                                void* dx, float* part, long long rows, int C, float slope, int relu_in,
                                int need_dscale, int final_partials, hipStream_t s) {
  const int groups = C / V;
  const int nblocks = affine_bwd_blocks(rows, C), xcd = g_xpt_xcd_affinity && nblocks >= 8;
  const dim3 grid(xcd ? xpt_xcd_pad(nblocks) : nblocks, (groups + PW_MAX_GROUPS - 1) / PW_MAX_GROUPS);
  hipLaunchKernelGGL((affine_act_bwd_kernel<T, V>), grid, dim3(256), 0, s, (const T*)x, (const T*)y, (const T*)dy,
                     dy_pitch, a, (T*)dx, part, rows, C, affine_bwd_rows_per_block(rows), slope, relu_in, need_dscale,
                     final_partials, nblocks, xcd);
}

// dx + per-block partial sums part[blk][2][C]; final_partials: the second row holds the block's share of dgamma
static void affine_bwd_launch(const void* x, const void* y, const void* dy, long long dy_pitch, const Affine& a,
                              void* dx, float* part, long long rows, int C, float slope, int relu_in, int need_dscale,
                              int final_partials, int dtype, hipStream_t s) {
  // widest vector (elements) every tensor involved allows
  const int esz = dtype == 0 ? 4 : 2;
  int v = dtype == 0 ? 4 : 8;
  auto ok = [&](const void* p, long long pitch) {
    return p == nullptr || (((uintptr_t)p) % (size_t)(v * esz) == 0 && pitch % v == 0);
  };
  while (v > 1 && !(C % v == 0 && ok(x, C) && ok(y, C) && ok(dy, dy_pitch) && ok(dx, C))) v >>= 1;
#define XPT_AFF(T, V) \
  affine_bwd_launch_v<T, V>(x, y, dy, dy_pitch, a, dx, part, rows, C, slope, relu_in, need_dscale, final_partials, s)
  if (dtype == 0) {
    if (v == 4) XPT_AFF(float, 4);
    else if (v == 2) XPT_AFF(float, 2);
    else XPT_AFF(float, 1);
  } else {
    if (v == 8) XPT_AFF(xpt_half_t, 8);
    else if (v == 4) XPT_AFF(xpt_half_t, 4);
    else if (v == 2) XPT_AFF(xpt_half_t, 2);
    else XPT_AFF(xpt_half_t, 1);
  }
#undef XPT_AFF
}

static int affine_bwd_check(const void* x, const void* y, const void* dy, const float* gamma, const float* beta,
                            const float* mean, const float* var, long long rows, int C, float slope, int relu_in,
                            int dtype) {
  XPT_CHECK_PTR(dy); XPT_CHECK_PTR(beta);
  if (slope != 1.f) XPT_CHECK_PTR(y);
  if (rows <= 0 || C <= 0) return XPT_ERR_SHAPE;
  if (dtype != 0 && dtype != 1) return XPT_ERR_ARG;
  if (gamma && (!mean || !var)) return XPT_ERR_NULL;
  if ((gamma || relu_in) && !x) return XPT_ERR_NULL;
  return XPT_OK;
}

}  // namespace

extern "C" {

int xpt_affine_act_fwd(const void* x, const float* gamma, const float* beta, const float* mean, const float* var,
                       float eps, const void* residual, void* y, long long rows, int C, float slope, int relu_in,
                       int dtype, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(beta); XPT_CHECK_PTR(y);
  if (residual && slope != 1.f) return XPT_ERR_ARG;   // the backward reads act' off the output sign
  if (gamma && (!mean || !var)) return XPT_ERR_NULL;
  if (rows <= 0 || C <= 0) return XPT_ERR_SHAPE;
  if (dtype != 0 && dtype != 1) return XPT_ERR_ARG;
  const long long n = rows * C;
  if (n >= (1LL << 31)) return XPT_ERR_SHAPE;            // (the kernel splits flat indices in 32 bits)
  const Affine a{gamma, beta, mean, var, eps};
  const XcdSweep sw = xpt_xcd_sweep(n, 4096);
  XPT_BEGIN_LAUNCH();
  if (dtype == 0)
    hipLaunchKernelGGL(affine_act_fwd_kernel<float>, dim3(sw.grid), dim3(256), 0, (hipStream_t)stream,
                       (const float*)x, a, (const float*)residual, (float*)y, n, C, slope, relu_in, sw);
  else
    hipLaunchKernelGGL(affine_act_fwd_kernel<xpt_half_t>, dim3(sw.grid), dim3(256), 0, (hipStream_t)stream,
                       (const xpt_half_t*)x, a, (const xpt_half_t*)residual, (xpt_half_t*)y, n, C, slope,
                       relu_in, sw);
  return xpt_launch_status();
}

size_t xpt_affine_act_bwd_workspace_floats(long long rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  return (size_t)affine_bwd_blocks(rows, C) * 2 * (size_t)C;
}

/* dx may be NULL (no data gradient wanted); dgamma must be NULL iff gamma is NULL (bias-only epilogue). */
int xpt_affine_act_bwd(const void* x, const void* y, const void* dy, long long dy_pitch, const float* gamma,
                       const float* beta, const float* mean, const float* var, float eps, void* dx, float* dbeta,
                       float* dgamma,
                       float* workspace, size_t workspace_floats, long long rows, int C, float slope, int relu_in,
                       int dtype, void* stream) {
  XPT_CHECK_PTR(dbeta); XPT_CHECK_PTR(workspace);
  const int rc = affine_bwd_check(x, y, dy, gamma, beta, mean, var, rows, C, slope, relu_in, dtype);
  if (rc != XPT_OK) return rc;
  if ((gamma == nullptr) != (dgamma == nullptr)) return XPT_ERR_ARG;
  if (workspace_floats < xpt_affine_act_bwd_workspace_floats(rows, C)) return XPT_ERR_WORKSPACE;
  const Affine a{gamma, beta, mean, var, eps};
  hipStream_t s = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
  if (dy_pitch < C) return XPT_ERR_SHAPE;
  affine_bwd_launch(x, y, dy, dy_pitch, a, dx, workspace, rows, C, slope, relu_in, dgamma != nullptr, 0, dtype, s);
  hipLaunchKernelGGL(affine_finish_kernel, dim3((C + 3) / 4), dim3(256), 0, s, workspace, a, dbeta, dgamma, C,
                     affine_bwd_blocks(rows, C));
  return xpt_launch_status();
}

int xpt_affine_act_bwd_blocks(long long rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  return affine_bwd_blocks(rows, C);
}

/* Deferred parameter gradients: dx as above; partials[blk][0][c] = this block's share of dbeta[c] and, when gamma is
 * given, partials[blk][1][c] = its share of dgamma[c] -- to be added up later by xpt_reduce_partials. */
int xpt_affine_act_bwd_partials(const void* x, const void* y, const void* dy, long long dy_pitch, const float* gamma,
                                const float* beta, const float* mean, const float* var, float eps, void* dx,
                                float* partials,
                                size_t partial_floats, long long rows, int C, float slope, int relu_in, int dtype,
                                void* stream) {
  XPT_CHECK_PTR(partials);
  const int rc = affine_bwd_check(x, y, dy, gamma, beta, mean, var, rows, C, slope, relu_in, dtype);
  if (rc != XPT_OK) return rc;
  if (partial_floats < (size_t)affine_bwd_blocks(rows, C) * 2 * (size_t)C) return XPT_ERR_WORKSPACE;
  const Affine a{gamma, beta, mean, var, eps};
  XPT_BEGIN_LAUNCH();
  if (dy_pitch < C) return XPT_ERR_SHAPE;
  affine_bwd_launch(x, y, dy, dy_pitch, a, dx, partials, rows, C, slope, relu_in, gamma != nullptr, 1, dtype,
                    (hipStream_t)stream);
  return xpt_launch_status();
}

/* depth = safe_rcp(sigmoid(x) + 0.01), disp = safe_rcp(depth) (float32, n elements); backward: g_depth / g_disp may be NULL. */
int xpt_depth_head_fwd(const float* x, float* depth, float* disp, long long n, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(depth); XPT_CHECK_PTR(disp);
  if (n <= 0) return XPT_ERR_SHAPE;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(depth_head_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, depth, disp, n);
  return xpt_launch_status();
}

int xpt_depth_head_bwd(const float* x, const float* g_depth, const float* g_disp, float* gx, long long n, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(gx);
  if (!g_depth && !g_disp) return XPT_ERR_NULL;
  if (n <= 0) return XPT_ERR_SHAPE;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(depth_head_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, g_depth, g_disp, gx, n);
  return xpt_launch_status();
}

/* GlobalAveragePooling2D (pose_net.py:45) of an NHWC map x [B,HW,C] (dtype 0 float32 / 1 bfloat16) -> y float32 [B,C];
 * bwd: g float32 [B,C] -> dx [B,HW,C] of that dtype = g / HW. */
int xpt_global_avgpool_fwd(const void* x, float* y, int B, int HW, int C, int dtype, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(y);
  if (B <= 0 || HW <= 0 || C <= 0) return XPT_ERR_SHAPE;
  if (dtype != 0 && dtype != 1) return XPT_ERR_ARG;
  const int total = B * C;
  XPT_BEGIN_LAUNCH();
  if (dtype == 0)
    hipLaunchKernelGGL(gap_fwd_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const float*)x, y, total, HW, C);
  else
    hipLaunchKernelGGL(gap_fwd_kernel<xpt_half_t>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const xpt_half_t*)x, y, total, HW, C);
  return xpt_launch_status();
}

int xpt_global_avgpool_bwd(const float* g, void* dx, int B, int HW, int C, int dtype, void* stream) {
  XPT_CHECK_PTR(g); XPT_CHECK_PTR(dx);
  if (B <= 0 || HW <= 0 || C <= 0) return XPT_ERR_SHAPE;
  if (dtype != 0 && dtype != 1) return XPT_ERR_ARG;
  const long long total = (long long)B * HW * C;
  if (total >= (1LL << 31)) return XPT_ERR_SHAPE;        // (the kernel splits flat indices in 32 bits)
  XPT_BEGIN_LAUNCH();
  if (dtype == 0)
    hipLaunchKernelGGL(gap_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, g, (float*)dx,
                       total, HW, C);
  else
    hipLaunchKernelGGL(gap_bwd_kernel<xpt_half_t>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, g,
                       (xpt_half_t*)dx, total, HW, C);
  return xpt_launch_status();
}

/* lo.resize_image (layer_ops.py:43-50) at an exact factor 2 on one-channel maps: src float32 [M,h,w] -> out [M,2h,2w]
 * (dtype 0 float32 / 1 bfloat16: the decoder concatenates it with bf16 features); bwd: g [M,2h,2w] with a pixel pitch of
 * g_pitch elements (dtype as above) -> dsrc float32 [M,h,w]. */
int xpt_upsample2x_fwd(const float* src, void* out, long long M, int h, int w, int dtype, void* stream) {
  XPT_CHECK_PTR(src); XPT_CHECK_PTR(out);
  if (M <= 0 || h <= 0 || w <= 0) return XPT_ERR_SHAPE;
  if (dtype != 0 && dtype != 1) return XPT_ERR_ARG;
  const long long total = M * 4 * h * w;
  if (total >= 0x7fffffffLL) return XPT_ERR_SHAPE;          // (32-bit index splits in the kernel)
  const XcdSweep sw = xpt_xcd_sweep(total, 4096);
  XPT_BEGIN_LAUNCH();
  if (dtype == 0)
    hipLaunchKernelGGL(upsample2x_fwd_kernel<float>, dim3(sw.grid), dim3(256), 0, (hipStream_t)stream, src,
                       (float*)out, total, h, w, sw);
  else
    hipLaunchKernelGGL(upsample2x_fwd_kernel<xpt_half_t>, dim3(sw.grid), dim3(256), 0, (hipStream_t)stream,
                       src, (xpt_half_t*)out, total, h, w, sw);
  return xpt_launch_status();
}

int xpt_upsample2x_bwd_add(const void* g, long long g_pitch, const float* addend, float* dsrc, long long M, int h, int w, int dtype,
                           void* stream) {
  XPT_CHECK_PTR(g); XPT_CHECK_PTR(dsrc);
  if (M <= 0 || h <= 0 || w <= 0 || g_pitch < 1) return XPT_ERR_SHAPE;
  if (dtype != 0 && dtype != 1) return XPT_ERR_ARG;
  const long long total = M * h * w;
  if (total >= 0x7fffffffLL) return XPT_ERR_SHAPE;
  const XcdSweep sw = xpt_xcd_sweep(total, 4096);
  XPT_BEGIN_LAUNCH();
  if (dtype == 0)
    hipLaunchKernelGGL(upsample2x_bwd_kernel<float>, dim3(sw.grid), dim3(256), 0, (hipStream_t)stream,
                       (const float*)g, g_pitch, addend, dsrc, total, h, w, sw);
  else
    hipLaunchKernelGGL(upsample2x_bwd_kernel<xpt_half_t>, dim3(sw.grid), dim3(256), 0, (hipStream_t)stream,
                       (const xpt_half_t*)g, g_pitch, addend, dsrc, total, h, w, sw);
  return xpt_launch_status();
}

int xpt_upsample2x_bwd(const void* g, long long g_pitch, float* dsrc, long long M, int h, int w, int dtype, void* stream) {
  return xpt_upsample2x_bwd_add(g, g_pitch, nullptr, dsrc, M, h, w, dtype, stream);
}

/* The same for nscales (1..4) prediction maps in one launch; a scale whose g_depth[s] and g_disp[s] are both NULL gets gx = 0. */
int xpt_depth_head_ms_fwd(int nscales, const float* const* x, float* const* depth, float* const* disp, const long long* n,
                          void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(depth); XPT_CHECK_PTR(disp); XPT_CHECK_PTR(n);
  if (nscales < 1 || nscales > 4) return XPT_ERR_SHAPE;
  DepthHeadMs a = {};
  long long most = 0;
  for (int s = 0; s < nscales; ++s) {
    if (!x[s] || !depth[s] || !disp[s]) return XPT_ERR_NULL;
    if (n[s] <= 0) return XPT_ERR_SHAPE;
    a.x[s] = x[s]; a.depth[s] = depth[s]; a.disp[s] = disp[s]; a.n[s] = n[s];
    if (n[s] > most) most = n[s];
  }
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(depth_head_ms_fwd_kernel, dim3(grid_for(most), nscales), dim3(256), 0, (hipStream_t)stream, a);
  return xpt_launch_status();
}

int xpt_depth_head_ms_bwd(int nscales, const float* const* x, const float* const* g_depth, const float* const* g_disp,
                          float* const* gx, const long long* n, void* stream) {
  XPT_CHECK_PTR(x); XPT_CHECK_PTR(g_depth); XPT_CHECK_PTR(g_disp); XPT_CHECK_PTR(gx); XPT_CHECK_PTR(n);
  if (nscales < 1 || nscales > 4) return XPT_ERR_SHAPE;
  DepthHeadMs a = {};
  long long most = 0;
  for (int s = 0; s < nscales; ++s) {
    if (!x[s] || !gx[s]) return XPT_ERR_NULL;
    if (n[s] <= 0) return XPT_ERR_SHAPE;
    a.x[s] = x[s]; a.depth[s] = gx[s]; a.g_depth[s] = g_depth[s]; a.g_disp[s] = g_disp[s]; a.n[s] = n[s];
    if (n[s] > most) most = n[s];
  }
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(depth_head_ms_bwd_kernel, dim3(grid_for(most), nscales), dim3(256), 0, (hipStream_t)stream, a);
  return xpt_launch_status();
}

/* 3x3 / stride 1 / SAME average pooling (divisor excludes the padding) of an NHWC tensor, times `scale`; adjoint = 1
 * applies the transposed operator (the backward).  in: row pitch in_pitch >= C elements; out dense. */
int xpt_avgpool3_same(const void* in, long long in_pitch, void* out, int B, int H, int W, int C, float scale, int adjoint,
                      int dtype, void* stream) {
  XPT_CHECK_PTR(in); XPT_CHECK_PTR(out);
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || in_pitch < C) return XPT_ERR_SHAPE;
  if ((dtype != 0 && dtype != 1) || (adjoint != 0 && adjoint != 1)) return XPT_ERR_ARG;
  const int esz = dtype == 0 ? 4 : 2;
  int v = dtype == 0 ? 4 : 8;
  while (v > 1 && !(C % v == 0 && in_pitch % v == 0 && ((uintptr_t)in) % (size_t)(v * esz) == 0 &&
                    ((uintptr_t)out) % (size_t)(v * esz) == 0))
    v >>= 1;
  const long long total = (long long)B * H * W * (C / v);
  if (total >= (1LL << 31)) return XPT_ERR_SHAPE;        // (the kernel splits flat indices in 32 bits)
  const dim3 grid(grid_for(total));
  hipStream_t s = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
#define XPT_POOL(T, V) \
  hipLaunchKernelGGL((avgpool3_kernel<T, V>), grid, dim3(256), 0, s, (const T*)in, in_pitch, (T*)out, B, H, W, C, scale, adjoint)
  if (dtype == 0) {
    if (v == 4) XPT_POOL(float, 4);
    else if (v == 2) XPT_POOL(float, 2);
    else XPT_POOL(float, 1);
  } else {
    if (v == 8) XPT_POOL(xpt_half_t, 8);
    else if (v == 4) XPT_POOL(xpt_half_t, 4);
    else if (v == 2) XPT_POOL(xpt_half_t, 2);
    else XPT_POOL(xpt_half_t, 1);
  }
#undef XPT_POOL
  return xpt_launch_status();
}

/* out [rows, C] (dense) = sum of n (2..8) inputs [rows, C] with row pitches >= C; fp32 accumulation in input order. */
int xpt_sum_rows(const void* const* inputs, const long long* pitches, int n, void* out, long long rows, int C, int dtype,
                 void* stream) {
  XPT_CHECK_PTR(inputs); XPT_CHECK_PTR(pitches); XPT_CHECK_PTR(out);
  if (n < 2 || n > 8) return XPT_ERR_ARG;
  if (rows <= 0 || C <= 0) return XPT_ERR_SHAPE;
  if (dtype != 0 && dtype != 1) return XPT_ERR_ARG;
  SumInputs in;
  in.n = n;
  const int esz = dtype == 0 ? 4 : 2;
  int v = dtype == 0 ? 4 : 8;
  for (int i = 0; i < n; ++i) {
    if (inputs[i] == nullptr) return XPT_ERR_NULL;
    if (pitches[i] < C) return XPT_ERR_SHAPE;
    in.ptr[i] = inputs[i];
    in.pitch[i] = pitches[i];
  }
  auto ok = [&](const void* p, long long pitch) { return ((uintptr_t)p) % (size_t)(v * esz) == 0 && pitch % v == 0; };
  for (;;) {
    bool all = C % v == 0 && ok(out, C);
    for (int i = 0; i < n && all; ++i) all = ok(inputs[i], pitches[i]);
    if (all || v == 1) break;
    v >>= 1;
  }
  const long long total = rows * (C / v);
  if (total >= (1LL << 31)) return XPT_ERR_SHAPE;        // (the kernel splits flat indices in 32 bits)
  const XcdSweep sw = xpt_xcd_sweep(total, 4096);
  const dim3 grid(sw.grid);
  hipStream_t s = (hipStream_t)stream;
  XPT_BEGIN_LAUNCH();
#define XPT_SUM(T, V) hipLaunchKernelGGL((sum_rows_kernel<T, V>), grid, dim3(256), 0, s, in, (T*)out, rows, C, sw)
  if (dtype == 0) {
    if (v == 4) XPT_SUM(float, 4);
    else if (v == 2) XPT_SUM(float, 2);
    else XPT_SUM(float, 1);
  } else {
    if (v == 8) XPT_SUM(xpt_half_t, 8);
    else if (v == 4) XPT_SUM(xpt_half_t, 4);
    else if (v == 2) XPT_SUM(xpt_half_t, 2);
    else XPT_SUM(xpt_half_t, 1);
  }
#undef XPT_SUM
  return xpt_launch_status();
}

/* dst[i][0 .. bytes[i]) = src[i][0 .. bytes[i]) for n (1..8) device buffers, one launch (no overlap between any src and dst) */
int xpt_multi_copy(const void* const* src, void* const* dst, const long long* bytes, int n, void* stream) {
  XPT_CHECK_PTR(src); XPT_CHECK_PTR(dst); XPT_CHECK_PTR(bytes);
  if (n < 1 || n > 8) return XPT_ERR_ARG;
  CopyJobs jobs{};
  jobs.n = n;
  long long vecs = 0;
  for (int i = 0; i < n; ++i) {
    if (src[i] == nullptr || dst[i] == nullptr) return XPT_ERR_NULL;
    if (bytes[i] <= 0) return XPT_ERR_SHAPE;
    jobs.src[i] = (const unsigned char*)src[i];
    jobs.dst[i] = (unsigned char*)dst[i];
    jobs.bytes[i] = bytes[i];
    jobs.first[i] = vecs;
    vecs += (bytes[i] + 15) / 16;
  }
  for (int i = n; i <= 8; ++i) jobs.first[i] = vecs;
  long long blocks = (vecs + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, jobs);
  return xpt_launch_status();
}

/* out [rows, Ct] bf16 (dense, Ct % 8 == 0, 16-byte aligned) = the n (1..4) inputs [rows, channels[i]] (row pitches in
 * elements) side by side, zeros in the remaining channels. */
int xpt_concat_channels(const void* const* inputs, const long long* pitches, const int* channels, int n, void* out,
                        long long rows, int Ct, void* stream) {
  XPT_CHECK_PTR(inputs); XPT_CHECK_PTR(pitches); XPT_CHECK_PTR(channels); XPT_CHECK_PTR(out);
  if (n < 1 || n > 4) return XPT_ERR_ARG;
  if (rows <= 0 || Ct <= 0 || Ct % 8 != 0 || ((uintptr_t)out) % 16 != 0) return XPT_ERR_SHAPE;
  CatInputs in{};
  in.n = n;
  int c = 0;
  for (int i = 0; i < n; ++i) {
    if (inputs[i] == nullptr) return XPT_ERR_NULL;
    if (channels[i] <= 0 || pitches[i] < channels[i]) return XPT_ERR_SHAPE;
    in.ptr[i] = (const unsigned short*)inputs[i];
    in.pitch[i] = pitches[i];
    in.first[i] = c;
    c += channels[i];
  }
  for (int i = n; i <= 4; ++i) in.first[i] = c;
  if (c > Ct) return XPT_ERR_SHAPE;
  const long long total = rows * (Ct / 8);
  if (total >= (1LL << 31)) return XPT_ERR_SHAPE;
  XPT_BEGIN_LAUNCH();
  const XcdSweep sw = xpt_xcd_sweep(total, 4096);
  hipLaunchKernelGGL(concat_channels_kernel, dim3(sw.grid), dim3(256), 0, (hipStream_t)stream, in,
                     (unsigned short*)out, (unsigned)total, Ct / 8, sw);
  return xpt_launch_status();
}

}  // extern "C"
