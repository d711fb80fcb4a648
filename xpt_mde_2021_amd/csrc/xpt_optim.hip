// xpt_optim.hip -- fused Adam over the flat parameter / gradient buffers (row a14 of the hot path).
// Replaces tf.optimizers.Adam(lr) as used by optimizer_factory (model/model_util/optimizers.py:7-13) and
// optimizer.apply_gradients (model/train_val.py:86).  Keras semantics (NOT torch's):
//   lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   p -= lr_t * m / (sqrt(v) + eps)           with eps = 1e-7 applied to the UNcorrected sqrt(v).
// One pass over 4 streams (p, g, m, v): 28 B / parameter, float4 accesses, HBM-bound.

#include "xpt_common.h"

__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, const float* __restrict__ step_ptr, float lr,
                            float b1, float b2, float eps, float grad_scale, int zero_grad,
                            xpt_half_t* __restrict__ shadow) {
  const float t = step_ptr[0];
  const float lr_t = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    float* pa = &pp.x; float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = ga[k] * grad_scale;
      ma[k] = b1 * ma[k] + (1.f - b1) * gk;
      va[k] = b2 * va[k] + (1.f - b2) * gk * gk;
      pa[k] -= lr_t * ma[k] / (sqrtf(va[k]) + eps);
    }
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
    if (zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (shadow) {                      // bf16 copy of the updated weights for the bf16 convolutions / GEMMs
#pragma unroll
      for (int k = 0; k < 4; ++k) shadow[4 * i + k] = xpt_float2half(pa[k]);
    }
  }
  // tail (n not a multiple of 4)
  for (long long i = (n4 << 2) + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gk = g[i] * grad_scale;
    const float mk = b1 * m[i] + (1.f - b1) * gk;
    const float vk = b2 * v[i] + (1.f - b2) * gk * gk;
    m[i] = mk; v[i] = vk;
    p[i] -= lr_t * mk / (sqrtf(vk) + eps);
    if (zero_grad) g[i] = 0.f;
    if (shadow) shadow[i] = xpt_float2half(p[i]);
  }
}

extern "C" int xpt_adam_step(float* param, float* grad, float* m, float* v, long long n, const float* step,
                             float lr, float beta1, float beta2, float eps, float grad_scale, int zero_grad,
                             void* shadow_bf16, void* stream) {
  XPT_CHECK_PTR(param); XPT_CHECK_PTR(grad); XPT_CHECK_PTR(m); XPT_CHECK_PTR(v); XPT_CHECK_PTR(step);
  if (n <= 0) return XPT_ERR_SHAPE;
  if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return XPT_ERR_ARG;
  long long blocks = ((n >> 2) + 255) / 256;
  if (blocks > 2048) blocks = 2048;   // 256 CUs x 8 blocks, grid-stride the rest
  if (blocks < 1) blocks = 1;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n,
                     step, lr, beta1, beta2, eps, grad_scale, zero_grad, (xpt_half_t*)shadow_bf16);
  return xpt_launch_status();
}

// tf.optimizers.SGD(learning_rate) (momentum 0, optimizers.py:10-11): p -= lr * g; same buffers, same bf16 shadow refresh.
__global__ void sgd_kernel(float* __restrict__ p, float* __restrict__ g, long long n, float lr, float grad_scale, int zero_grad,
                           xpt_half_t* __restrict__ shadow) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride) {
    const float pk = p[i] - lr * (g[i] * grad_scale);
    p[i] = pk;
    if (zero_grad) g[i] = 0.f;
    if (shadow) shadow[i] = xpt_float2half(pk);
  }
}

extern "C" int xpt_sgd_step(float* param, float* grad, long long n, float lr, float grad_scale, int zero_grad, void* shadow_bf16,
                            void* stream) {
  XPT_CHECK_PTR(param); XPT_CHECK_PTR(grad);
  if (n <= 0) return XPT_ERR_SHAPE;
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  XPT_BEGIN_LAUNCH();
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, n, lr, grad_scale,
                     zero_grad, (xpt_half_t*)shadow_bf16);
  return xpt_launch_status();
}
