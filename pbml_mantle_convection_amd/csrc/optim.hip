// Fused multi-tensor Adam over flat f32 buffers (torch.optim.Adam semantics: L2 weight decay folded
// into the gradient, bias-corrected moments; reference multigpu.py:761-763).
#include "common.h"

namespace {

__global__ void k_step_inc(int32_t* step) { if (threadIdx.x == 0 && blockIdx.x == 0) *step += 1; }

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, int64_t numel, const float* __restrict__ lr_dev,
                                              float b1, float b2, float eps, float wd, float gscale,
                                              const int32_t* __restrict__ step_dev) {
  const float lr = *lr_dev;
  const int t = *step_dev;
  const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const int64_t n4 = numel / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gr = G[j] * gscale + wd * P[j];
      M[j] = b1 * M[j] + (1.f - b1) * gr;
      V[j] = b2 * V[j] + (1.f - b2) * gr * gr;
      P[j] -= step_size * M[j] / (sqrtf(V[j]) * inv_sqrt_bc2 + eps);
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < numel; i += stride) {
    float gr = g[i] * gscale + wd * p[i];
    float mi = b1 * m[i] + (1.f - b1) * gr, vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi; v[i] = vi;
    p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
  }
}

}  // namespace

extern "C" int mc_adam_step_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                                 const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                                 float grad_scale, int32_t* step_count_dev, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !lr_dev || !step_count_dev || numel <= 0) return MC_EINVAL;
  if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) return MC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(64), 0, s, step_count_dev);
  int64_t blocks = (numel / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, numel, lr_dev, beta1,
                     beta2, eps, weight_decay, grad_scale, step_count_dev);
  MC_CHECK_LAUNCH();
  return MC_OK;
}
