// irx_optim.hip — fused Adam over ONE flat fp32 parameter buffer (all parameters / gradients / moments are views
// of four flat arrays, instancerefer_amd/optim.py). Replaces torch.optim.Adam's ~10 multi-tensor launches and
// 160 per-parameter host-side scalar reads per step with a single HBM-bound launch: 16 B/lane, 4 streams in,
// 3 streams out = 28 B/element. Semantics = torch.optim.Adam(lr, betas, eps, weight_decay) (L2 added to the
// gradient; reference scripts/train.py:121: Adam(lr=1e-3, weight_decay=1e-5)).
#include "irx_common.h"

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                              float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                              float grad_scale) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  const float step_size = lr / bc1;
  for (; i + 3 < n; i += stride) {
    float4 pp = *reinterpret_cast<float4*>(p + i);
    const float4 gg = *reinterpret_cast<const float4*>(g + i);
    float4 mm = *reinterpret_cast<float4*>(m + i);
    float4 vv = *reinterpret_cast<float4*>(v + i);
    float* pa = reinterpret_cast<float*>(&pp);
    const float* ga = reinterpret_cast<const float*>(&gg);
    float* ma = reinterpret_cast<float*>(&mm);
    float* va = reinterpret_cast<float*>(&vv);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = ga[j] * grad_scale + wd * pa[j];
      ma[j] = b1 * ma[j] + (1.f - b1) * gr;
      va[j] = b2 * va[j] + (1.f - b2) * gr * gr;
      const float denom = sqrtf(va[j]) / bc2_sqrt + eps;
      pa[j] -= step_size * (ma[j] / denom);
    }
    *reinterpret_cast<float4*>(p + i) = pp;
    *reinterpret_cast<float4*>(m + i) = mm;
    *reinterpret_cast<float4*>(v + i) = vv;
  }
  // tail (n not a multiple of 4): handled by the first threads of block 0
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t t = (n & ~(size_t)3) + threadIdx.x;
    const float gr = g[t] * grad_scale + wd * p[t];
    m[t] = b1 * m[t] + (1.f - b1) * gr;
    v[t] = b2 * v[t] + (1.f - b2) * gr * gr;
    p[t] -= step_size * (m[t] / (sqrtf(v[t]) / bc2_sqrt + eps));
  }
}

extern "C" int irx_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                             float grad_scale, void* stream) {
  IRX_REQUIRE(step >= 1, "irx_adam_step: step must be >= 1");
  if (n == 0) return IRX_OK;
  IRX_REQUIRE(params && grads && exp_avg && exp_avg_sq, "irx_adam_step: null pointer");
  IRX_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
              "irx_adam_step: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  long long blocks = (long long)((n / 4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  k_adam<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                                                        weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
  IRX_CHECK_LAUNCH("irx_adam_step");
  return IRX_OK;
}
