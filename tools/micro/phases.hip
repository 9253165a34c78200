// Dev probe: workgroups that alternate an MFMA chain (fp32 16x16x4, ~what an item of k_spconv2<128,128> issues) and a non-MFMA
// phase (LDS traffic + two barriers), with 1, 2, 3 or 4 workgroups resident per CU (dynamic LDS sets the residency).
// hipcc --offload-arch=gfx950 -O3 tools/micro/phases.hip -o tools/micro/phases && tools/micro/phases
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(int items, int mfma_per_item, int lds_per_item, float* out) {
  extern __shared__ float4 lds[];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  const float a = threadIdx.x * 1e-3f, b = 1.f;
  for (int it = 0; it < items; ++it) {
    for (int m = 0; m < mfma_per_item; m += 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    __syncthreads();
    for (int l = 0; l < lds_per_item; ++l) {
      const float4 v = lds[(threadIdx.x + 64 * l + it) & 2047];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      lds[(threadIdx.x * 7 + l) & 2047] = s;
    }
    __syncthreads();
  }
  float r = s.x + s.y + s.z + s.w;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += acc[i][0];
  if (r == 12345.678f) out[0] = r;
}
int main() {
  float* out; (void)hipMalloc(&out, 64);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int items = 400, mfma = 104;             // 104 MFMAs x 32 cycles = 3300 full-rate cycles per item
  for (int ldsn : {0, 12, 24}) {
    for (int res : {1, 2, 3, 4}) {
      const size_t lds = (160 * 1024) / res - 1024;
      const int blocks = 256 * res;
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      k<<<blocks, 256, lds>>>(items, mfma, ldsn, out);
      (void)hipEventRecord(e0);
      for (int r = 0; r < 3; ++r) k<<<blocks, 256, lds>>>(items, mfma, ldsn, out);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / 3;
      const double tf = 2048.0 * mfma * items * 4 * blocks / us / 1e6;
      printf("non-MFMA phase %2d LDS round trips | %d workgroups per CU: %8.1f us, %6.1f TFLOP/s (%.0f %% of 157), %.0f cycles per item and CU at 2.4 GHz\n",
             ldsn, res, us, tf, 100 * tf / 157.3, us * 2400.0 / (items * res));
    }
  }
  return 0;
}
