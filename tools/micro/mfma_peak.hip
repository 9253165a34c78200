// Dev microbenchmark: sustained v_mfma_f32_16x16x4_f32 rate vs independent accumulators and waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs_per_cu, int threads) {
  float* out; hipMalloc(&out, 256 * 8 * 1024 * 4);
  int iters = 4000;
  int grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<grid, threads>>>(out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<grid, threads>>>(out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * (threads / 64) * iters * 8.0 * NACC * 2048.0;
  printf("nacc %d wg/cu %d threads %d : %.2f ms  %.1f TF\n", NACC, wgs_per_cu, threads, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<1>(1, 256); run<2>(1, 256); run<4>(1, 256); run<2>(2, 256); run<4>(2, 256); run<1>(2, 256); run<1>(4, 256); run<2>(4, 256);
  return 0;
}
