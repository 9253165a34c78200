// Dev probe: are 16-byte global loads at 4-byte alignment legal / how fast on gfx950? (rows of 135 floats)
// hipcc --offload-arch=gfx950 -O3 tools/micro/unaligned_load.hip -o /tmp/unaligned && /tmp/unaligned
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k_rows(const float* __restrict__ x, int n, int ld, float* __restrict__ out) {
  // 32 lanes per row of 128 floats (float4 each), rows of stride ld floats
  const int r = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  if (r >= n) return;
  const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * ld + 4 * l);
  out[(size_t)r * 32 + l] = v.x + v.y + v.z + v.w;
}
int main() {
  const int n = 1 << 20;
  for (int ld : {128, 136, 135}) {
    std::vector<float> h((size_t)n * ld);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 1000) * 0.001f;
    float *x, *o;
    hipMalloc(&x, h.size() * 4); hipMalloc(&o, (size_t)n * 32 * 4);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 3; ++it) k_rows<<<n / 8, 256>>>(x, n, ld, o);
    hipEventRecord(a);
    for (int it = 0; it < 10; ++it) k_rows<<<n / 8, 256>>>(x, n, ld, o);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<float> ho((size_t)n * 32);
    hipError_t e = hipMemcpy(ho.data(), o, ho.size() * 4, hipMemcpyDeviceToHost);
    double err = 0;
    for (int r = 0; r < n; r += 9973) for (int l = 0; l < 32; ++l) {
      const float* p = &h[(size_t)r * ld + 4 * l];
      err = fmax(err, fabs((double)ho[(size_t)r * 32 + l] - (double)(p[0] + p[1] + p[2] + p[3])));
    }
    printf("ld %d: %s, %.1f us/launch, %.0f GB/s of row bytes, max err %g\n", ld, hipGetErrorString(e), ms * 100,
           (double)n * 512 / (ms * 1e-4) / 1e9, err);
    hipFree(x); hipFree(o);
  }
  return 0;
}
