// Dev probe: which SIMD does each wave of a 256-thread workgroup land on (HW_ID register), 2 workgroups per CU?
// hipcc --offload-arch=gfx950 -O3 tools/micro/simd_map.hip -o tools/micro/simd_map && tools/micro/simd_map
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int spin) {
  __shared__ float pad[16000];                       // 64 KB: two workgroups per CU
  if (threadIdx.x == 0) pad[0] = 1.f;
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID, all 32 bits
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}                   // keep everything resident at the same time
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw;
  if (pad[0] == 2.f) out[0] = 0;
}
int main() {
  const int nb = 512;
  unsigned* d; (void)hipMalloc(&d, nb * 4 * sizeof(unsigned));
  k<<<nb, 256>>>(d, 200000);
  std::vector<unsigned> h(nb * 4);
  (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  int hist[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < nb; ++b) {
    unsigned mask = 0;
    for (int w = 0; w < 4; ++w) mask |= 1u << ((h[b * 4 + w] >> 4) & 3);
    hist[__builtin_popcount(mask)]++;
    if (b < 6) {
      printf("block %d:", b);
      for (int w = 0; w < 4; ++w) printf("  wave %d -> simd %u cu %u se %u (hw %08x)", w, (h[b * 4 + w] >> 4) & 3, (h[b * 4 + w] >> 8) & 15, (h[b * 4 + w] >> 13) & 7, h[b * 4 + w]);
      printf("\n");
    }
  }
  printf("workgroups whose 4 waves sit on 1 / 2 / 3 / 4 distinct SIMDs: %d / %d / %d / %d\n", hist[1], hist[2], hist[3], hist[4]);
  return 0;
}
