#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short s[64 * 4];
  // lane i's own 8-byte chunk holds (i*4 + e)
  for (int e = 0; e < 4; ++e) s[threadIdx.x * 4 + e] = (short)(threadIdx.x * 4 + e);
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s + threadIdx.x * 4));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  k<<<1, 64>>>(d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" (l%d,e%d)", h[l*4+e] / 4, h[l*4+e] % 4); printf("\n"); }
  return 0;
}
