#include <hip/hip_runtime.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* x, unsigned nbytes, const unsigned* offs, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned sm[64 * 4 * 2];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) sm[i] = 0xdeadbeefu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
  unsigned off = offs[lane];
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)sm, 16, off, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(sm + 256), 16, off, 0, 16, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  u32x4 v = *reinterpret_cast<u32x4*>(&sm[lane * 4]);
  u32x4 w = *reinterpret_cast<u32x4*>(&sm[256 + lane * 4]);
  for (int j = 0; j < 4; ++j) { out[lane * 8 + j] = v[j]; out[lane * 8 + 4 + j] = w[j]; }
}
int main() {
  unsigned *x, *offs, *out;
  hipMalloc(&x, 4096); hipMalloc(&offs, 256); hipMalloc(&out, 64 * 32);
  unsigned hx[1024]; for (int i = 0; i < 1024; ++i) hx[i] = i;
  unsigned ho[64]; for (int i = 0; i < 64; ++i) ho[i] = (i % 3 == 0) ? 0x80000000u : (unsigned)((i * 37) % 100) * 32u;
  hipMemcpy(x, hx, 4096, hipMemcpyHostToDevice); hipMemcpy(offs, ho, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(x, 4096, offs, out);
  unsigned r[512]; hipMemcpy(r, out, 2048, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) for (int j = 0; j < 8; ++j) {
    unsigned e = (ho[i] & 0x80000000u) ? 0u : ho[i] / 4 + j;
    if (r[i * 8 + j] != e) { if (bad < 8) printf("lane %d j %d got %08x want %08x\n", i, j, r[i * 8 + j], e); ++bad; }
  }
  printf("glds oob test: %d mismatches\n", bad);
  return bad != 0;
}
