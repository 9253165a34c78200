// Dev probe: how many bytes per clock can one CU pull from the L2 when every load hits there? (the weight slices of
// k_spconv2 are such a stream).  Each wave reads 1 KiB pieces (16 B per lane) of an L2-resident buffer, NL independent
// loads in flight per wave, 8 waves per CU, every CU busy.  The buffer (1.7 MB = 27 fp32 128x128 slices) is larger than the
// 32 KB L1, and waves walk it at different phases, so nothing is served by the L1.
// hipcc --offload-arch=gfx950 -O3 tools/micro/l2_stream.hip -o /tmp/l2_stream && /tmp/l2_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NL>
__global__ __launch_bounds__(256, 2) void k_stream(const uint4* __restrict__ buf, int n_kib, int iters, unsigned* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned acc = 0;
  // piece index advances by NL per iteration; start at a wave- and block-dependent phase
  unsigned p = (blockIdx.x * 37u + wave * 11u) % (unsigned)n_kib;
  for (int it = 0; it < iters; ++it) {
    uint4 v[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      unsigned q = p + i * 4;                         // the 4 waves of a block interleave: consecutive KiB of a 16 KiB slice
      q %= (unsigned)n_kib;
      v[i] = buf[(size_t)q * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    p += NL * 4;
    p %= (unsigned)n_kib;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <int NL>
void run(const uint4* buf, int n_kib, unsigned* out, int blocks) {
  const int iters = 4096 / NL;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k_stream<NL><<<blocks, 256>>>(buf, n_kib, iters, out);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) k_stream<NL><<<blocks, 256>>>(buf, n_kib, iters, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = 5.0 * blocks * 4 * (double)iters * NL * 1024;
  const double gbs = bytes / (ms * 1e-3) / 1e9;
  printf("blocks %4d  loads in flight per wave %2d : %7.0f GB/s total, %6.1f GB/s per CU (%.1f B/clk at 2.4 GHz)\n", blocks, NL, gbs,
         gbs / 256, gbs / 256 / 2.4);
}
int main() {
  const int n_kib = 27 * 64;                          // 1.7 MB
  uint4* buf; unsigned* out;
  hipMalloc(&buf, (size_t)n_kib * 1024); hipMalloc(&out, 64);
  hipMemset(buf, 1, (size_t)n_kib * 1024);
  for (int blocks : {512, 256}) {
    run<4>(buf, n_kib, out, blocks);
    run<8>(buf, n_kib, out, blocks);
    run<16>(buf, n_kib, out, blocks);
    run<24>(buf, n_kib, out, blocks);
  }
  // a buffer that fits the L1 (16 KiB) for comparison
  printf("L1-resident (16 KiB):\n");
  run<16>(buf, 16, out, 512);
  return 0;
}
