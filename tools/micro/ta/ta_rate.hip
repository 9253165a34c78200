// Dev probe: vector-memory instruction rate per CU on gfx950 for the access shapes k_spconv3 issues (all 16 B per lane):
//   0 contiguous 1 KB | 1 32 rows x 32 B (A-fragment gather) | 2 same, 62 % of the rows out of range | 3 all out of range
//   4 4 whole 256-B rows per instruction | 5 = 0 into LDS (buffer_load .. lds) | 6 = 4 into LDS | 7 = 4 with 62 % rows out of range into LDS
// rows come from a window of `win` bytes (L2-resident when small).  Build: hipcc --offload-arch=gfx950 -O3 ta_rate.hip -o ta_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define OOB 0x80000000u
template <int MODE>
__global__ __launch_bounds__(256, 3) void k(const unsigned char* x, unsigned nbytes, unsigned win, int iters, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[4 * 8 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
  unsigned seed = (blockIdx.x * 4 + wave) * 2654435761u + 12345u;
  const unsigned base = (unsigned)((blockIdx.x % 8) * (size_t)win);     // each XCD its own window
  u32x4 acc = {0, 0, 0, 0};
  const unsigned rowsel = (MODE == 4 || MODE >= 6) ? (unsigned)(lane >> 4) : (unsigned)(lane & 31);
  const unsigned inrow = (MODE == 4 || MODE >= 6) ? (unsigned)(lane & 15) * 16u : (unsigned)(lane >> 5) * 16u;
  for (int it = 0; it < iters; ++it) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      seed = seed * 1664525u + 1013904223u;
      unsigned off;
      if (MODE == 0 || MODE == 5) off = base + ((seed >> 8) % (win / 1024)) * 1024u + lane * 16u;
      else {
        // per-row pseudo-random row index: hash(seed, rowsel)
        unsigned h = (seed ^ (rowsel * 0x9E3779B9u)) * 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
        const unsigned row = h % (win / 256);
        off = base + row * 256u + inrow + (MODE == 1 || MODE == 2 || MODE == 3 ? ((seed >> 28) & 3) * 32u : 0u);
        if ((MODE == 2 || MODE == 7) && (h >> 20) % 100 < 62) off = OOB;
        if (MODE == 3) off = OOB;
      }
      if (MODE >= 5) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(sm + (wave * 8 + u) * 1024), 16, off, 0, 0, 0);
      } else {
        v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
      }
    }
    if (MODE >= 5) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
      acc[0] += *reinterpret_cast<unsigned*>(sm + (wave * 8) * 1024 + lane * 4);
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[0] += v[u][0] ^ v[u][3];
    }
  }
  if (acc[0] == 0x12345u) sink[0] = acc[0];
}
template <int MODE>
static void run(const unsigned char* x, unsigned nbytes, unsigned win, unsigned* sink, const char* name) {
  const int iters = 256, blocks = 768;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(x, nbytes, win, 8, sink);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(x, nbytes, win, iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = (double)blocks * 4 * iters * 8 / 256.0;
  printf("%-46s win %5.1f MB: %8.1f us  %.1f ns per instruction and CU (~%.0f cycles at 2.4 GHz)  %.2f TB/s nominal\n", name, win / 1e6, ms * 1e3,
         ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4, (double)blocks * 4 * iters * 8 * 1024 / ms / 1e9);
}
int main() {
  const unsigned nbytes = 512u << 20;
  unsigned char* x; unsigned* sink;
  hipMalloc(&x, nbytes); hipMemset(x, 1, nbytes); hipMalloc(&sink, 64);
  for (unsigned win : {1u << 20, 3u << 20, 24u << 20}) {
    run<0>(x, nbytes, win, sink, "0 contiguous 1 KB");
    run<1>(x, nbytes, win, sink, "1 32 rows x 32 B");
    run<2>(x, nbytes, win, sink, "2 32 rows x 32 B, 62 % out of range");
    run<3>(x, nbytes, win, sink, "3 all out of range");
    run<4>(x, nbytes, win, sink, "4 4 rows x 256 B");
    run<5>(x, nbytes, win, sink, "5 contiguous 1 KB -> LDS");
    run<6>(x, nbytes, win, sink, "6 4 rows x 256 B -> LDS");
    run<7>(x, nbytes, win, sink, "7 4 rows x 256 B -> LDS, 62 % out of range");
  }
  return 0;
}
