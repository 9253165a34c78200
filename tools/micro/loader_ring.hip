// Dev probe (round 3): can dedicated LOADER waves keep four fp32-MFMA waves fed with W[k] chunks through an LDS ring?
//   workgroup = 4 consumer waves (each owns 32 of 128 output columns) + NL loader waves
//   chunk     = 16 reduction channels x 128 columns fp32 = 8 KiB = 8 x 1 KiB global_load_lds_dwordx4
//   loader    : wait slot free (done[] counter) -> issue its share of the chunk -> counted vmcnt -> ready[] += 1
//   consumer  : wait ready[] -> 2 x ds_read_b128 (its B fragments) -> done[] += 1 -> G groups x 8 MFMAs (A from an LDS tile)
// Weight source: 27 x 64 KiB (L2 resident), every workgroup streams all of it, like k_spconv2<128,128>.
// hipcc --offload-arch=gfx950 -O3 tools/micro/loader_ring.hip -o tools/micro/loader_ring && tools/micro/loader_ring
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define SPIN_MAX (1 << 22)

__device__ __forceinline__ unsigned lds_off(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

__device__ __forceinline__ void lds_inc(unsigned* p) {       // ds_add_u32 without a return value
  asm volatile("ds_add_u32 %0, %1" :: "v"(lds_off(p)), "v"(1u) : "memory");
}

template <int R, int NL, int G, int STAGE>
__global__ __launch_bounds__(256 + 64 * NL) void k(const float* __restrict__ w, int K, int nchunks, float* out,
                                                   unsigned* err) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ring = lds;                       // R x 2048 floats
  float* sA = lds + R * 2048;              // 32 rows x 132 floats (A tile stand-in)
  __shared__ unsigned ready[16], done[16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 32 * 132; i += blockDim.x) sA[i] = (float)(i & 7) * 0.125f;
  if (tid < 16) { ready[tid] = 0; done[tid] = 0; }
  __syncthreads();
  constexpr int PER = 8 / NL;              // DMA instructions per loader per chunk
  constexpr int D = (STAGE == 1) ? (R > 4 ? 4 : R) : R;   // chunks in flight per loader
  if (wave >= 4 && STAGE == 1) {
    // register-staged loader: D chunk-shares in flight in VGPRs (PER float4 each), written to the ring with ds_write_b128
    const int l = wave - 4;
    volatile unsigned* vdone = done;
    float4 st[D][PER];
    auto issue = [&](int c, auto I_) {
      constexpr int I = decltype(I_)::value;
      const int k = (c >> 3) % K, j = c & 7;
      const float4* src = reinterpret_cast<const float4*>(w + ((size_t)(k * 8 + j)) * 2048 + (l * PER) * 256) + lane;
#pragma unroll
      for (int i = 0; i < PER; ++i) st[I][i] = src[i * 64];
    };
    auto land = [&](int c, auto I_) {
      constexpr int I = decltype(I_)::value;
      const int slot = c % R, gen = c / R;
      int spins = 0;
      while (vdone[slot] < 4u * gen) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_MAX) { if (lane == 0) atomicAdd(err, 1u); break; }
      }
      float4* dst = reinterpret_cast<float4*>(ring + slot * 2048 + (l * PER) * 256) + lane;
#pragma unroll
      for (int i = 0; i < PER; ++i) dst[i * 64] = st[I][i];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) lds_inc(&ready[slot]);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    static_assert(D <= 4, "register sets");
    // prologue: D - 1 chunks in flight
    if (D > 1 && 0 < nchunks) issue(0, I0{});
    if (D > 2 && 1 < nchunks) issue(1, I1{});
    if (D > 3 && 2 < nchunks) issue(2, I2{});
    for (int c = 0; c < nchunks; c += D) {
      // unrolled by D so that every register set is a compile-time name
      if (D == 1) { issue(c, I0{}); land(c, I0{}); continue; }
      if (D == 2) {
        if (c + 1 < nchunks) issue(c + 1, I1{}); land(c, I0{});
        if (c + 1 < nchunks) { if (c + 2 < nchunks) issue(c + 2, I0{}); land(c + 1, I1{}); }
      } else if (D == 3) {
        if (c + 2 < nchunks) issue(c + 2, I2{}); land(c, I0{});
        if (c + 1 < nchunks) { if (c + 3 < nchunks) issue(c + 3, I0{}); land(c + 1, I1{}); }
        if (c + 2 < nchunks) { if (c + 4 < nchunks) issue(c + 4, I1{}); land(c + 2, I2{}); }
      } else {
        if (c + 3 < nchunks) issue(c + 3, I3{}); land(c, I0{});
        if (c + 1 < nchunks) { if (c + 4 < nchunks) issue(c + 4, I0{}); land(c + 1, I1{}); }
        if (c + 2 < nchunks) { if (c + 5 < nchunks) issue(c + 5, I1{}); land(c + 2, I2{}); }
        if (c + 3 < nchunks) { if (c + 6 < nchunks) issue(c + 6, I2{}); land(c + 3, I3{}); }
      }
    }
    return;
  }
  if (wave >= 4) {
    const int l = wave - 4;
    volatile unsigned* vdone = done;
    for (int c = 0; c < nchunks + D - 1; ++c) {
      if (c < nchunks) {
        const int slot = c % R, gen = c / R;
        int spins = 0;
        while (vdone[slot] < 4u * gen) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > SPIN_MAX) { if (lane == 0) atomicAdd(err, 1u); break; }
        }
        const int k = (c >> 3) % K, j = c & 7;
        const float* src = w + ((size_t)(k * 8 + j)) * 2048 + (l * PER) * 256 + lane * 4;
        const unsigned dst = lds_off(ring + slot * 2048 + (l * PER) * 256);
#pragma unroll
        for (int i = 0; i < PER; ++i) glds16(src + i * 256, dst + i * 1024);
      }
      if (c >= D - 1) {                    // chunk c - (D - 1) has landed once at most (D - 1) * PER loads are outstanding
        if (c < nchunks) wait_vmcnt<(D - 1) * PER>(); else wait_vmcnt<0>();
        if (lane == 0) lds_inc(&ready[(c - (D - 1)) % R]);
      }
    }
    return;
  }
  // ---- consumer: B fragments of chunk c + 1 and the flag of chunk c + 2 are requested before the MFMAs of chunk c ----
  const int m = lane & 15, g4 = lane >> 4;
  f32x4 acc[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[g][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  volatile unsigned* vready = ready;
  auto wait_ready = [&](int c) {
    const int slot = c % R, gen = c / R;
    int spins = 0;
    while (vready[slot] < (unsigned)NL * (gen + 1)) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_MAX) { if (lane == 0) atomicAdd(err, 1u); break; }
    }
  };
  auto bptr = [&](int c, int t) { return reinterpret_cast<const float4*>(ring + (c % R) * 2048 + wave * 512 + t * 256 + lane * 4); };
  wait_ready(0);
  float4 b0 = *bptr(0, 0), b1 = *bptr(0, 1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) lds_inc(&done[0]);
  unsigned flag_next = (nchunks > 1) ? vready[1 % R] : 0u;       // flag of chunk 1
  for (int c = 0; c < nchunks; ++c) {
    float4 n0 = b0, n1 = b1;
    unsigned flag_after = 0u;
    const bool more = c + 1 < nchunks;
    if (more) {
      if (flag_next < (unsigned)NL * ((c + 1) / R + 1)) wait_ready(c + 1);      // rare: the loader is behind
      n0 = *bptr(c + 1, 0);
      n1 = *bptr(c + 1, 1);
      if (c + 2 < nchunks) flag_after = vready[(c + 2) % R];
    }
    const int j = c & 7;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float4 a = *reinterpret_cast<const float4*>(sA + (16 * (g & 1) + m) * 132 + 16 * j + 4 * g4);
      acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc[g][0], 0, 0, 0);
      acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc[g][1], 0, 0, 0);
      acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc[g][0], 0, 0, 0);
      acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc[g][1], 0, 0, 0);
      acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, acc[g][0], 0, 0, 0);
      acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, acc[g][1], 0, 0, 0);
      acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, acc[g][0], 0, 0, 0);
      acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, acc[g][1], 0, 0, 0);
    }
    if (more) {
      asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(n0.x), "v"(n1.x) : "memory");
      if (lane == 0) lds_inc(&done[(c + 1) % R]);
      b0 = n0; b1 = n1;
      flag_next = flag_after;
    }
  }
  float r = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) r += acc[g][0][0] + acc[g][1][1];
  if (r == 12345.678f) out[0] = r;
}

template <int R, int NL, int G, int STAGE>
static void run(const float* w, float* out, unsigned* err, int res) {
  const int K = 27, items = 108, nchunks = items * 8;
  const size_t lds = (160 * 1024) / res - 1024;
  if (lds < (size_t)R * 8192 + 32 * 132 * 4) return;
  (void)hipFuncSetAttribute((const void*)k<R, NL, G, STAGE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int blocks = 256 * res * 2;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipMemset(err, 0, 4);
  k<R, NL, G, STAGE><<<blocks, 256 + 64 * NL, lds>>>(w, K, nchunks, out, err);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) k<R, NL, G, STAGE><<<blocks, 256 + 64 * NL, lds>>>(w, K, nchunks, out, err);
  (void)hipEventRecord(e1);
  if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); exit(1); }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned herr = 0; (void)hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
  const double us = ms * 1e3 / 3;
  const double tf = 2048.0 * 8 * G * nchunks * 4 * blocks / us / 1e6 + 1e-9;
  const double gbs = 8192.0 * nchunks * blocks / us / 1e3;
  printf("%s ring %d x 8 KiB, %d loader(s), %d WG/CU, %d group(s)/chunk: %8.1f us  %6.1f TFLOP/s (%3.0f %% of 157)  W stream %6.0f GB/s = %4.1f B/clk/CU  give-ups %u\n",
         STAGE ? "reg-staged" : "LDS-DMA   ", R, NL, res, G, us, tf, 100 * tf / 157.3, gbs, gbs / 256 / 2.4, herr);
}

int main() {
  float *w, *out; unsigned* err;
  const size_t wn = (size_t)27 * 128 * 128;
  (void)hipMalloc(&w, wn * 4); (void)hipMalloc(&out, 64); (void)hipMalloc(&err, 4);
  float* h = (float*)malloc(wn * 4);
  for (size_t i = 0; i < wn; ++i) h[i] = (float)((i * 2654435761u) >> 20 & 255) / 256.f - 0.5f;
  (void)hipMemcpy(w, h, wn * 4, hipMemcpyHostToDevice);
  for (int res : {1, 2, 3}) {
    run<2, 1, 0, 1>(w, out, err, res); run<3, 1, 0, 1>(w, out, err, res); run<4, 1, 0, 1>(w, out, err, res); run<4, 2, 0, 1>(w, out, err, res);
    run<3, 1, 1, 1>(w, out, err, res); run<3, 1, 2, 1>(w, out, err, res); run<3, 2, 2, 1>(w, out, err, res); run<4, 2, 2, 1>(w, out, err, res);
    run<3, 1, 3, 1>(w, out, err, res); run<3, 2, 3, 1>(w, out, err, res); run<3, 2, 4, 1>(w, out, err, res);
  }
  return 0;
}
