// Dev probe: do a wave's MFMAs overlap with ANOTHER resident wave's LDS / vector-memory / VALU work on the same SIMD?
// 512 workgroups of 256 threads (2 per CU; each SIMD hosts one wave of each). Even workgroups run a fp32 MFMA chain
// (16 independent accumulators, like k_wgrad_pairs), odd ones run the "other" work; each is also timed alone.
// hipcc --offload-arch=gfx950 -O3 tools/micro/coissue.hip -o tools/micro/coissue && tools/micro/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma_work(int iters, float* out) {
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
// the same FLOPs with v_mfma_f32_32x32x2_f32 (4096 FLOP per instruction, 4 independent accumulators)
__device__ __forceinline__ void mfma_work32(int iters, float* out) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>   // 1 = LDS b128 reads, 2 = LDS b128 writes, 3 = global loads (L1-resident), 4 = VALU fma, 5 = barriers + LDS
__device__ __forceinline__ void other_work(int iters, const float4* __restrict__ g, float* out, float4* lds) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int t = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 1 || MODE == 5) {
        const float4 v = lds[(t + 64 * u + it) & 2047];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      } else if (MODE == 2) {
        lds[(t + 64 * u + it) & 2047] = acc;
        acc.x += 1.f;
      } else if (MODE == 3) {
        const float4 v = g[(t + 256 * u + 64 * (it & 7)) & 4095];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc.x = fmaf(acc.x, 1.0001f, acc.y), acc.y = fmaf(acc.y, 0.9999f, acc.z), acc.z = fmaf(acc.z, 1.0002f, acc.w), acc.w = fmaf(acc.w, 0.9998f, acc.x);
      }
    }
    if (MODE == 5) __syncthreads();
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[1] = acc.x;
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_mix(int which, int it_m, int it_o, const float4* __restrict__ g, float* out) {
  __shared__ float4 lds[2048 + 2048];          // 64 KB: two workgroups per CU
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  const bool is_m = (blockIdx.x & 1) == 0;
  if (which == 0 && !is_m) return;             // MFMA workgroups only
  if (which == 1 && is_m) return;              // other workgroups only
  if (MODE == 7 || MODE == 8) {                  // 32x32x2 chains: 7 = even workgroups only / both, 8 = beside LDS reads
    if (is_m || MODE == 7) mfma_work32(it_m, out);
    else other_work<1>(it_o, g, out, lds);
    return;
  }
  if (is_m || MODE == 6) mfma_work(it_m, out);   // MODE 6: the odd workgroups run the MFMA chain too (2 MFMA waves per SIMD)
  else other_work<MODE>(it_o, g, out, lds);
}

template <int MODE>
float run(int which, int it_m, int it_o, const float4* g, float* out) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  k_mix<MODE><<<512, 256>>>(which, it_m, it_o, g, out);
  (void)hipEventRecord(a);
  for (int r = 0; r < 5; ++r) k_mix<MODE><<<512, 256>>>(which, it_m, it_o, g, out);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms * 200.f;                            // us per launch
}

template <int MODE>
void test(const char* name, int it_o, const float4* g, float* out) {
  const int it_m = 2000;
  const float tm = run<MODE>(0, it_m, it_o, g, out), to = run<MODE>(1, it_m, it_o, g, out), tb = run<MODE>(2, it_m, it_o, g, out);
  printf("%-34s MFMA alone %7.1f us | other alone %7.1f us | together %7.1f us  (max %7.1f, sum %7.1f)\n", name, tm, to, tb,
         tm > to ? tm : to, tm + to);
}

int main() {
  float4* g; float* out;
  (void)hipMalloc(&g, 4096 * sizeof(float4)); (void)hipMalloc(&out, 64);
  (void)hipMemset(g, 0, 4096 * sizeof(float4));
  test<1>("LDS ds_read_b128", 6000, g, out);
  test<2>("LDS ds_write_b128", 6000, g, out);
  test<3>("global_load_dwordx4 (L1 hits)", 2500, g, out);
  test<4>("VALU fma", 1500, g, out);
  test<5>("LDS reads + barrier per 8", 3000, g, out);
  test<6>("MFMA chain in BOTH workgroups", 0, g, out);
  test<7>("32x32x2 chain in BOTH workgroups", 0, g, out);
  test<8>("32x32x2 chain | LDS ds_read_b128", 6000, g, out);
  return 0;
}
