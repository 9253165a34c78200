// irx_stem.hip — the stem convolution (3^3, Cin = C0 <= 8 input features -> 32 channels; reference
// models/basic_blocks.py:63-65) is HBM/gather-bound (7 flop/B): a 28-byte input row per pair. MFMA tiles would be
// 78 % padding, so these are plain VALU kernels organised for coalescing:
//   forward : 8 threads per output row (4 channels each); the 27 x Cin x 32 weights live in LDS; per valid
//             neighbour one broadcast read of the input row and Cin LDS float4 reads.
//   wgrad   : dense im2col tile assembled in LDS per 64-row chunk, MFMA with the rows as the reduction dimension
//             (see k_stem_wgrad); per-workgroup partial sums are reduced by k_wgrad_reduce (deterministic).
#include <stdlib.h>
#include "irx_common.h"

#define ST_COUT 32

// Rows per thread: the kernel is bound by its LDS weight reads (one ds_read_b128 per (offset, channel) per wave costs 8
// LDS cycles whether or not the lanes read the same address), so every weight fragment read is reused for ST_R rows.
#define ST_R 4

// One input row (CIN <= 8 floats, rows only 4-byte aligned: ldx = 7 for xyz + rgb + height) as TWO 16-byte loads — elements
// [0, 4) and [CIN - 4, CIN) — instead of CIN dword loads (16-byte global loads at dword alignment run at full speed on gfx950,
// tools/micro/unaligned_load.hip; both loads stay inside the row). Round 4: measured NEUTRAL on the forward (124.9 vs 124.6 us on
// the 489 k-voxel level): the kernel executes every offset at which ANY of a wave's 32 row slots has a neighbour — nearly all
// 27 — so its 28 FMAs per (slot, offset) run 5x more often than there are pairs (27 N x 7 x 32 x 2 = 5.9 GFLOP = 38 us at the
// fp32 vector peak); what would move it is a dense im2col tile in LDS + fp32 MFMA as k_stem_wgrad does (same 38 us floor at
// full MFMA efficiency), or bf16 operands for the stem (a numerics change the emulation oracle would have to follow).
template <int CIN>
__device__ __forceinline__ void st_load_row(const float* __restrict__ xr, float (&v)[CIN]) {
  if constexpr (CIN >= 4) {
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const f4u lo = *reinterpret_cast<const f4u*>(xr);
    const f4u hi = *reinterpret_cast<const f4u*>(xr + (CIN - 4));
#pragma unroll
    for (int c = 0; c < CIN; ++c) v[c] = c < 4 ? lo[c] : hi[c - (CIN - 4)];
  } else {
#pragma unroll
    for (int c = 0; c < CIN; ++c) v[c] = xr[c];
  }
}

template <int CIN>
__global__ __launch_bounds__(256) void k_stem_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                  const int32_t* __restrict__ nbr, int ld, int n_out, int K,
                                                  float* __restrict__ y, int ldx, int y_bf) {
  __shared__ __attribute__((aligned(16))) float sW[27 * CIN * ST_COUT];
  for (int i = threadIdx.x; i < K * CIN * ST_COUT; i += 256) sW[i] = w[i];
  __syncthreads();
  const int row0 = blockIdx.x * (32 * ST_R) + (threadIdx.x >> 3);     // rows row0 + 32 j
  const int c4 = (threadIdx.x & 7) * 4;
  float4 acc[ST_R];
#pragma unroll
  for (int j = 0; j < ST_R; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < K; ++k) {
    int idx[ST_R];
    bool any = false;
#pragma unroll
    for (int j = 0; j < ST_R; ++j) {
      const int row = row0 + 32 * j;
      idx[j] = (row < n_out) ? nbr[(size_t)k * ld + row] : -1;
      any |= idx[j] >= 0;
    }
    if (!__any(any)) continue;                                        // wave-uniform skip
    float xv[ST_R][CIN];
#pragma unroll
    for (int j = 0; j < ST_R; ++j) {
      const float* xr = x + (size_t)(idx[j] < 0 ? 0 : idx[j]) * ldx;  // missing neighbour: row 0, masked below
      st_load_row<CIN>(xr, xv[j]);
#pragma unroll
      for (int c = 0; c < CIN; ++c) xv[j][c] = idx[j] >= 0 ? xv[j][c] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      const float4 wv = *reinterpret_cast<const float4*>(&sW[(k * CIN + c) * ST_COUT + c4]);
#pragma unroll
      for (int j = 0; j < ST_R; ++j) {
        acc[j].x = fmaf(xv[j][c], wv.x, acc[j].x);
        acc[j].y = fmaf(xv[j][c], wv.y, acc[j].y);
        acc[j].z = fmaf(xv[j][c], wv.z, acc[j].z);
        acc[j].w = fmaf(xv[j][c], wv.w, acc[j].w);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < ST_R; ++j) {
    const int row = row0 + 32 * j;
    if (row < n_out) irx_st4(y, (size_t)row * ST_COUT + c4, y_bf, acc[j]);
  }
}

// part[blk][k*CIN + c][n].  Per 64-row chunk the dense im2col tile X[row][k*CIN + c] (zeros where the neighbour is
// missing) is assembled directly in LDS from one coalesced table read + the valid neighbours' CIN-float rows, and
// dW += X^T * dY runs on MFMA with the 64 rows as the reduction dimension (16x16x4 fp32; M = 27*CIN padded to 192).
// The im2col matrix is never materialised in HBM and there are no atomics; workgroup partials are reduced by
// k_wgrad_reduce (deterministic). Wave w owns M-tiles {w, w+4, w+8} x both N-tiles (6 accumulators).
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CIN>
__global__ __launch_bounds__(256, 2) void k_stem_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                       const int32_t* __restrict__ nbr, int ld, int n_out,
                                                       int rows_per_block, float* __restrict__ part, int ldx,
                                                       int dy_bf) {
  constexpr int K = 27;
  constexpr int MREAL = K * CIN;                 // 189 for CIN = 7
  constexpr int MT = (MREAL + 15) / 16;          // 12
  constexpr int MPW = (MT + 3) / 4;              // M-tiles per wave (3)
  // Row stride 209 = 17 (mod 64): the im2col WRITES have lane == row at a fixed column — with the former stride 208
  // (16 mod 64) they hit 4 banks (16-way conflict, the kernel's bound); 17 is coprime to 64 (conflict-free writes) and
  // still keeps the four rows of an MFMA fragment read >= 16 banks apart (a 2-way conflict on 3 banks of 64).
  constexpr int LDX = MT * 16 + 17;
  constexpr int LDD = ST_COUT + 16;              // 48
  __shared__ __attribute__((aligned(16))) float sX[64 * LDX];
  __shared__ __attribute__((aligned(16))) float sD[64 * LDD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mm = lane & 15, g4 = lane >> 4;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > n_out) r1 = n_out;
  f32x4 acc[MPW][2];
#pragma unroll
  for (int a = 0; a < MPW; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) acc[a][b2] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < 64 * LDX; i += 256) sX[i] = 0.f;     // pad columns stay zero for the whole kernel

  for (int q0 = r0; q0 < r1; q0 += 64) {
    __syncthreads();                                         // previous chunk's fragment reads are done
    {
      // 27 x 64 (offset, row) entries, 7 per thread: ALL table reads first, then ALL row reads, then the LDS writes —
      // one entry at a time the loop was a chain of dependent loads (index -> row) with nothing in flight
      // (102 -> 77 us; the same batching in k_stem_fwd, whose 4 rows per thread already overlap, cost occupancy: 80 -> 98 us)
      constexpr int EPT = (K * 64 + 255) / 256;
      int idx[EPT];
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * 256;
        const int k = i >> 6, r = i & 63;
        idx[e] = (i < K * 64 && q0 + r < r1) ? nbr[(size_t)k * ld + q0 + r] : -1;
      }
      float v[EPT][CIN];
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const float* xr = x + (size_t)(idx[e] < 0 ? 0 : idx[e]) * ldx;
        st_load_row<CIN>(xr, v[e]);
#pragma unroll
        for (int c = 0; c < CIN; ++c) v[e][c] = idx[e] >= 0 ? v[e][c] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * 256;
        if (i < K * 64) {
          const int k = i >> 6, r = i & 63;
#pragma unroll
          for (int c = 0; c < CIN; ++c) sX[r * LDX + k * CIN + c] = v[e][c];
        }
      }
    }
    for (int f = tid; f < 64 * (ST_COUT / 4); f += 256) {
      const int r = f >> 3, c4 = (f & 7) * 4;
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q0 + r < r1) d = irx_ld4(dy, (size_t)(q0 + r) * ST_COUT + c4, dy_bf);
      *reinterpret_cast<float4*>(&sD[r * LDD + c4]) = d;
    }
    __syncthreads();
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
      const int rr = ks * 4 + g4;
      const float b0 = sD[rr * LDD + mm], b1 = sD[rr * LDD + 16 + mm];
#pragma unroll
      for (int a = 0; a < MPW; ++a) {
        const int mt = wave + 4 * a;
        if (mt < MT) {
          const float av = sX[rr * LDX + mt * 16 + mm];      // A[m][kk = row]
          acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[a][0], 0, 0, 0);
          acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[a][1], 0, 0, 0);
        }
      }
    }
  }
  float* out = part + (size_t)blockIdx.x * MREAL * ST_COUT;
#pragma unroll
  for (int a = 0; a < MPW; ++a) {
    const int mt = wave + 4 * a;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mrow = mt * 16 + g4 * 4 + r;               // = k*CIN + c
        if (mt < MT && mrow < MREAL) out[(size_t)mrow * ST_COUT + t * 16 + mm] = acc[a][t][r];
      }
  }
}

// Forward as a dense im2col tile + fp32 MFMA (round 5). k_stem_fwd above executes 27 x CIN x 32 MASKED multiply-
// adds per row on the vector ALU whether a neighbour exists or not (19 % useful at 5 neighbours per voxel) and is bound by that: 124 us
// on the 489 k-voxel level. Here, per 64-row chunk, the im2col tile X[row][k * CIN + c] (zeros where the neighbour is missing) is
// assembled in LDS exactly as k_stem_wgrad does — one coalesced table read, the valid neighbours' rows as two 16-byte loads — and
// Y = X W runs on v_mfma_f32_16x16x4_f32: M = 64 rows (wave w owns rows 16 w ..), N = 32, K = 27 CIN padded to 192. The B operand
// (W, 24 KB) lives in REGISTERS — lane (n = l & 15, g = l >> 4) holds W[4 s + g][16 t + n] for the 48 steps s and both column
// halves t, 96 VGPRs, loaded once per workgroup — so LDS only holds the X tile (53.5 KB) and a 9 KB output staging tile: two
// workgroups per CU, whose gather / multiply / store phases overlap each other. The accumulators are transposed through LDS once
// per chunk for whole-row stores.
template <int CIN>
__global__ __launch_bounds__(256, 2) void k_stem_fwd_mfma(const float* __restrict__ x, const float* __restrict__ w,
                                                          const int32_t* __restrict__ nbr, int ld, int n_out,
                                                          int rows_per_block, float* __restrict__ y, int ldx, int y_bf) {
  constexpr int K = 27;
  constexpr int MREAL = K * CIN;                 // 189 for CIN = 7
  constexpr int KS = (MREAL + 3) / 4;            // reduction steps of 4 (48)
  constexpr int LDX = ((MREAL + 15) / 16) * 16 + 17;   // 209: see k_stem_wgrad (conflict-free column writes, <= 2-way fragment reads)
  constexpr int LDO = ST_COUT + 4;               // output staging [64][36]
  __shared__ __attribute__((aligned(16))) float sX[64 * LDX];
  __shared__ __attribute__((aligned(16))) float sO[64 * LDO];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mm = lane & 15, g4 = lane >> 4;
  float wr[KS][2];
#pragma unroll
  for (int s2 = 0; s2 < KS; ++s2) {
    const int k = 4 * s2 + g4;
    wr[s2][0] = (k < MREAL) ? w[(size_t)k * ST_COUT + mm] : 0.f;
    wr[s2][1] = (k < MREAL) ? w[(size_t)k * ST_COUT + 16 + mm] : 0.f;
  }
  for (int i = tid; i < 64 * LDX; i += 256) sX[i] = 0.f;     // pad columns stay zero for the whole kernel
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > n_out) r1 = n_out;
  // software pipeline: chunk q + 1's table entries are requested before chunk q's MFMA loop and its rows from the middle of it, so
  // the two dependent round trips of the gather travel under the matrix work instead of in front of it
  constexpr int EPT = (K * 64 + 255) / 256;
  int idx[EPT];
  float v[EPT][CIN];
  auto fetch_idx = [&](int q) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = tid + e * 256;
      const int k = i >> 6, r = i & 63;
      idx[e] = (i < K * 64 && q + r < r1) ? nbr[(size_t)k * ld + q + r] : -1;
    }
  };
  auto fetch_rows = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      // (predicating the load on idx >= 0 — four slots of five are empty — measured 6 % SLOWER: the branch per element costs more
      //  than the dummy, always-cached load of row 0 it saves)
      const float* xr = x + (size_t)(idx[e] < 0 ? 0 : idx[e]) * ldx;
      st_load_row<CIN>(xr, v[e]);
    }
  };
  if (r0 < r1) {
    fetch_idx(r0);
    fetch_rows();
  }
  for (int q0 = r0; q0 < r1; q0 += 64) {
    __syncthreads();                                         // previous chunk's fragment reads are done
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = tid + e * 256;
      if (i < K * 64) {
        const int k = i >> 6, r = i & 63;
#pragma unroll
        for (int c = 0; c < CIN; ++c) sX[r * LDX + k * CIN + c] = idx[e] >= 0 ? v[e][c] : 0.f;
      }
    }
    __syncthreads();
    const bool more = q0 + 64 < r1;
    if (more) fetch_idx(q0 + 64);
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    const float* xa = sX + (16 * wave + mm) * LDX + g4;
#pragma unroll
    for (int s2 = 0; s2 < KS / 2; ++s2) {
      const float av = xa[4 * s2];
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wr[s2][0], d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wr[s2][1], d1, 0, 0, 0);
    }
    if (more) fetch_rows();
#pragma unroll
    for (int s2 = KS / 2; s2 < KS; ++s2) {
      const float av = xa[4 * s2];
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wr[s2][0], d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wr[s2][1], d1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sO[(16 * wave + 4 * g4 + r) * LDO + mm] = d0[r];
      sO[(16 * wave + 4 * g4 + r) * LDO + 16 + mm] = d1[r];
    }
    __syncthreads();
    for (int f = tid; f < 64 * (ST_COUT / 4); f += 256) {
      const int r = f >> 3, c4 = (f & 7) * 4;
      if (q0 + r < r1) irx_st4(y, (size_t)(q0 + r) * ST_COUT + c4, y_bf, *reinterpret_cast<const float4*>(&sO[r * LDO + c4]));
    }
  }
}

bool irx_stem_supported(int K, int cin, int cout) { return K == 27 && cout == ST_COUT && cin >= 1 && cin <= 8; }

int irx_stem_fwd_launch(const float* x, const float* w, const int32_t* nbr, int ld, int n_out, int K, int cin,
                        float* y, hipStream_t st, int ldx, int y_bf) {
  if (ldx <= 0) ldx = cin;
  // im2col + MFMA form (k_stem_fwd_mfma) for the usual 7-channel stem; "stem_mfma" / IRX_STEM_MFMA=0 keeps the vector-ALU kernel (dev A/B; the
  // two differ in summation order only)
  const bool mfma = irx_knob(IRX_KNOB_STEM_MFMA) != 0;
  if (mfma && K == 27 && cin == 7 && n_out >= 64) {
    int blocks = irx_cdiv(n_out, 64 * 4);                    // >= 4 chunks of 64 rows per workgroup
    if (blocks > 2048) blocks = 2048;
    int rpb = irx_cdiv(n_out, blocks);
    rpb = irx_cdiv(rpb, 64) * 64;
    blocks = irx_cdiv(n_out, rpb);
    irx_bracket_begin(st);
    k_stem_fwd_mfma<7><<<blocks, 256, 0, st>>>(x, w, nbr, ld, n_out, rpb, y, ldx, y_bf);
    irx_bracket_end(st);
    IRX_CHECK_LAUNCH("irx_spconv_fwd(stem, mfma)");
    return IRX_OK;
  }
  const int grid = irx_cdiv(n_out, 32 * ST_R);
  irx_bracket_begin(st);
  switch (cin) {
    case 1: k_stem_fwd<1><<<grid, 256, 0, st>>>(x, w, nbr, ld, n_out, K, y, ldx, y_bf); break;
    case 2: k_stem_fwd<2><<<grid, 256, 0, st>>>(x, w, nbr, ld, n_out, K, y, ldx, y_bf); break;
    case 3: k_stem_fwd<3><<<grid, 256, 0, st>>>(x, w, nbr, ld, n_out, K, y, ldx, y_bf); break;
    case 4: k_stem_fwd<4><<<grid, 256, 0, st>>>(x, w, nbr, ld, n_out, K, y, ldx, y_bf); break;
    case 5: k_stem_fwd<5><<<grid, 256, 0, st>>>(x, w, nbr, ld, n_out, K, y, ldx, y_bf); break;
    case 6: k_stem_fwd<6><<<grid, 256, 0, st>>>(x, w, nbr, ld, n_out, K, y, ldx, y_bf); break;
    case 7: k_stem_fwd<7><<<grid, 256, 0, st>>>(x, w, nbr, ld, n_out, K, y, ldx, y_bf); break;
    default: k_stem_fwd<8><<<grid, 256, 0, st>>>(x, w, nbr, ld, n_out, K, y, ldx, y_bf); break;
  }
  irx_bracket_end(st);
  IRX_CHECK_LAUNCH("irx_spconv_fwd(stem)");
  return IRX_OK;
}

int irx_stem_wgrad_blocks(int n_out) {
  int b = irx_cdiv(n_out, 256);          // >= 4 chunks of 64 rows per workgroup
  static const int cap = getenv("IRX_STEM_WGRAD_BLOCKS") ? atoi(getenv("IRX_STEM_WGRAD_BLOCKS")) : 512;   // (1024: the same kernel time, a reduce twice as long; 256: k_stem_wgrad 81 -> 109 us) dev knob
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return b;
}

int irx_stem_wgrad_launch(const float* x, const float* dy, const int32_t* nbr, int ld, int n_out, int cin,
                          int blocks, float* part, hipStream_t st, int ldx, int dy_bf) {
  if (ldx <= 0) ldx = cin;
  int rpb = irx_cdiv(n_out, blocks);
  rpb = irx_cdiv(rpb, 64) * 64;
  irx_bracket_begin(st);
  switch (cin) {
    case 1: k_stem_wgrad<1><<<blocks, 256, 0, st>>>(x, dy, nbr, ld, n_out, rpb, part, ldx, dy_bf); break;
    case 2: k_stem_wgrad<2><<<blocks, 256, 0, st>>>(x, dy, nbr, ld, n_out, rpb, part, ldx, dy_bf); break;
    case 3: k_stem_wgrad<3><<<blocks, 256, 0, st>>>(x, dy, nbr, ld, n_out, rpb, part, ldx, dy_bf); break;
    case 4: k_stem_wgrad<4><<<blocks, 256, 0, st>>>(x, dy, nbr, ld, n_out, rpb, part, ldx, dy_bf); break;
    case 5: k_stem_wgrad<5><<<blocks, 256, 0, st>>>(x, dy, nbr, ld, n_out, rpb, part, ldx, dy_bf); break;
    case 6: k_stem_wgrad<6><<<blocks, 256, 0, st>>>(x, dy, nbr, ld, n_out, rpb, part, ldx, dy_bf); break;
    case 7: k_stem_wgrad<7><<<blocks, 256, 0, st>>>(x, dy, nbr, ld, n_out, rpb, part, ldx, dy_bf); break;
    default: k_stem_wgrad<8><<<blocks, 256, 0, st>>>(x, dy, nbr, ld, n_out, rpb, part, ldx, dy_bf); break;
  }
  irx_bracket_end(st);
  IRX_CHECK_LAUNCH("irx_spconv_wgrad(stem)");
  return IRX_OK;
}
