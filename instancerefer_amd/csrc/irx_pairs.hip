// irx_pairs.hip — per-offset compacted pair lists ("rulebook") of a neighbour table and the weight-gradient that
// streams them.  For offset k the valid entries of nbr[k][:] become in_list[k][0..cnt_k) (input rows) and
// out_list[k][0..cnt_k) (output rows), in ascending output-row order. Built once per (level, table) on the device
// (tile counts -> prefix -> ballot/popcount scatter; counts stay in device memory, no host sync) and reused by every
// weight-gradient of that level (both convs of a ResidualBlock).
//
// k_wgrad_pairs: workgroup (split s, offset k) takes an equal share of list k and processes it in DENSE stages of 64
// pairs (the table-driven kernel gets only ~1/3 of a 64-row chunk filled): gather x[in] and dy[out] rows (16 B/lane)
// -> LDS -> v_mfma_f32_16x16x4_f32 with the pairs as the reduction dimension; the next stage's indices and rows are
// prefetched into registers before the current stage's MFMAs. Per-(split, offset) partial sums, deterministic reduce.
#include <stdlib.h>
#include <type_traits>
#include "irx_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
static inline hipStream_t S(void* s) { return (hipStream_t)s; }

#define PB_TILE 2048   // rows per count/scatter workgroup (256 threads x 8)
#define IRX_PAIRS_MAX_TABLES 16

__global__ __launch_bounds__(256) void k_pairs_count(const int32_t* __restrict__ nbr, int ld, int n_out,
                                                     int32_t* __restrict__ tile_counts, int ntiles) {
  __shared__ int s_w[4];
  const int k = blockIdx.y, tile = blockIdx.x;
  int cnt = 0;
  for (int it = 0; it < 8; ++it) {
    const int q = tile * PB_TILE + it * 256 + threadIdx.x;
    if (q < n_out && nbr[(size_t)k * ld + q] >= 0) ++cnt;
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) tile_counts[k * ntiles + tile] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(256) void k_pairs_write(const int32_t* __restrict__ nbr, int ld, int n_out,
                                                     const int32_t* __restrict__ tile_counts, int ntiles,
                                                     int32_t* __restrict__ in_list, int32_t* __restrict__ out_list,
                                                     int ldp, int32_t* __restrict__ counts) {
  __shared__ int s_red[4];
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int k = blockIdx.y, tile = blockIdx.x;
  int acc = 0;
  for (int t = threadIdx.x; t < tile; t += 256) acc += tile_counts[k * ntiles + t];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    s_base = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    if (tile == ntiles - 1) counts[k] = s_base + tile_counts[k * ntiles + tile];
  }
  __syncthreads();
  int running = s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int it = 0; it < 8; ++it) {
    const int q = tile * PB_TILE + it * 256 + threadIdx.x;
    int idx = -1;
    if (q < n_out) idx = nbr[(size_t)k * ld + q];
    const unsigned long long b = __ballot(idx >= 0);
    if (lane == 0) s_wave[wave] = __popcll(b);
    __syncthreads();
    int wbase = 0, total = 0;
    for (int w = 0; w < 4; ++w) {
      const int c = s_wave[w];
      if (w < wave) wbase += c;
      total += c;
    }
    if (idx >= 0) {
      const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      const int pos = running + wbase + __popcll(b & lt);
      in_list[(size_t)k * ldp + pos] = idx;
      out_list[(size_t)k * ldp + pos] = q;
    }
    running += total;
    __syncthreads();
  }
}

// Work split proportional to the list lengths (they differ ~3x between the centre and the corner offsets and live in
// device memory): with G workgroups for T = sum_k ceil(cnt_k / 64) stages, offset k is cut into
// ceil(nst_k / ceil(T / G)) shares (1 .. smax), so every workgroup gets about the same number of 64-pair stages.
// (Equal splits per offset made the centre offset's workgroups the critical path at ~2.8x the average work.)
// xcd != 0 (k_wgrad3 on levels with enough rows): the share count is a multiple of 8 — every list is first cut into 8
// equal segments, one per XCD (the lists are in ascending output-row order, so segment x of every offset covers about
// the same rows), and a segment into m shares (pairs_xcd_m); the kernel maps workgroup b to XCD segment b % 8.
__device__ __forceinline__ int pairs_xcd_m(int nst, int tgt, int smax) {
  int m = (nst + 4 * tgt) / (8 * tgt);                   // nst / (8 tgt), rounded to nearest
  const int cap = smax >> 3;
  if (m > cap) m = cap;
  if (m < 1) m = 1;
  return m;
}
__device__ __forceinline__ int pairs_shares(const int32_t* __restrict__ counts, int K, int k, int G, int smax, int xcd = 0) {
  int T = 0;
  for (int j = 0; j < K; ++j) T += (counts[j] + 63) >> 6;
  int tgt = (T + G - 1) / G;
  if (tgt < 1) tgt = 1;
  const int nst = (counts[k] + 63) >> 6;
  if (xcd) return 8 * pairs_xcd_m(nst, tgt, smax);
  int sp = (nst + tgt - 1) / tgt;
  if (sp < 1) sp = 1;
  if (sp > smax) sp = smax;
  return sp;
}

// The work unit of a weight-gradient workgroup: (share s of offset k, of nsplit). xcd = 0: blockIdx = (s, k); xcd != 0: the
// XCD-segment mapping of a 1-D grid (pairs_shares above; k_wgrad3 explains it). false: this workgroup has nothing to do.
__device__ __forceinline__ bool pairs_unit(const int32_t* __restrict__ counts, int K, int G, int smax, int xcd, int& s, int& k,
                                           int& nsplit) {
  if (!xcd) {
    s = blockIdx.x; k = blockIdx.y;
    nsplit = pairs_shares(counts, K, k, G, smax);
    return s < nsplit;
  }
  const int xseg = blockIdx.x & 7, r = blockIdx.x >> 3;
  int T = 0;
  for (int j = 0; j < K; ++j) T += (counts[j] + 63) >> 6;
  int tgt = (T + G - 1) / G;
  if (tgt < 1) tgt = 1;
  int cum = 0;
  k = -1; s = 0; nsplit = 8;
  for (int j = 0; j < K; ++j) {
    const int m = pairs_xcd_m((counts[j] + 63) >> 6, tgt, smax);
    if (k < 0 && r < cum + m) { k = j; s = xseg * m + (r - cum); nsplit = 8 * m; }
    cum += m;
  }
  return k >= 0;
}

// part[s][k][c][n] = sum over share s of list k:  x[in][c] * dy[out][n]   (s < pairs_shares(k); grid = (smax, K))
// ST (bf16 storage, with BF only): x and dy rows are bf16 in HBM; a staging thread loads 8 bytes (its 4 channels) and
// widens them to fp32 when it writes the LDS tiles (the fragment reads below are unchanged).
// Dev-only ablation of k_wgrad_pairs' fp32 full-stage chain (results WRONG, timing only): 1 = no next-stage row loads,
// 2 = no LDS fragment reads, 4 = no MFMA.
#ifndef IRX_WP_ABL
#define IRX_WP_ABL 0
#endif
template <bool ST>
__device__ __forceinline__ typename std::conditional<ST, uint2, float4>::type wp_ld(const float* __restrict__ p, size_t elem) {
  if constexpr (ST) return *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p) + elem);
  else return *reinterpret_cast<const float4*>(p + elem);
}
__device__ __forceinline__ float4 wp_f4(float4 v) { return v; }
__device__ __forceinline__ float4 wp_f4(uint2 v) { return irx_bf4_to_f4(v); }

// STD: element type of dy when it differs from x's (the wide stem in bf16 storage mode: fp32 input rows, bf16 gradient);
// ldx: row stride of x in elements (CIN for a dense tensor; the wide stem reads the leading CIN columns of wider rows).
template <int CIN, int COUT, bool BF, bool ST, bool STD = ST>
__global__ __launch_bounds__(256, 2) void k_wgrad_pairs(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const int32_t* __restrict__ in_list,
                                                        const int32_t* __restrict__ out_list, int ldp,
                                                        const int32_t* __restrict__ counts, int K, int G, int smax,
                                                        float* __restrict__ part, int ldx = CIN, int xcd = 0) {
  constexpr int TC = CIN / 16, TN = COUT / 16;
  constexpr int CW = (TC >= 4) ? TC / 4 : 1;             // c-tiles per wave
  constexpr int NW = (TC >= 4) ? TN : TN / (4 / TC);     // n-tiles per wave
  constexpr int LDX = CIN + 16, LDD = COUT + 16;         // consecutive pairs 16 banks apart
  constexpr int LPX = CIN / 4, LPD = COUT / 4;           // lanes (float4) per row
  constexpr int PX = 256 / LPX, PD = 256 / LPD;          // pairs per workgroup pass
  constexpr int NX = 64 / PX, ND = 64 / PD;              // passes per 64-pair stage
  __shared__ __attribute__((aligned(16))) float sX[64 * LDX];
  __shared__ __attribute__((aligned(16))) float sD[64 * LDD];
  __shared__ int sIn[2][64];
  __shared__ int sOutRow[2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g4 = lane >> 4;
  int s, k, nsplit;
  if (!pairs_unit(counts, K, G, smax, xcd, s, k, nsplit)) return;      // block-uniform
  const int ct0 = (TC >= 4) ? wave * CW : (wave % TC);
  const int nt0 = (TC >= 4) ? 0 : (wave / TC) * NW;
  const int cnt = counts[k];
  // equal shares of this offset's stages
  const int nst = (cnt + 63) / 64;
  const int st0 = (int)(((long long)nst * s) / nsplit), st1 = (int)(((long long)nst * (s + 1)) / nsplit);
  const int32_t* il = in_list + (size_t)k * ldp;
  const int32_t* ol = out_list + (size_t)k * ldp;
  const int xr = tid / LPX, xc = (tid % LPX) * 4;        // this thread's pair / column in an x pass
  const int dr = tid / LPD, dc = (tid % LPD) * 4;

  f32x4 acc[CW][NW];
#pragma unroll
  for (int a = 0; a < CW; ++a)
#pragma unroll
    for (int b = 0; b < NW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  static_assert(BF || !(ST || STD), "bf16 storage implies bf16 operands");
  using RT = typename std::conditional<ST, uint2, float4>::type;
  using RD = typename std::conditional<STD, uint2, float4>::type;
  RT rx[NX];
  RD rd[ND];
  int par = 0;
  // prologue: indices + rows of the first stage, indices of the second
  int nin = -1, nout = -1;
  if (st0 < st1) {
    if (tid < 64) {
      const int p = st0 * 64 + tid;
      sIn[0][tid] = p < cnt ? il[p] : -1;
      sOutRow[0][tid] = p < cnt ? ol[p] : -1;
      const int p1 = p + 64;
      if (st0 + 1 < st1) {
        nin = p1 < cnt ? il[p1] : -1;
        nout = p1 < cnt ? ol[p1] : -1;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int idx = sIn[0][xr + i * PX];
      rx[i] = idx >= 0 ? wp_ld<ST>(x, (size_t)idx * ldx + xc) : RT{};
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int idx = sOutRow[0][dr + i * PD];
      rd[i] = idx >= 0 ? wp_ld<STD>(dy, (size_t)idx * COUT + dc) : RD{};
    }
  }
  for (int st = st0; st < st1; ++st) {
    __syncthreads();                                     // previous stage's fragment reads are done
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): the rows requested during the previous chain
#pragma unroll
    for (int i = 0; i < NX; ++i) *reinterpret_cast<float4*>(&sX[(xr + i * PX) * LDX + xc]) = wp_f4(rx[i]);
#pragma unroll
    for (int i = 0; i < ND; ++i) *reinterpret_cast<float4*>(&sD[(dr + i * PD) * LDD + dc]) = wp_f4(rd[i]);
    // next stage's indices (other parity buffer), visible after the barrier below. They were requested ONE STAGE AGO
    // (nin / nout, landed under the vmcnt(0) above): loading them here put a global round trip of the index lists —
    // streamed once, so from HBM — in front of every stage's barrier
    const int parn = par ^ 1;
    if (st + 1 < st1 && tid < 64) {
      sIn[parn][tid] = nin;
      sOutRow[parn][tid] = nout;
    }
    if (st + 2 < st1 && tid < 64) {
      const int p = (st + 2) * 64 + tid;
      nin = p < cnt ? il[p] : -1;
      nout = p < cnt ? ol[p] : -1;
    }
    __syncthreads();
    const bool has_next = st + 1 < st1;
    int npairs = cnt - st * 64;
    if (npairs > 64) npairs = 64;
    const int nks = (npairs + 3) >> 2;
    if constexpr (BF) {
      // bf16 operands (irx_set_compute_dtype(1)): a lane of the 16x16x16 MFMA holds FOUR consecutive pairs of its
      // channel (A) / column (B): 4 LDS reads + 2 packed converts per fragment, one MFMA where fp32 issues four.
      // Rows of missing pairs are zero in LDS, so partial stages just run fewer 16-pair steps.
      const int n16 = has_next ? 4 : ((npairs + 15) >> 4);
      // raw fp32 fragments of 16-pair step k16 + 1 are read from LDS before the MFMAs of step k16
      float a[CW][4], b[NW][4];
      auto read16 = [&](int k16, float (&ra)[CW][4], float (&rb)[NW][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          const int pp = 16 * k16 + 4 * g4 + i4;
#pragma unroll
          for (int i = 0; i < CW; ++i) ra[i][i4] = sX[pp * LDX + (ct0 + i) * 16 + m];
#pragma unroll
          for (int i = 0; i < NW; ++i) rb[i][i4] = sD[pp * LDD + (nt0 + i) * 16 + m];
        }
      };
      read16(0, a, b);
#pragma unroll
      for (int k16 = 0; k16 < 4; ++k16) {
        if (k16 < n16) {
          if (has_next) {                                  // next stage's rows, spread over the chain as in fp32
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
              const int ks = k16 * 4 + i4;
              if (ks < NX) {
                const int idx = sIn[parn][xr + ks * PX];
                const RT v = wp_ld<ST>(x, (size_t)(idx < 0 ? 0 : idx) * ldx + xc);
                rx[ks] = idx >= 0 ? v : RT{};
              }
              if (ks < ND) {
                const int idx = sOutRow[parn][dr + ks * PD];
                const RD v = wp_ld<STD>(dy, (size_t)(idx < 0 ? 0 : idx) * COUT + dc);
                rd[ks] = idx >= 0 ? v : RD{};
              }
            }
          }
          s16x4 pa[CW], pb[NW];
#pragma unroll
          for (int i = 0; i < CW; ++i) pa[i] = irx_frag_bf16(irx_pk_bf16(a[i][0], a[i][1]), irx_pk_bf16(a[i][2], a[i][3]));
#pragma unroll
          for (int i = 0; i < NW; ++i) pb[i] = irx_frag_bf16(irx_pk_bf16(b[i][0], b[i][1]), irx_pk_bf16(b[i][2], b[i][3]));
          if (k16 + 1 < 4 && k16 + 1 < n16) read16(k16 + 1, a, b);   // (a, b are free again: packed above)
#pragma unroll
          for (int i = 0; i < CW; ++i)
#pragma unroll
            for (int jn = 0; jn < NW; ++jn)
              acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa[i], pb[jn], acc[i][jn], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else if (has_next) {
      // A stage that has a successor is a full one (16 k-steps): the next stage's rows are requested from INSIDE its
      // MFMA chain, one x and one dy load per step, so the wave never queues at the vector-memory pipe with an idle
      // MFMA pipe behind it (the lesson of k_spconv2). Loads are unconditional (missing pairs re-read row 0 and are
      // zeroed by a select), so the chain has no branches.
      // fragments of step ks + 1 are read from LDS BEFORE the MFMAs of step ks (the sched_barrier that pins the global
      // loads into the chain would otherwise make every step wait out its own LDS latency)
      float a[CW], b[NW];
#pragma unroll
      for (int i = 0; i < CW; ++i) a[i] = sX[g4 * LDX + (ct0 + i) * 16 + m];     // A[m = c][kk = pair]
#pragma unroll
      for (int i = 0; i < NW; ++i) b[i] = sD[g4 * LDD + (nt0 + i) * 16 + m];     // B[kk = pair][n]
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks < NX && !(IRX_WP_ABL & 1)) {
          const int idx = sIn[parn][xr + ks * PX];
          const float4 v = *reinterpret_cast<const float4*>(x + (size_t)(idx < 0 ? 0 : idx) * ldx + xc);
          rx[ks] = idx >= 0 ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (ks < ND && !(IRX_WP_ABL & 1)) {
          const int idx = sOutRow[parn][dr + ks * PD];
          const float4 v = *reinterpret_cast<const float4*>(dy + (size_t)(idx < 0 ? 0 : idx) * COUT + dc);
          rd[ks] = idx >= 0 ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float an[CW], bn[NW];
        if (ks + 1 < 16 && !(IRX_WP_ABL & 2)) {
          const int pp = (ks + 1) * 4 + g4;
#pragma unroll
          for (int i = 0; i < CW; ++i) an[i] = sX[pp * LDX + (ct0 + i) * 16 + m];
#pragma unroll
          for (int i = 0; i < NW; ++i) bn[i] = sD[pp * LDD + (nt0 + i) * 16 + m];
        }
#pragma unroll
        for (int i = 0; i < CW; ++i)
#pragma unroll
          for (int jn = 0; jn < NW; ++jn) {
            if constexpr ((IRX_WP_ABL & 4) != 0) acc[i][jn][0] += a[i] * b[jn];
            else acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[jn], acc[i][jn], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < 16 && !(IRX_WP_ABL & 2)) {
#pragma unroll
          for (int i = 0; i < CW; ++i) a[i] = an[i];
#pragma unroll
          for (int i = 0; i < NW; ++i) b[i] = bn[i];
        }
      }
    } else {
      for (int ks = 0; ks < nks; ++ks) {
        const int pp = ks * 4 + g4;
        float a[CW], b[NW];
#pragma unroll
        for (int i = 0; i < CW; ++i) a[i] = sX[pp * LDX + (ct0 + i) * 16 + m];   // A[m = c][kk = pair]
#pragma unroll
        for (int i = 0; i < NW; ++i) b[i] = sD[pp * LDD + (nt0 + i) * 16 + m];   // B[kk = pair][n]
#pragma unroll
        for (int i = 0; i < CW; ++i)
#pragma unroll
          for (int jn = 0; jn < NW; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[jn], acc[i][jn], 0, 0, 0);
      }
    }
    par = parn;
  }
  float* out = part + ((size_t)s * K + k) * CIN * COUT;
#pragma unroll
  for (int i = 0; i < CW; ++i)
#pragma unroll
    for (int jn = 0; jn < NW; ++jn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = (ct0 + i) * 16 + g4 * 4 + r;
        const int n = (nt0 + jn) * 16 + m;
        out[(size_t)c * COUT + n] = acc[i][jn][r];
      }
}

// ---- fp32 weight-gradient, second generation (round 3): ONE barrier per stage, nothing outside the MFMA chain -----------
// k_wgrad_pairs above alternates "wait for the rows, write them to LDS, barrier" with the MFMA chain, and its ablation showed
// the two parts simply ADD (MFMA alone 187 us, everything else alone 97 us, together 284 us at 128 x 128 on the 81 k level):
// two resident workgroups whose phases drift freely overlap a chain with the other's non-MFMA phase only by chance (the
// `phases` probe: 71 % of the MFMA peak at two workgroups per CU). Here the overlap is built into every wave instead:
//   * stages of 32 pairs, the LDS tiles double-buffered (same 74 KB at 128 x 128 as one 64-pair stage: still two workgroups per CU);
//   * during the chain over stage t (8 k-steps) a wave requests the rows of stage t + 2 (k-steps 0..3, into the register set
//     that stage t's rows left) and writes the rows of stage t + 1 — requested one chain earlier — to the other LDS buffer
//     (k-steps 4..7); the pair indices run three to four stages ahead through a 4-slot LDS ring;
//   * the single barrier at the top of a stage publishes buffer t and retires the reads of buffer t - 1.
// Same pair order, same 4-pair MFMA groups, same (split, offset) partials as k_wgrad_pairs: results are bit-identical
// (missing pairs are zero rows; a partial last stage runs its full chain over zeros).
template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void k_wgrad_pairs2(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const int32_t* __restrict__ in_list,
                                                         const int32_t* __restrict__ out_list, int ldp,
                                                         const int32_t* __restrict__ counts, int K, int G, int smax,
                                                         float* __restrict__ part, int ldx = CIN, int xcd = 0) {
  constexpr int SP = 32;                                  // pairs per stage
  constexpr int TC = CIN / 16, TN = COUT / 16;
  constexpr int CW = (TC >= 4) ? TC / 4 : 1;             // c-tiles per wave
  constexpr int NW = (TC >= 4) ? TN : TN / (4 / TC);     // n-tiles per wave
  constexpr int LDX = CIN + 16, LDD = COUT + 16;         // consecutive pairs 16 banks apart
  constexpr int LPX = CIN / 4, LPD = COUT / 4;           // lanes (float4) per row
  constexpr int PX = 256 / LPX, PD = 256 / LPD;          // pairs per workgroup pass
  constexpr int NX = SP / PX, ND = SP / PD;              // passes per stage (1, 2 or 4)
  static_assert(NX >= 1 && NX <= 4 && ND >= 1 && ND <= 4, "stage passes");
  __shared__ __attribute__((aligned(16))) float sX[2][SP * LDX];
  __shared__ __attribute__((aligned(16))) float sD[2][SP * LDD];
  __shared__ int sIdx[4][64];                             // ring of stages: [0, 32) input rows, [32, 64) output rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g4 = lane >> 4;
  int s, k, nsplit;
  if (!pairs_unit(counts, K, G, smax, xcd, s, k, nsplit)) return;      // block-uniform
  const int ct0 = (TC >= 4) ? wave * CW : (wave % TC);
  const int nt0 = (TC >= 4) ? 0 : (wave / TC) * NW;
  const int cnt = counts[k];
  // the share of this workgroup in 64-pair units (k_pairs_reduce and the v1 kernel count the same way), walked in 32-pair stages
  const int nst = (cnt + 63) / 64;
  const int st0 = (int)(((long long)nst * s) / nsplit), st1 = (int)(((long long)nst * (s + 1)) / nsplit);
  // (always an even number of stages — the loop below runs them in unconditional pairs, one per register set / LDS buffer:
  // with a conditional second half the compiler sinks the first half's row loads into it, out of the chain they are meant to
  // hide behind; at most the last 32-pair stage of an offset's list is empty)
  const int s0 = 2 * st0, s1 = 2 * st1;
  const int n_it = s1 - s0;
  const int xr = tid / LPX, xc = (tid % LPX) * 4;        // this thread's pair / column in an x pass
  const int dr = tid / LPD, dc = (tid % LPD) * 4;

  f32x4 acc[CW][NW];
#pragma unroll
  for (int a = 0; a < CW; ++a)
#pragma unroll
    for (int b = 0; b < NW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Every global read is a raw buffer load: an out-of-range offset returns zeros, so a missing pair needs neither a branch nor
  // a select on the loaded value (with plain loads the compiler predicates them, and every branch inside the chain costs the
  // exact vmcnt bookkeeping: it waits for ALL outstanding loads at the next LDS write). 32-bit offsets: the launcher checks
  // that both tensors stay below 2 GiB and falls back to k_wgrad_pairs otherwise.
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, 0x7FFFFFF0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc((void*)(in_list + (size_t)k * ldp), 0, cnt * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)(out_list + (size_t)k * ldp), 0, cnt * 4, 0x00020000);
  // lanes 0..31 of every wave: input row of pair j of a stage, lanes 32..63: its output row; -1 beyond the list / the share
  // (all four waves fetch and later store the same 64 values: no wave-dependent branch in the chain)
  // (the loaded words are combined one chain later: any use right behind the load would put a vmcnt wait — a global round
  // trip — into the chain; three plain ints, because a struct captured by the stage lambda was promoted to LDS)
#define WP2_LDIDX(stage_, vi_, vo_, inv_)                                              \
  {                                                                                    \
    const int p_ = (stage_) * SP + (lane & 31);                                        \
    inv_ = ((stage_) < s1 && p_ < cnt) ? 0 : -1;                                       \
    const unsigned off_ = inv_ ? OOB : (unsigned)p_ * 4u;                              \
    vi_ = __builtin_amdgcn_raw_buffer_load_b32(rs_i, off_, 0, 0);                      \
    vo_ = __builtin_amdgcn_raw_buffer_load_b32(rs_o, off_, 0, 0);                      \
  }
#define WP2_IDX(vi_, vo_, inv_) ((((lane & 32) ? (vo_) : (vi_))) | (inv_))
  const unsigned xcb = (unsigned)xc * 4u, dcb = (unsigned)dc * 4u, ldxb = (unsigned)ldx * 4u;
  auto x_off = [&](int slot, int i) __attribute__((always_inline)) -> unsigned {
    const int idx = sIdx[slot][xr + i * PX];
    return idx < 0 ? OOB : (unsigned)idx * ldxb + xcb;
  };
  auto d_off = [&](int slot, int i) __attribute__((always_inline)) -> unsigned {
    const int idx = sIdx[slot][32 + dr + i * PD];
    return idx < 0 ? OOB : (unsigned)idx * (unsigned)(COUT * 4) + dcb;
  };
  auto ldrow = [&](const __amdgpu_buffer_rsrc_t& r, unsigned off) __attribute__((always_inline)) -> float4 {
    auto v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return *reinterpret_cast<float4*>(&v);
  };

  float4 rx[2][NX], rd[2][ND];                            // register set q holds the rows of a stage of parity q
  int nvi = -1, nvo = -1, ninv = -1;
  if (n_it > 0) {
    int a0, b0, c0, a1, b1, c1, a2, b2, c2;
    WP2_LDIDX(s0, a0, b0, c0);
    WP2_LDIDX(s0 + 1, a1, b1, c1);
    WP2_LDIDX(s0 + 2, a2, b2, c2);
    WP2_LDIDX(s0 + 3, nvi, nvo, ninv);
    sIdx[0][lane] = WP2_IDX(a0, b0, c0);
    sIdx[1][lane] = WP2_IDX(a1, b1, c1);
    sIdx[2][lane] = WP2_IDX(a2, b2, c2);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NX; ++i) rx[0][i] = ldrow(rs_x, x_off(0, i));
#pragma unroll
    for (int i = 0; i < ND; ++i) rd[0][i] = ldrow(rs_d, d_off(0, i));
#pragma unroll
    for (int i = 0; i < NX; ++i) rx[1][i] = ldrow(rs_x, x_off(1, i));
#pragma unroll
    for (int i = 0; i < ND; ++i) rd[1][i] = ldrow(rs_d, d_off(1, i));
#pragma unroll
    for (int i = 0; i < NX; ++i) *reinterpret_cast<float4*>(&sX[0][(xr + i * PX) * LDX + xc]) = rx[0][i];
#pragma unroll
    for (int i = 0; i < ND; ++i) *reinterpret_cast<float4*>(&sD[0][(dr + i * PD) * LDD + dc]) = rd[0][i];
  }

  // one stage: P = parity of the stage relative to s0 (LDS buffer read, register set refilled)
  auto stage = [&](auto Pc, int rel) __attribute__((always_inline)) {
    constexpr int P = decltype(Pc)::value, Q = P ^ 1;
    sIdx[(rel + 3) & 3][lane] = WP2_IDX(nvi, nvo, ninv);   // requested one chain ago (the same value from every wave)
    __syncthreads();                                       // buffer P and the ring slot are published; reads of buffer Q retired
    const int slot2 = (rel + 2) & 3;
    unsigned ox[NX], od[ND];                               // byte offsets of the rows of stage rel + 2
#pragma unroll
    for (int i = 0; i < NX; ++i) ox[i] = x_off(slot2, i);
#pragma unroll
    for (int i = 0; i < ND; ++i) od[i] = d_off(slot2, i);
    const float* bx = sX[P];
    const float* bd = sD[P];
    float a[CW], b[NW];
#pragma unroll
    for (int i = 0; i < CW; ++i) a[i] = bx[g4 * LDX + (ct0 + i) * 16 + m];     // A[m = c][kk = pair]
#pragma unroll
    for (int i = 0; i < NW; ++i) b[i] = bd[g4 * LDD + (nt0 + i) * 16 + m];     // B[kk = pair][n]
#pragma unroll
    for (int ks = 0; ks < SP / 4; ++ks) {
      if (ks == 0) WP2_LDIDX(s0 + rel + 4, nvi, nvo, ninv);
      if (ks < NX) rx[P][ks] = ldrow(rs_x, ox[ks]);        // rows of stage rel + 2
      if (ks < ND) rd[P][ks] = ldrow(rs_d, od[ks]);
      if (ks >= 4 && ks - 4 < NX)                          // rows of stage rel + 1 -> the other buffer
        *reinterpret_cast<float4*>(&sX[Q][(xr + (ks - 4) * PX) * LDX + xc]) = rx[Q][ks - 4];
      if (ks >= 4 && ks - 4 < ND)
        *reinterpret_cast<float4*>(&sD[Q][(dr + (ks - 4) * PD) * LDD + dc]) = rd[Q][ks - 4];
      float an[CW], bn[NW];
      if (ks + 1 < SP / 4) {
        const int pp = (ks + 1) * 4 + g4;
#pragma unroll
        for (int i = 0; i < CW; ++i) an[i] = bx[pp * LDX + (ct0 + i) * 16 + m];
#pragma unroll
        for (int i = 0; i < NW; ++i) bn[i] = bd[pp * LDD + (nt0 + i) * 16 + m];
      }
#pragma unroll
      for (int i = 0; i < CW; ++i)
#pragma unroll
        for (int jn = 0; jn < NW; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[jn], acc[i][jn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < SP / 4) {
#pragma unroll
        for (int i = 0; i < CW; ++i) a[i] = an[i];
#pragma unroll
        for (int i = 0; i < NW; ++i) b[i] = bn[i];
      }
    }
  };
  for (int rel = 0; rel < n_it; rel += 2) {
    stage(std::integral_constant<int, 0>{}, rel);
    stage(std::integral_constant<int, 1>{}, rel + 1);
  }
  float* out = part + ((size_t)s * K + k) * CIN * COUT;
#pragma unroll
  for (int i = 0; i < CW; ++i)
#pragma unroll
    for (int jn = 0; jn < NW; ++jn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = (ct0 + i) * 16 + g4 * 4 + r;
        const int n = (nt0 + jn) * 16 + m;
        out[(size_t)c * COUT + n] = acc[i][jn][r];
      }
}

#undef WP2_LDIDX
#undef WP2_IDX

// dw[k] = sum of the shares of offset k in a fixed order (deterministic). One workgroup never straddles two offsets
// (cin * cout is a multiple of 256 for the supported channel counts).
__global__ __launch_bounds__(256) void k_pairs_reduce(const float* __restrict__ part, const int32_t* __restrict__ counts,
                                                      int K, int G, int smax, size_t per_offset, size_t elems,
                                                      float* __restrict__ dw, int xcd = 0) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // block-uniform offset index -> the share count is computed with scalar loads by every wave (no LDS round trip)
  const int k = (int)(((size_t)blockIdx.x * blockDim.x) / per_offset);
  const int n = pairs_shares(counts, K, k, G, smax, xcd);
  if (i >= elems) return;
  float s = 0.f;                                  // loads in batches of 8, adds in the original order (bit-identical)
  int j = 0;
  for (; j + 8 <= n; j += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(j + u) * elems + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; j < n; ++j) s += part[(size_t)j * elems + i];
  dw[i] = s;
}

// ------------------------------------------------------------------------------------ C entry points ---
extern "C" size_t irx_pairs_workspace_bytes(int n_out, int K) {
  if (n_out <= 0 || K <= 0) return 0;
  return (size_t)K * irx_cdiv(n_out, PB_TILE) * sizeof(int32_t) + 64;
}

extern "C" int irx_pairs_build(const int32_t* nbr, int ld, int n_out, int K, int32_t* in_list, int32_t* out_list,
                               int ldp, int32_t* counts, void* workspace, size_t workspace_bytes, void* stream) {
  IRX_REQUIRE(n_out >= 0 && K >= 1 && K <= 65535 && counts, "irx_pairs_build: bad arguments");
  if (n_out == 0) {
    IRX_CHECK_HIP(hipMemsetAsync(counts, 0, K * sizeof(int32_t), S(stream)), "irx_pairs_build(memset)");
    return IRX_OK;
  }
  IRX_REQUIRE(nbr && in_list && out_list, "irx_pairs_build: null pointer");
  IRX_REQUIRE(ld >= n_out && ldp >= n_out, "irx_pairs_build: ld %d / ldp %d < n_out %d", ld, ldp, n_out);
  if (workspace == nullptr || workspace_bytes < irx_pairs_workspace_bytes(n_out, K)) {
    irx_set_error("irx_pairs_build: workspace %zu < %zu", workspace_bytes, irx_pairs_workspace_bytes(n_out, K));
    return IRX_ERR_WORKSPACE;
  }
  const int ntiles = irx_cdiv(n_out, PB_TILE);
  dim3 grid(ntiles, K);
  k_pairs_count<<<grid, 256, 0, S(stream)>>>(nbr, ld, n_out, (int32_t*)workspace, ntiles);
  IRX_CHECK_LAUNCH("irx_pairs_build(count)");
  k_pairs_write<<<grid, 256, 0, S(stream)>>>(nbr, ld, n_out, (const int32_t*)workspace, ntiles, in_list, out_list,
                                            ldp, counts);
  IRX_CHECK_LAUNCH("irx_pairs_build(write)");
  return IRX_OK;
}

// G = workgroup budget (~1024, fewer for small layers: >= ~8 stages of useful work per workgroup on average);
// smax = shares cap per offset = 3x the average (the centre offset of a 3^3 kernel holds ~2.8x the average pairs).
// ---- several tables in one call --------------------------------------------------------------------------------
// The backward pass of one encoder needs the pair lists of eight tables (four 27-neighbour tables, four child tables):
// built one by one that is 16 launches, ~32 allocations and 8 library calls per encoder and step, all on the host's
// critical path.  Here: two launches for all of them; blockIdx.x runs over the concatenated tiles of every table.
struct IrxPairJobs {
  const int32_t* nbr[IRX_PAIRS_MAX_TABLES];
  int32_t* in_list[IRX_PAIRS_MAX_TABLES];
  int32_t* out_list[IRX_PAIRS_MAX_TABLES];
  int32_t* counts[IRX_PAIRS_MAX_TABLES];
  int32_t* tile_counts[IRX_PAIRS_MAX_TABLES];
  int ld[IRX_PAIRS_MAX_TABLES], n_out[IRX_PAIRS_MAX_TABLES], K[IRX_PAIRS_MAX_TABLES], ldp[IRX_PAIRS_MAX_TABLES];
  int tile_end[IRX_PAIRS_MAX_TABLES];       // running total of tiles up to and including table j
  int n;
};

__device__ __forceinline__ int pair_job_of(const IrxPairJobs& J, int bx, int& tile) {
  int j = 0;
  while (j < J.n - 1 && bx >= J.tile_end[j]) ++j;
  tile = bx - (j ? J.tile_end[j - 1] : 0);
  return j;
}

__global__ __launch_bounds__(256) void k_pairs_count_multi(IrxPairJobs J) {
  __shared__ int s_w[4];
  int tile;
  const int j = pair_job_of(J, blockIdx.x, tile), k = blockIdx.y;
  if (k >= J.K[j]) return;
  const int ntiles = J.tile_end[j] - (j ? J.tile_end[j - 1] : 0);
  const int32_t* nbr = J.nbr[j];
  int cnt = 0;
  for (int it = 0; it < 8; ++it) {
    const int q = tile * PB_TILE + it * 256 + threadIdx.x;
    if (q < J.n_out[j] && nbr[(size_t)k * J.ld[j] + q] >= 0) ++cnt;
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) J.tile_counts[j][k * ntiles + tile] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(256) void k_pairs_write_multi(IrxPairJobs J) {
  __shared__ int s_red[4];
  __shared__ int s_wave[4];
  __shared__ int s_base;
  int tile;
  const int j = pair_job_of(J, blockIdx.x, tile), k = blockIdx.y;
  if (k >= J.K[j]) return;
  const int ntiles = J.tile_end[j] - (j ? J.tile_end[j - 1] : 0);
  const int32_t* nbr = J.nbr[j];
  const int32_t* tile_counts = J.tile_counts[j];
  const int ld = J.ld[j], n_out = J.n_out[j], ldp = J.ldp[j];
  int acc = 0;
  for (int t = threadIdx.x; t < tile; t += 256) acc += tile_counts[k * ntiles + t];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    s_base = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    if (tile == ntiles - 1) J.counts[j][k] = s_base + tile_counts[k * ntiles + tile];
  }
  __syncthreads();
  int running = s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int it = 0; it < 8; ++it) {
    const int q = tile * PB_TILE + it * 256 + threadIdx.x;
    int idx = -1;
    if (q < n_out) idx = nbr[(size_t)k * ld + q];
    const unsigned long long b = __ballot(idx >= 0);
    if (lane == 0) s_wave[wave] = __popcll(b);
    __syncthreads();
    int wbase = 0, total = 0;
    for (int w = 0; w < 4; ++w) {
      const int c = s_wave[w];
      if (w < wave) wbase += c;
      total += c;
    }
    if (idx >= 0) {
      const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      const int pos = running + wbase + __popcll(b & lt);
      J.in_list[j][(size_t)k * ldp + pos] = idx;
      J.out_list[j][(size_t)k * ldp + pos] = q;
    }
    running += total;
    __syncthreads();
  }
}

// tables: n_tables <= IRX_PAIRS_MAX_TABLES; arrays of per-table arguments with irx_pairs_build's meaning. workspace:
// sum over tables of irx_pairs_workspace_bytes(n_out, K).
extern "C" int irx_pairs_build_multi(int n_tables, const int32_t* const* nbr, const int* ld, const int* n_out, const int* K,
                                     int32_t* const* in_list, int32_t* const* out_list, const int* ldp,
                                     int32_t* const* counts, void* workspace, size_t workspace_bytes, void* stream) {
  IRX_REQUIRE(n_tables >= 0 && n_tables <= IRX_PAIRS_MAX_TABLES, "irx_pairs_build_multi: %d tables (max %d)", n_tables,
              IRX_PAIRS_MAX_TABLES);
  if (n_tables == 0) return IRX_OK;
  IRX_REQUIRE(nbr && ld && n_out && K && in_list && out_list && ldp && counts, "irx_pairs_build_multi: null pointer");
  IrxPairJobs J;
  J.n = 0;
  size_t need = 0;
  int tiles = 0, kmax = 0;
  for (int t = 0; t < n_tables; ++t) {
    IRX_REQUIRE(n_out[t] >= 0 && K[t] >= 1 && K[t] <= 27 && counts[t], "irx_pairs_build_multi: bad table %d", t);
    if (n_out[t] == 0) {
      IRX_CHECK_HIP(hipMemsetAsync(counts[t], 0, K[t] * sizeof(int32_t), S(stream)), "irx_pairs_build_multi(memset)");
      continue;
    }
    IRX_REQUIRE(nbr[t] && in_list[t] && out_list[t] && ld[t] >= n_out[t] && ldp[t] >= n_out[t],
                "irx_pairs_build_multi: table %d: null pointer or ld / ldp < n_out", t);
    const int j = J.n++;
    J.nbr[j] = nbr[t]; J.in_list[j] = in_list[t]; J.out_list[j] = out_list[t]; J.counts[j] = counts[t];
    J.ld[j] = ld[t]; J.n_out[j] = n_out[t]; J.K[j] = K[t]; J.ldp[j] = ldp[t];
    J.tile_counts[j] = (int32_t*)((char*)workspace + need);
    need += irx_pairs_workspace_bytes(n_out[t], K[t]);
    tiles += irx_cdiv(n_out[t], PB_TILE);
    J.tile_end[j] = tiles;
    if (K[t] > kmax) kmax = K[t];
  }
  if (J.n == 0) return IRX_OK;
  if (workspace == nullptr || workspace_bytes < need) {
    irx_set_error("irx_pairs_build_multi: workspace %zu < %zu", workspace_bytes, need);
    return IRX_ERR_WORKSPACE;
  }
  dim3 grid(tiles, kmax);
  k_pairs_count_multi<<<grid, 256, 0, S(stream)>>>(J);
  IRX_CHECK_LAUNCH("irx_pairs_build_multi(count)");
  k_pairs_write_multi<<<grid, 256, 0, S(stream)>>>(J);
  IRX_CHECK_LAUNCH("irx_pairs_build_multi(write)");
  return IRX_OK;
}

// ---- bf16 weight-gradient, third generation (round 4): bf16 rows stay bf16, transposing LDS reads, 32x32x16 MFMA ---------
// k_wgrad_pairs in bf16 storage mode is the fp32 design fed narrower rows: it widens them to fp32 when it writes the LDS
// tiles, reads every fragment element with its own ds_read_b32, packs pairs of them back to bf16 and runs the half-rate
// 16x16x16 MFMA — 40 LDS reads and 20 packs per 16 MFMAs, two barriers and a full vector-memory drain per 64-pair stage
// (150 us in the kernel for the 0.79 M pairs of the largest 128-channel level: 7 % of the matrix-core peak).  Both operands of the weight
// gradient have the PAIRS as their reduction dimension, i.e. a lane's 8 reduction elements come from 8 different rows — the
// one access pattern gfx950 has an instruction for:
//   * rows are gathered with raw buffer loads (16 B per lane, a missing pair is an out-of-range offset = zeros) and written
//     to LDS as they are: bf16, [pair][channel], 64-byte windows XOR-swizzled by the pair's low bits instead of padded;
//   * ds_read_b64_tr_b16 hands a lane 4 consecutive pairs of ITS channel (a 16-lane group reads a [4 pairs][16 channels]
//     block); two of them are one operand of v_mfma_f32_32x32x16_bf16 — 1 LDS read per MFMA at 128 x 128 (wave tile 64 x 64),
//     no packs, no conversion; the swizzle makes the four pair rows of a read land in four different 16-bank windows;
//   * the row buffers are double-buffered: ONE barrier per stage; TWO register sets of rows in flight — stage t+3's rows are
//     requested from inside stage t's MFMA chain into the set stage t+1's rows have just left (written to LDS behind the MFMAs
//     of the stage's first step); the index lists are fetched four stages per load, one chunk ahead, into an 8-slot LDS ring:
//     every load has at least two whole iterations to land and none sits behind a branch (exact wait counts);
//   * fp32 accumulators in registers for the whole work unit, deterministic per-unit partials + k_pairs_reduce as
//     k_wgrad_pairs; on levels with enough work the units are XCD segments (pairs_unit): the 27 offsets of an eighth of the
//     level's rows run on ONE XCD at the same time, so a gathered row is an L2 hit for 26 of them (the (share, offset) grid
//     missed L2 on 82 % of its requests: 445 MB from the fabric for 41 MB of rows).
// Measured (rocprofv3 kernel trace, B = 16 scene pyramid): 150.5 -> 46.8 us at 128 x 128 on 81 k rows, 161.6 -> 49.8 us at
// 64 x 64 on 259 k rows; ablations and counters in profiles/r04_wgrad3_*.txt, DESIGN.md section 4.
typedef __bf16 w3_bf16x8 __attribute__((ext_vector_type(8)));
typedef short w3_s16x4 __attribute__((ext_vector_type(4)));
typedef short w3_s16x8 __attribute__((ext_vector_type(8)));
typedef float w3_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned w3_u32x4 __attribute__((ext_vector_type(4)));
#define W3_OOB 0x80000000u
// dev ablation (timing only, results wrong): 1 = no row loads, 2 = no MFMA, 4 = no fragment reads, 8 = no LDS row writes,
// 16 = no partial-sum stores
#ifndef IRX_W3_ABL
#define IRX_W3_ABL 0
#endif

// 64-byte window swizzle of a row image with RB bytes per row: the four rows of an aligned group of four pairs must occupy
// four different 16-bank windows of the 64-bank LDS for the transposing read to be conflict-free
template <int RB>
__device__ __forceinline__ int w3_swz(int row) {
  constexpr int WPR = RB / 64;                       // windows per row
  return (((row & 3) * WPR) >> 2) & (WPR - 1);       // 4 windows: row & 3; 2 windows: bit 1 of the row; 1 window: 0
}

__device__ __forceinline__ w3_bf16x8 w3_frag(const unsigned char* lds, int byte_off_lo, int byte_off_hi) {
  typedef __attribute__((address_space(3))) w3_s16x4* lp;
  const w3_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds + byte_off_lo));
  const w3_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds + byte_off_hi));
  const w3_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(w3_bf16x8, v);
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void k_wgrad3(const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy,
                                                   const int32_t* __restrict__ in_list, const int32_t* __restrict__ out_list,
                                                   int ldp, const int32_t* __restrict__ counts, int K, int G, int smax,
                                                   float* __restrict__ part, int ldx, int xcd) {
  static_assert((CIN == 64 || CIN == 128) && (COUT == 64 || COUT == 128), "wave tiles are (CIN/2) x (COUT/2)");
  constexpr int RBX = CIN * 2, RBD = COUT * 2;           // bytes of a row image
  constexpr int LPX = RBX / 16, LPD = RBD / 16;          // lanes (16 B) per row
  constexpr int PX = 256 / LPX, PD = 256 / LPD;          // rows per staging pass
  constexpr int NX = 64 / PX, ND = 64 / PD;              // passes per 64-pair stage
  constexpr int CB = CIN / 64, NB = COUT / 64;           // 32 x 32 blocks per wave, each way
  constexpr int BX = 64 * RBX, BD = 64 * RBD;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BX + BD)];
  __shared__ __attribute__((aligned(16))) unsigned sIdx[8][4][64];   // byte offsets of stage u's rows in slot u & 7: [0] x, [1] dy
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // xcd: 1-D grid, workgroup b belongs to XCD b % 8 (the dispatcher deals consecutive workgroups round-robin over the XCDs) and
  // takes the (b / 8)-th unit of that XCD's list — offset by offset, the m_k shares of list k's segment b % 8 (pairs_unit).
  // The units of one XCD therefore walk the same eighth of the level's rows, all at the same time (the grid fits the chip at
  // once): a row fetched for one offset is an L2 hit for the other 26 instead of 27 trips to the fabric.
  int s, k, nsplit;
  if (!pairs_unit(counts, K, G, smax, xcd, s, k, nsplit)) return;      // block-uniform
  const int cnt = counts[k];
  const int nst = (cnt + 63) / 64;
  const int st0 = (int)(((long long)nst * s) / nsplit), st1 = (int)(((long long)nst * (s + 1)) / nsplit);
  const int32_t* il = in_list + (size_t)k * ldp;
  const int32_t* ol = out_list + (size_t)k * ldp;
  const unsigned rbx_g = (unsigned)ldx * 2u;             // bytes of a row of x in HBM

  w3_f32x16 acc[CB][NB];
#pragma unroll
  for (int a = 0; a < CB; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  if (st0 < st1) {
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, 0x7FFFFFF0, 0x00020000);
    // staging role of this thread: row xr (+ i * PX) / 16-byte piece xc of the x image, the same for dy
    const int xr = tid / LPX, xc = tid % LPX;
    const int dr = tid / LPD, dc = tid % LPD;
    const int xw = xr * RBX + ((((xc >> 2) ^ w3_swz<RBX>(xr))) << 6) + (xc & 3) * 16;      // (PX, PD are multiples of 4: the
    const int dw_ = BX + dr * RBD + ((((dc >> 2) ^ w3_swz<RBD>(dr))) << 6) + (dc & 3) * 16; //  swizzle is the same every pass)
    // fragment role: lane l of a 16-lane group reads pair (l >> 2) & 3 of a [4 pairs][16 channels] block, 8 bytes at channel
    // 4 (l & 3); lanes 16..31 take the block's upper 16 channels, lanes 32..63 the pairs 8..15 of the 16-pair step
    const int li = lane & 15, g16 = (lane >> 4) & 1, kg = lane >> 5, pr = li >> 2;
    const int wc = wave >> 1, wn = wave & 1;
    int aoff[CB], boff[NB];
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int cbyte = ((wc * CB + i) * 32 + 16 * g16 + 4 * (li & 3)) * 2;
      aoff[i] = (8 * kg + pr) * RBX + (((cbyte >> 6) ^ w3_swz<RBX>(pr)) << 6) + (cbyte & 63);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int cbyte = ((wn * NB + j) * 32 + 16 * g16 + 4 * (li & 3)) * 2;
      boff[j] = BX + (8 * kg + pr) * RBD + (((cbyte >> 6) ^ w3_swz<RBD>(pr)) << 6) + (cbyte & 63);
    }
    // index role: wave 0 fetches input rows, wave 1 output rows, FOUR stages (256 indices, 16 B per lane) per load; waves 2, 3
    // run the same instructions on an out-of-range offset (no memory access): no load of the loop sits behind a branch and
    // every wait count is exact.  Lane l holds indices 4 (l & 15) .. + 3 of the chunk's stage l >> 4.
    const __amdgpu_buffer_rsrc_t rs_i =
        __builtin_amdgcn_make_buffer_rsrc((void*)(wave == 0 ? il : ol), 0, (unsigned)ldp * 4u, 0x00020000);
    const unsigned rb_mine = wave == 0 ? rbx_g : (unsigned)RBD;
    auto fetch_chunk = [&](int u0) __attribute__((always_inline)) -> w3_u32x4 {      // stages u0 .. u0 + 3
      const unsigned o = wave < 2 ? (unsigned)(u0 * 64 + lane * 4) * 4u : W3_OOB;
      return __builtin_amdgcn_raw_buffer_load_b128(rs_i, o, 0, 0);
    };
    // ring of 8 stage slots: [slot][list: 0 x, 1 dy, (2, 3 unused)][64] byte offsets, W3_OOB = no such pair
    auto store_chunk = [&](w3_u32x4 raw, int u0) __attribute__((always_inline)) {
      const int u = u0 + (lane >> 4);
      const int p = u * 64 + (lane & 15) * 4;
      w3_u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (u < st1 && p + e < cnt) ? raw[e] * rb_mine : W3_OOB;
      *reinterpret_cast<w3_u32x4*>(&sIdx[u & 7][wave][(lane & 15) * 4]) = o;
    };
    w3_u32x4 rx[2][NX], rd[2][ND];                        // rows of stage u in set u & 1 (relative to st0: chunks start even)
    auto request_x = [&](int set, int u, int i) __attribute__((always_inline)) {
      const unsigned o = (IRX_W3_ABL & 1) ? W3_OOB : sIdx[u & 7][0][xr + i * PX];
      rx[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, o + (unsigned)xc * 16u, 0, 0);
    };
    auto request_d = [&](int set, int u, int i) __attribute__((always_inline)) {
      const unsigned o = (IRX_W3_ABL & 1) ? W3_OOB : sIdx[u & 7][1][dr + i * PD];
      rd[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, o + (unsigned)dc * 16u, 0, 0);
    };
    auto request_quarter = [&](int set, int u, int q) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < NX; ++i)
        if ((i * 4) / NX == q) request_x(set, u, i);
#pragma unroll
      for (int i = 0; i < ND; ++i)
        if ((i * 4) / ND == q) request_d(set, u, i);
    };
    auto write_rows = [&](int set, int buf) __attribute__((always_inline)) {
      if (IRX_W3_ABL & 8) return;
      unsigned char* b = smem + buf * (BX + BD);
#pragma unroll
      for (int i = 0; i < NX; ++i) *reinterpret_cast<w3_u32x4*>(b + xw + i * PX * RBX) = rx[set][i];
#pragma unroll
      for (int i = 0; i < ND; ++i) *reinterpret_cast<w3_u32x4*>(b + dw_ + i * PD * RBD) = rd[set][i];
    };

    // Stage u (absolute) lives in LDS buffer (u - st0) & 1 and its rows travel in register set (u - st0) & 1; its indices
    // sit in ring slot u & 7.  Timeline of iteration t (MFMAs of stage t):
    //   rows(t+1) [requested at t-2]  registers -> the other buffer          rows(t+3) requested into the set just freed
    //   first iteration of a chunk of 4: indices of stages c+4..c+7 [requested at c-4] registers -> ring, c+8..c+11 requested
    // so a row load has two whole iterations and an index load four to land before anything waits for it.
    // ---- prologue: the first 8 stages' indices in one round trip, the first three stages' rows in the next ----
    // (first iteration of chunk c: indices c+4..c+7 registers -> ring, c+8..c+11 requested)
    w3_u32x4 ic;                                          // indices of stages c+4 .. c+7 while chunk c runs
    {
      const w3_u32x4 ic0 = fetch_chunk(st0);
      ic = fetch_chunk(st0 + 4);
      store_chunk(ic0, st0);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) request_quarter(0, st0, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) request_quarter(1, st0 + 1, q);
    write_rows(0, 0);                                     // (waits for stage st0's rows only)
#pragma unroll
    for (int q = 0; q < 4; ++q) request_quarter(0, st0 + 2, q);
    __syncthreads();
    for (int c = st0; c < st1; c += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = c + j;
        const unsigned char* b = smem + (j & 1) * (BX + BD);
        w3_bf16x8 fa[2][CB], fb[2][NB];
        auto frags = [&](int ks) __attribute__((always_inline)) {
          if (IRX_W3_ABL & 4) return;
#pragma unroll
          for (int i = 0; i < CB; ++i) fa[ks & 1][i] = w3_frag(b, aoff[i] + (16 * ks) * RBX, aoff[i] + (16 * ks + 4) * RBX);
#pragma unroll
          for (int jn = 0; jn < NB; ++jn) fb[ks & 1][jn] = w3_frag(b, boff[jn] + (16 * ks) * RBD, boff[jn] + (16 * ks + 4) * RBD);
        };
        auto mfmas = [&](int ks) __attribute__((always_inline)) {
          if (IRX_W3_ABL & 2) return;
#pragma unroll
          for (int i = 0; i < CB; ++i)
#pragma unroll
            for (int jn = 0; jn < NB; ++jn)
              acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][i], fb[ks & 1][jn], acc[i][jn], 0, 0, 0);
        };
        // (stages past the share's end — a chunk is always run to its 4th stage — have out-of-range rows: zeros, no memory
        //  access; their MFMAs add nothing)
        frags(0);
        frags(1);
        mfmas(0);
        __builtin_amdgcn_sched_barrier(0);
        write_rows((j + 1) & 1, (j + 1) & 1);
        if (j == 0) {
          store_chunk(ic, c + 4);
          ic = fetch_chunk(c + 8);
        }
        request_quarter((j + 1) & 1, t + 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        frags(2);
        mfmas(1);
        request_quarter((j + 1) & 1, t + 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        frags(3);
        mfmas(2);
        request_quarter((j + 1) & 1, t + 3, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(3);
        request_quarter((j + 1) & 1, t + 3, 3);
        __syncthreads();
      }
    }
  }
  float* out = part + ((size_t)s * K + k) * CIN * COUT;
  {
    const int kg = lane >> 5, wc = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int i = 0; i < CB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (wc * CB + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
          const int n = (wn * NB + j) * 32 + (lane & 31);
          if (!(IRX_W3_ABL & 16) || acc[i][j][r] == 12345.f) out[(size_t)c * COUT + n] = acc[i][j][r];
        }
  }
}

static int pairs_budget(int n_out, int K) {
  static const int target = getenv("IRX_PAIRS_BUDGET") ? atoi(getenv("IRX_PAIRS_BUDGET")) : 1024;   // dev A/B knob
  int s = irx_cdiv(target, K);
  const int max_s = irx_cdiv(n_out, 512);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return s * K;
}
static int pairs_smax(int n_out, int K) { return 3 * (pairs_budget(n_out, K) / K); }

extern "C" size_t irx_spconv_wgrad_pairs_workspace_bytes(int n_out, int K, int cin, int cout) {
  if (n_out <= 0 || K <= 0 || cin <= 0 || cout <= 0) return 0;
  return (size_t)pairs_smax(n_out, K) * K * cin * cout * sizeof(float);
}

// dev A/B: IRX_WGRAD_V1=1 runs the fp32 weight-gradient on the first-generation kernel (bit-identical results)
static bool wp_v1() { return irx_knob(IRX_KNOB_WGRAD_V1) != 0; }   // (the parity test flips it with irx_debug_set_knob)

template <int CIN>
static void launch_wp(int cout, dim3 grid, hipStream_t st, const float* x, const float* dy, const int32_t* il,
                      const int32_t* ol, int ldp, const int32_t* counts, int K, int G, int smax, float* part, bool bf_rows,
                      bool gen2, int xcd) {
  irx_bracket_begin(st);
  if constexpr (CIN >= 64) {
    if (irx_conv_bf16() && bf_rows && gen2 && cout >= 64 && irx_knob(IRX_KNOB_WGRAD3) != 0) {
      const unsigned short* xb = reinterpret_cast<const unsigned short*>(x);
      const unsigned short* db = reinterpret_cast<const unsigned short*>(dy);
      const dim3 g3 = xcd ? dim3(8 * (G / 8 + K + 1)) : grid;
      if (cout == 128) k_wgrad3<CIN, 128><<<g3, 256, 0, st>>>(xb, db, il, ol, ldp, counts, K, G, smax, part, CIN, xcd);
      else k_wgrad3<CIN, 64><<<g3, 256, 0, st>>>(xb, db, il, ol, ldp, counts, K, G, smax, part, CIN, xcd);
      irx_bracket_end(st);
      return;
    }
  }
  if (irx_conv_bf16() && bf_rows) {
    if (cout == 128) k_wgrad_pairs<CIN, 128, true, true><<<grid, 256, 0, st>>>(x, dy, il, ol, ldp, counts, K, G, smax, part);
    else if (cout == 64) k_wgrad_pairs<CIN, 64, true, true><<<grid, 256, 0, st>>>(x, dy, il, ol, ldp, counts, K, G, smax, part);
    else k_wgrad_pairs<CIN, 32, true, true><<<grid, 256, 0, st>>>(x, dy, il, ol, ldp, counts, K, G, smax, part);
  } else if (irx_conv_bf16()) {
    if (cout == 128) k_wgrad_pairs<CIN, 128, true, false><<<grid, 256, 0, st>>>(x, dy, il, ol, ldp, counts, K, G, smax, part);
    else if (cout == 64) k_wgrad_pairs<CIN, 64, true, false><<<grid, 256, 0, st>>>(x, dy, il, ol, ldp, counts, K, G, smax, part);
    else k_wgrad_pairs<CIN, 32, true, false><<<grid, 256, 0, st>>>(x, dy, il, ol, ldp, counts, K, G, smax, part);
  } else if (!gen2 || wp_v1()) {
    const dim3 g1 = xcd ? dim3(8 * (G / 8 + K + 1)) : grid;
    if (cout == 128) k_wgrad_pairs<CIN, 128, false, false><<<g1, 256, 0, st>>>(x, dy, il, ol, ldp, counts, K, G, smax, part, CIN, xcd);
    else if (cout == 64) k_wgrad_pairs<CIN, 64, false, false><<<g1, 256, 0, st>>>(x, dy, il, ol, ldp, counts, K, G, smax, part, CIN, xcd);
    else k_wgrad_pairs<CIN, 32, false, false><<<g1, 256, 0, st>>>(x, dy, il, ol, ldp, counts, K, G, smax, part, CIN, xcd);
  } else {
    const dim3 g2 = xcd ? dim3(8 * (G / 8 + K + 1)) : grid;
    if (cout == 128) k_wgrad_pairs2<CIN, 128><<<g2, 256, 0, st>>>(x, dy, il, ol, ldp, counts, K, G, smax, part, CIN, xcd);
    else if (cout == 64) k_wgrad_pairs2<CIN, 64><<<g2, 256, 0, st>>>(x, dy, il, ol, ldp, counts, K, G, smax, part, CIN, xcd);
    else k_wgrad_pairs2<CIN, 32><<<g2, 256, 0, st>>>(x, dy, il, ol, ldp, counts, K, G, smax, part, CIN, xcd);
  }
  irx_bracket_end(st);
}

// The wide stem's leading 128 input channels through the pair lists (irx_spconv_wgrad_impl): x is fp32 with row stride
// ldx (4-byte aligned rows), dy [n_out][cout = 32] fp32 or (dy_bf, bf16 storage mode) bf16; sum = [K][128][cout].
size_t irx_wgrad_pairs_wide_workspace_bytes(int n_out, int K, int cout) {
  return irx_spconv_wgrad_pairs_workspace_bytes(n_out, K, 128, cout);
}
int irx_wgrad_pairs_wide_launch(const float* x, int ldx, const float* dy, const int32_t* in_list, const int32_t* out_list,
                                int ldp, const int32_t* counts, int n_out, int K, int cout, float* part, float* sum,
                                hipStream_t st, int dy_bf) {
  IRX_REQUIRE(cout == 32, "irx_spconv_wgrad(wide stem, pairs): cout = %d", cout);
  const int G = pairs_budget(n_out, K), smax = pairs_smax(n_out, K);
  dim3 grid(smax, K);
  irx_bracket_begin(st);
  if (irx_conv_bf16() && dy_bf)
    k_wgrad_pairs<128, 32, true, false, true><<<grid, 256, 0, st>>>(x, dy, in_list, out_list, ldp, counts, K, G, smax, part, ldx);
  else if (irx_conv_bf16())
    k_wgrad_pairs<128, 32, true, false, false><<<grid, 256, 0, st>>>(x, dy, in_list, out_list, ldp, counts, K, G, smax, part, ldx);
  else if (wp_v1() || (size_t)n_out * ldx * sizeof(float) >= ((size_t)1 << 31))       // (a stride-1 stem: rows of x = n_out)
    k_wgrad_pairs<128, 32, false, false, false><<<grid, 256, 0, st>>>(x, dy, in_list, out_list, ldp, counts, K, G, smax, part, ldx);
  else
    k_wgrad_pairs2<128, 32><<<grid, 256, 0, st>>>(x, dy, in_list, out_list, ldp, counts, K, G, smax, part, ldx);
  irx_bracket_end(st);
  IRX_CHECK_LAUNCH("irx_spconv_wgrad(wide stem, pairs)");
  const size_t elems = (size_t)K * 128 * cout;
  k_pairs_reduce<<<irx_cdiv((long long)elems, 256), 256, 0, st>>>(part, counts, K, G, smax, (size_t)128 * cout, elems, sum);
  IRX_CHECK_LAUNCH("irx_spconv_wgrad(wide stem, pairs reduce)");
  return IRX_OK;
}

extern "C" int irx_debug_occupancy_wp(int which) {
  int n = -1;
  hipError_t e = hipSuccess;
  switch (which) {
    case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_wgrad_pairs2<128, 128>, 256, 0); break;
    case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_wgrad_pairs2<64, 64>, 256, 0); break;
    case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_wgrad_pairs<128, 128, true, true, true>, 256, 0); break;
    default: return -2;
  }
  return e == hipSuccess ? n : -1;
}

extern "C" int irx_spconv_wgrad_pairs(const float* x, const float* dy, const int32_t* in_list,
                                      const int32_t* out_list, int ldp, const int32_t* counts, int n_in, int n_out, int K,
                                      int cin, int cout, float* dw, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  IRX_REQUIRE(n_in >= 0, "irx_spconv_wgrad_pairs: n_in < 0");
  return irx_spconv_wgrad_pairs_impl(x, dy, in_list, out_list, ldp, counts, n_out, K, cin, cout, dw, workspace,
                                     workspace_bytes, stream, 0, n_in);
}

extern "C" int irx_spconv_wgrad_pairs_t(const void* x, const void* dy, const int32_t* in_list, const int32_t* out_list,
                                        int ldp, const int32_t* counts, int n_in, int n_out, int K, int cin, int cout,
                                        float* dw, int rows_bf, void* workspace, size_t workspace_bytes, void* stream) {
  IRX_REQUIRE(n_in >= 0, "irx_spconv_wgrad_pairs_t: n_in < 0");
  return irx_spconv_wgrad_pairs_impl((const float*)x, (const float*)dy, in_list, out_list, ldp, counts, n_out, K, cin, cout,
                                     dw, workspace, workspace_bytes, stream, rows_bf != 0, n_in);
}

// bf_rows != 0 (executor, bf16 storage mode): x and dy are bf16 tensors
int irx_spconv_wgrad_pairs_impl(const float* x, const float* dy, const int32_t* in_list, const int32_t* out_list, int ldp,
                                const int32_t* counts, int n_out, int K, int cin, int cout, float* dw, void* workspace,
                                size_t workspace_bytes, void* stream, int bf_rows, int n_in) {
  IRX_REQUIRE(!bf_rows || irx_conv_bf16(), "irx_spconv_wgrad_pairs: bf16 rows need the bf16 compute mode");
  IRX_REQUIRE(n_out >= 0 && K >= 1 && dw, "irx_spconv_wgrad_pairs: bad arguments");
  IRX_REQUIRE(irx_spconv2_supported(cin, cout), "irx_spconv_wgrad_pairs: channels (%d, %d) unsupported (32/64/128)", cin, cout);
  const size_t elems = (size_t)K * cin * cout;
  if (n_out == 0) {
    IRX_CHECK_HIP(hipMemsetAsync(dw, 0, elems * sizeof(float), S(stream)), "irx_spconv_wgrad_pairs(memset)");
    return IRX_OK;
  }
  IRX_REQUIRE(x && dy && in_list && out_list && counts, "irx_spconv_wgrad_pairs: null pointer");
  IRX_REQUIRE(((((uintptr_t)x | (uintptr_t)dy)) & 15) == 0, "irx_spconv_wgrad_pairs: x / dy must be 16-byte aligned");
  int G = pairs_budget(n_out, K);
  const int smax = pairs_smax(n_out, K);
  const size_t need = irx_spconv_wgrad_pairs_workspace_bytes(n_out, K, cin, cout);
  if (workspace == nullptr || workspace_bytes < need) {
    irx_set_error("irx_spconv_wgrad_pairs: workspace %zu < %zu", workspace_bytes, need);
    return IRX_ERR_WORKSPACE;
  }
  float* part = (float*)workspace;
  dim3 grid(smax, K);
  const bool gen2 = n_in > 0 && (size_t)n_in * cin * sizeof(float) < ((size_t)1 << 31) &&
                    (size_t)n_out * cout * sizeof(float) < ((size_t)1 << 31);
  // k_wgrad3 on a level with enough work: XCD-segment units (pairs_shares) — few enough that the whole grid is resident at
  // once (2 workgroups per CU) and that the fp32 partials, 4 * cin * cout bytes per unit each way, stay a small part of the job
  int xcd = 0;
  if (irx_conv_bf16() && bf_rows && gen2 && cin >= 64 && cout >= 64 && irx_knob(IRX_KNOB_WGRAD3) != 0 && smax >= 8 &&
      (long)n_out * K >= irx_knob(IRX_KNOB_WGRAD3_XCD_MIN)) {
    const long units = irx_knob(IRX_KNOB_WGRAD3_UNITS);
    if (units >= 8) { xcd = 1; G = (int)units; }
  } else if (!irx_conv_bf16() && smax >= 8 && irx_knob(IRX_KNOB_WGRAD_XCD_F32) >= 8 &&
             (long)n_out * K >= irx_knob(IRX_KNOB_WGRAD3_XCD_MIN)) {
    xcd = 1;                               // the fp32 kernels on the same work-unit mapping (dev knob: units, 0 = off)
    G = (int)irx_knob(IRX_KNOB_WGRAD_XCD_F32);
  }
  if (cin == 128) launch_wp<128>(cout, grid, S(stream), x, dy, in_list, out_list, ldp, counts, K, G, smax, part, bf_rows != 0, gen2, xcd);
  else if (cin == 64) launch_wp<64>(cout, grid, S(stream), x, dy, in_list, out_list, ldp, counts, K, G, smax, part, bf_rows != 0, gen2, xcd);
  else launch_wp<32>(cout, grid, S(stream), x, dy, in_list, out_list, ldp, counts, K, G, smax, part, bf_rows != 0, gen2, xcd);
  IRX_CHECK_LAUNCH("irx_spconv_wgrad_pairs");
  k_pairs_reduce<<<irx_cdiv((long long)elems, 256), 256, 0, S(stream)>>>(part, counts, K, G, smax, (size_t)cin * cout, elems,
                                                                       dw, xcd);
  IRX_CHECK_LAUNCH("irx_spconv_wgrad_pairs(reduce)");
  return IRX_OK;
}
