// irx_spconv2.hip — second-generation sparse-conv kernels for the channel counts of the encoder
// (Cin, Cout in {32, 64, 128}); irx_spconv.hip keeps the generic fallbacks (odd channel counts).
//
// Forward / data-gradient  k_spconv2<CIN, COUT> (details at the kernel):
//   * workgroup = 64 consecutive (Morton-ordered) output rows, 4 waves; the fp32 output tile lives in LDS;
//   * setup: the tile's table columns are COMPACTED (lane == row, ballot + prefix popcount) into shared LDS pair
//     lists, so the MFMA only ever sees ceil(v/16) dense 16-pair groups instead of every row of the tile (executed
//     rows / useful pairs drop from 1.6-2.6x to ~1.2x on ScanNet-like surfaces);
//   * weight-stationary: each wave owns a slice of output channels and keeps W[k][:, slice] in VGPRs (16-byte
//     loads from a fragment-major weight image), so weights never pass through LDS and all four waves share one
//     gathered A tile; the next item's weights and rows are requested from inside the current item's MFMA chain;
//   * A rows are gathered with 16 B/lane coalesced loads (whole 128-512 B rows) into LDS, read back as
//     ds_read_b128 fragments (the MFMA k index is permuted so a lane's 4 consecutive floats feed 4 MFMAs);
//   * v_mfma_f32_16x16x4_f32 (exact fp32), results added into the LDS output tile (each wave owns its
//     channel slice -> no atomics), tile written once with 16 B/lane stores: deterministic.
// Weight-gradient  k_spconv2_wgrad<CIN, COUT>: same compaction; gathered x rows and dy rows of the valid
//   pairs are the MFMA reduction dimension; per (row split, offset) partial sums, deterministic reduction.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "irx_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Compute dtype of the fast sparse-conv kernels (irx_set_compute_dtype, include/irx.h): 0 = exact fp32 MFMA
// (v_mfma_f32_16x16x4_f32), 1 = bf16 operands (round-to-nearest-even of x and w at use) with fp32 accumulation
// (v_mfma_f32_16x16x16_bf16), tensors in HBM fp32; 2 = as 1 plus bf16 STORAGE of every activation / gradient tensor
// inside the encoder executor (irx_encoder.hip): the conv gathers 8-byte bf16 quads straight into the LDS A tile and
// its epilogue rounds the fp32 tile on the way out; statistics, accumulation and parameter gradients stay fp32.
static int g_irx_conv_bf16 = 0;
// The mode a launch uses: the calling thread's override when one is in force (IrxModeScope: the encoder executor pins the
// mode recorded in its descriptor table, so a pass issued by a library thread — or a backward whose forward ran under another
// setting — never reads a process-wide variable that the application may be changing), else the process-wide setting.
static thread_local int t_irx_mode = -1;
static inline int irx_mode_now() { return t_irx_mode >= 0 ? t_irx_mode : g_irx_conv_bf16; }
IrxModeScope::IrxModeScope(int mode) : prev(t_irx_mode) { t_irx_mode = mode; }
IrxModeScope::~IrxModeScope() { t_irx_mode = prev; }
extern "C" int irx_set_compute_dtype(int mode) {
  IRX_REQUIRE(mode >= 0 && mode <= 2, "irx_set_compute_dtype: %d is not 0 (fp32), 1 (bf16 operands) or 2 (bf16 storage)", mode);
  g_irx_conv_bf16 = mode;
  return IRX_OK;
}
extern "C" int irx_get_compute_dtype(void) { return g_irx_conv_bf16; }
bool irx_conv_bf16() { return irx_mode_now() != 0; }
bool irx_conv_bf16_storage() { return irx_mode_now() == 2; }

#define S2_TM 64
#ifndef IRX_S2_LDA_PAD
#define IRX_S2_LDA_PAD 8      // dev A/B: -DIRX_S2_LDA_PAD=4 is the round-1/2 layout (two-way conflicts on the A-fragment reads)
#endif

// Dev-only per-phase cycle attribution of k_spconv2 (tools/conv_phase_prof.py builds a -DIRX_S2_PROF variant).
#ifdef IRX_S2_PROF
__device__ unsigned long long g_s2_prof[16];
extern "C" int irx_debug_s2_prof(unsigned long long* out, int reset) {
  if (reset) {
    unsigned long long z[16] = {0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_s2_prof), z, sizeof(z));
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s2_prof), 16 * 8);
}
#define S2_TICK(i) do { const long long t_ = clock64(); prof[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define S2_TICK(i) do { } while (0)
#endif

// Dev-only ablation of k_spconv2 (tools/micro/run_abl.sh builds -DIRX_S2_ABL=<mask> variants; results are WRONG, only the
// timing is of interest): 1 = every weight load hits the same 1 KiB (no L2->L1 weight stream), 2 = every gathered row
// is row 0, 4 = no MFMA, 8 = no read-modify-write of the LDS output tile, 16 = every offset reads W[0] (weights certainly L2-resident),
// 64 = every wave reads slice 0 of W[0] (16 KiB: distinct load instructions, L1-resident), 32 = per-tile offset rotation,
// 128 = no per-item barriers, 256 = no A-tile writes, 512 = no group work at all, 2048 = no tile store.
#ifndef IRX_S2_ABL
#define IRX_S2_ABL 0
#endif
#ifndef IRX_S2_SPLITPF
#define IRX_S2_SPLITPF 1
#endif

struct PairList {
  int in_of_pair;   // lane p holds the input row of pair p (p < v)
  int row_of_pair;  // lane p holds the tile-local output row of pair p
  int v;            // number of valid pairs (wave-uniform)
};

// lane == tile row; `my` = table entry (input row or -1). Full permutation: valid lanes go to their rank,
// invalid lanes fill the tail, so every destination has exactly one writer.
__device__ static inline PairList compact_pairs(int my, int lane) {
  const unsigned long long valid = __ballot(my >= 0);
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int v = __popcll(valid);
  const int dst = (my >= 0) ? __popcll(valid & lt) : v + __popcll(~valid & lt);
  PairList p;
  p.in_of_pair = __builtin_amdgcn_ds_permute(dst << 2, my);
  p.row_of_pair = __builtin_amdgcn_ds_permute(dst << 2, lane);
  p.v = v;
  return p;
}

// One 16-pair group of the current item for this wave's channel slice: A fragments from the LDS tile, weights from
// VGPRs, result added into the LDS output tile.  PREFETCH = the global loads of a LATER item (weight slice + gathered
// rows) are issued from inside the MFMA chain, a few per MFMA block: the vector-memory pipe (~50 B/clk/CU) needs
// ~0.4 us per workgroup-item just to ACCEPT the 24 KiB a wave requests, and a wave that issues them back to back sits
// in that queue instead of feeding the MFMA pipe (measured: 34 % of all wave cycles).  The prefetch loads are
// UNCONDITIONAL (padded pairs re-read row `nrow` = a valid row; their results go to the dump row), so every chain
// issues exactly NJ * (NT + 1) loads and the consumer can wait with an exact s_waitcnt vmcnt(n).
// Weights: fp32 mode  WT = float4, one per fragment f = j * NT + t (4 consecutive reduction channels of one column);
//          bf16 mode  WT = uint4 = fragments 2p (.xy) and 2p + 1 (.zw) as 4 bf16 each, WN = NJ * NT / 2 per item.
// RT: one lane's share of a gathered row — float4 (fp32 storage) or uint2 = 4 bf16 (bf16 storage, ST)
template <bool ST>
__device__ __forceinline__ typename std::conditional<ST, uint2, float4>::type s2_ldrow(const float* __restrict__ x, size_t elem) {
  if constexpr (ST) return *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(x) + elem);
  else return *reinterpret_cast<const float4*>(x + elem);
}

template <int CIN, int COUT, int NJ, int NT, int LDA, int LDO, int PREFETCH, bool BF, bool ST, typename WT, int WN,
          typename RT>
__device__ __forceinline__ void s2_group(const float* __restrict__ sA, float* __restrict__ sOut,
                                         const unsigned char* __restrict__ lrow, int g, int vs, int m, int g4,
                                         int n_base, const WT (&wc)[WN], WT (&wx)[WN],
                                         RT (&sx)[NJ], const int (&nrow)[NJ], const float* __restrict__ x, int c4,
                                         const WT* __restrict__ wnk, int ldx) {
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // A tile: fp32 rows (stride LDA floats); bf16 mode: rows already rounded to bf16 by the writer (stride LDAB halves),
  // so a fragment is one ds_read_b64 and no convert sits in the MFMA chain
  using AT = typename std::conditional<BF, uint2, float4>::type;
  constexpr int LDAB = CIN + 8;
  const AT* pa = BF ? reinterpret_cast<const AT*>(reinterpret_cast<const unsigned short*>(sA) + (16 * g + m) * LDAB + 4 * g4)
                    : reinterpret_cast<const AT*>(&sA[(16 * g + m) * LDA + 4 * g4]);
  int orow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) orow[r] = lrow[(16 * g + 4 * g4 + r) & 63];
  AT a_nxt = pa[0];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const AT a4 = a_nxt;                           // fragment j was requested one step ago
    if (j + 1 < NJ) a_nxt = pa[4 * (j + 1)];       // 16 channels further: 4 elements of either type
    // PREFETCH: 0 = none, 1 = the whole chain, 2 / 3 = its first / second half (an item with two or more groups spreads
    // its requests over two MFMA chains: half the burst the L1 has to take)
    if (PREFETCH == 1 || (PREFETCH == 2 && j < NJ / 2) || (PREFETCH == 3 && j >= NJ / 2)) {
      // this step's share of the item's weight loads
#pragma unroll
      for (int i = (BF ? (j * NT) / 2 : j * NT); i < (BF ? ((j + 1) * NT) / 2 : (j + 1) * NT); ++i)
        wx[i] = wnk[(IRX_S2_ABL & 1) ? 0 : (size_t)i * 64];
      sx[j] = s2_ldrow<ST>(x, (size_t)nrow[j] * ldx + c4);
    }
    if constexpr ((IRX_S2_ABL & 4) != 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if constexpr (BF) acc[t][0] += __uint_as_float(a4.x ^ wc[(j * NT + t) >> 1].x);
        else acc[t][0] += a4.x * wc[j * NT + t].x + a4.w * wc[j * NT + t].w;
      }
    } else
    if constexpr (BF) {
      // a lane's 4 consecutive floats are exactly the 4 k-slots of the 16x16x16 bf16 MFMA: one MFMA replaces four
      const s16x4 pa = irx_frag_bf16(a4.x, a4.y);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int f = j * NT + t;
        const s16x4 pb = (f & 1) ? irx_frag_bf16(wc[f >> 1].z, wc[f >> 1].w) : irx_frag_bf16(wc[f >> 1].x, wc[f >> 1].y);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, pb, acc[t], 0, 0, 0);
      }
    } else {
      // alternate the accumulators so consecutive MFMAs never depend on each other
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wc[j * NT + t].x, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wc[j * NT + t].y, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wc[j * NT + t].z, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wc[j * NT + t].w, acc[t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);             // keep this step's loads between the MFMA blocks
  }
  // D layout: col = lane&15, row = (lane>>4)*4 + r -> pair 16g + 4*g4 + r.  Branch-free, batched read-modify-write:
  // padded pairs go to the dump row (row 64); this wave owns its channel slice, so there is no race.
  if constexpr ((IRX_S2_ABL & 8) != 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" ::"v"(acc[t]));
    return;
  }
  float o[4][NT];
  int oaddr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = (16 * g + 4 * g4 + r < vs) ? orow[r] : 64;
    oaddr[r] = row * LDO + n_base + m;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t) o[r][t] = sOut[oaddr[r] + 16 * t];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t) sOut[oaddr[r] + 16 * t] = o[r][t] + acc[t][r];
}

// Pop the lowest active offset of the mask: kq = offset (or -1 past the end), vq = its pair count, kl = last real offset.
__device__ __forceinline__ void s2_next_offset(unsigned& act, const int* __restrict__ sCnt, int& kq, int& vq, int& kl,
                                               int rot = 0, int K = 27) {
  if (act) {
    kq = __builtin_ctz(act);
    act &= act - 1;
    kq += rot;                                     // act is the mask rotated right by rot (mod K)
    if (kq >= K) kq -= K;
    vq = __builtin_amdgcn_readfirstlane(sCnt[kq]);
    kl = kq;
  } else {
    kq = -1;
    vq = 0;
  }
}

// This lane's input rows of item (kl, vq), one per gather pass; padded pairs re-read row 0.
template <int NIT, int PPP>
__device__ __forceinline__ void s2_gather_rows(const int* __restrict__ list, int pbase, int vq, int (&nrow)[NIT]) {
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int p = pbase + it * PPP;                // < 64: always inside the list (entries >= vq are stale, unused)
    const int r = list[p];
    nrow[it] = ((IRX_S2_ABL & 2) || p >= vq) ? 0 : r;
  }
}

// s_waitcnt vmcnt(n) with expcnt / lgkmcnt untouched (gfx9 encoding: vmcnt = simm16[3:0] | simm16[15:14] << 4)
template <int N>
__device__ __forceinline__ void s2_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

// wn: FRAGMENT-MAJOR weights [K][NCS][NJ][NT][64 lanes][4]: element i of lane l holds
//     W[k][c = 16j + 4(l>>4) + i][n = cs*16*NT + 16t + (l&15)]   (k_permute_w builds it from Conv3d.kernel),
// so each wave-level weight load is one contiguous, fully coalesced 1 KiB read.
// Workgroup = 64 output rows.
//   setup : the four waves compact the tile's table columns (lane == output row; ballot + prefix popcount) into the
//           SHARED pair lists sIn[k][p] (input row) / sRow[k][p] (tile-local output row) and counts sCnt[k]; the set of
//           active offsets is a 27-bit mask in SGPRs.
//   items : the active offsets in order (<= 64 pairs each).  Per item: the rows gathered for it (prefetched into
//           VGPRs DEPTH items ago) are written to the LDS A tile, then the MFMA groups run from VGPR-stationary weights
//           while the loads of item i + DEPTH are issued from inside the first group's MFMA chain.
//   DEPTH : 1 for Cin = 128 (two 64-VGPR weight sets), 2 for Cin <= 64 where an item's MFMA chain (~0.5-1 k cycles)
//           is shorter than the L2 latency (measured before: 30 % of the 64->64 wave cycles waiting in vmcnt(0)).
// Register discipline: the item loop is unrolled DEPTH + 1 times so that every register set is a compile-time name
// (no copies), and the explicit exact vmcnt tells the compiler's waitcnt pass that nothing consumed is pending.
// ST (with BF only): x is bf16 in HBM — a lane gathers 8 bytes (its 4 channels) and the LDS writer stores them as they
// are; y_bf (run time): the output tile is rounded to bf16 on the way out (a single split; offset-split slabs stay fp32).
template <int CIN, int COUT, bool BF, bool ST>
__global__ __launch_bounds__(256, 2)
void k_spconv2(const float* __restrict__ x, const float* __restrict__ wn, const int32_t* __restrict__ nbr, int ld,
               int n_out, int K, int flip_k, float* __restrict__ y, int k_per_split, int accumulate, int ldx, int y_bf,
               const int32_t* __restrict__ tile_order) {
  static_assert(BF || !ST, "bf16 storage implies bf16 operands");
  // ldx = row stride of x in floats (CIN for a dense tensor; > CIN when x is the leading CIN columns of wider rows:
  // the multiview stem, irx_spconv.hip "wide stem"; rows then need only 4-byte alignment)
  // accumulate != 0 (single split only): the tile is ADDED to the rows already in y (gradient accumulation).
  // blockIdx.y = offset split: this workgroup handles offsets [kb, ke) and writes its partial tile to slab
  // blockIdx.y of y (slabs are summed by k_wgrad_reduce; a single split writes the result directly).
  constexpr int TM = S2_TM;
  const int kb = blockIdx.y * k_per_split;
  const int ke = (kb + k_per_split < K) ? kb + k_per_split : K;
  y += (size_t)blockIdx.y * n_out * COUT;         // (slabs of an offset split are fp32: y_bf is 0 then)
  using RT = typename std::conditional<ST, uint2, float4>::type;
  constexpr int NT = (COUT >= 128) ? 2 : 1;       // 16-column tiles per wave
  constexpr int NCS = COUT / (16 * NT);           // channel slices (waves along N)
  constexpr int NGP = 4 / NCS;                    // waves along the pair-group dimension
  // A-tile row stride: + 8 floats. A fragment is one ds_read_b128 per lane; the LDS serves a wave's b128 read in four groups
  // of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) over 64 banks, and with a stride of Cin + 4 two lanes of
  // every group shared a bank quad (rocprofv3: SQ_LDS_BANK_CONFLICT = 30 % of SQ_LDS_IDX_ACTIVE); Cin + 8 is conflict-free.
  constexpr int LDA = CIN + IRX_S2_LDA_PAD;
  constexpr int LDO = COUT + 4;
  constexpr int LPR = CIN / 4;                    // lanes (float4) per gathered row
  constexpr int PPI = 64 / LPR;                   // pairs per wave-instruction
  constexpr int PPP = 4 * PPI;                    // pairs per workgroup pass
  constexpr int NIT = 64 / PPP;                   // gather passes for a full 64-pair item
  constexpr int NJ = CIN / 16;
  constexpr int KMAX = 27;
  constexpr int DEPTH = (CIN >= 128) ? 1 : 2;     // items of prefetch distance (bf16, Cin 128: depth 2 fits in VGPRs but measured equal)
  constexpr int NS = DEPTH + 1;                   // register sets
  using WT = typename std::conditional<BF, uint4, float4>::type;   // one weight load (16 B / lane)
  constexpr int WN = BF ? NJ * NT / 2 : NJ * NT;   // weight loads per item
  constexpr int LPC = WN + NJ;                    // loads per prefetch chain (exact)
  static_assert(NCS * NGP == 4, "4 waves");
  static_assert(NIT == NJ, "one gather pass per MFMA k-step");
  __shared__ __attribute__((aligned(16))) float sOut[(TM + 1) * LDO];   // + dump row for padded pairs
  // A tile, DOUBLE-buffered in bf16 mode (half-size tiles: no resident workgroup is lost): item i + 1 is then written
  // while slower waves still read item i, and the barrier that protected the single buffer goes away (one barrier per
  // item instead of two; barrier 2 of item i + 1 already orders the reads of item i before the writes of item i + 2):
  // 64->64 151 -> 146 us, 128->128 129 -> 128 us. In fp32 the second buffer costs the 64-channel kernels their third
  // resident workgroup per CU and that is worth more than the barrier (299 -> 337 us): single buffer there.
  constexpr bool DB = BF;
  constexpr int ABUF = BF ? 64 * (CIN + 8) / 2 : 64 * LDA;    // floats per buffer
  __shared__ __attribute__((aligned(16))) float sA_all[(DB ? 2 : 1) * ABUF];
  int abuf = 0;
  __shared__ int sIn[KMAX * TM];
  __shared__ unsigned char sRow[KMAX * TM];
  __shared__ int sCnt[32];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: branches on it are uniform for the compiler too
  const int m = lane & 15, g4 = lane >> 4;
  const int cs = wave % NCS, gp = wave / NCS;
  const int n_base = cs * 16 * NT;
  // tile_order (irx_sched.hip): the heaviest tiles start first, so that the last round of workgroups is short
  const int q0 = (tile_order ? __builtin_amdgcn_readfirstlane(tile_order[blockIdx.x]) : (int)blockIdx.x) * TM;
  const int sub = lane / LPR;                     // which of the PPI pairs of a pass this lane serves
  const int c4 = (lane % LPR) * 4;
  const int pbase = wave * PPI + sub;             // this lane's item-local pair in pass 0

  // ---- setup: compact the table columns (wave w takes offsets kb + w, kb + w + 4, ...) ----
  {
    constexpr int KPW = (KMAX + 3) / 4;
    int my[KPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      const int k = kb + wave + 4 * i;
      const int kt = flip_k ? (K - 1 - k) : k;
      my[i] = (k < ke && q0 + lane < n_out) ? nbr[(size_t)kt * ld + q0 + lane] : -1;
    }
    if (tid < 32) sCnt[tid] = 0;
    for (int i = tid; i < (TM + 1) * LDO; i += 256) sOut[i] = 0.f;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      const int k = kb + wave + 4 * i;
      if (k < ke) {
        const unsigned long long valid = __ballot(my[i] >= 0);
        if (my[i] >= 0) {
          const int dst = __popcll(valid & lt);
          sIn[k * TM + dst] = my[i];
          sRow[k * TM + dst] = (unsigned char)lane;
        }
        if (lane == 0) sCnt[k] = __popcll(valid);
      }
    }
  }
  __syncthreads();
  // active offsets of [kb, ke) as a bit mask (wave-uniform, in SGPRs)
  unsigned act;
  {
    const int c = (lane < 32) ? sCnt[lane] : 0;
    act = (unsigned)__ballot(c > 0);
  }
  // every tile walks the offsets from its own starting point: the workgroups resident at one time then stream
  // DIFFERENT weight slices instead of all asking the L2 for the same few lines at once
  int rot = 0;
  if ((IRX_S2_ABL & 32) && gridDim.y == 1 && K > 1) {
    rot = (int)((blockIdx.x * 11u) % (unsigned)K);
    act = ((act >> rot) | (act << (K - rot))) & ((1u << K) - 1u);
  }

  // ---- register sets and the prologue.  Sets are only ever indexed by compile-time constants (integral_constant
  // arguments of the generic lambdas below): a run-time choice of set would demote the arrays to scratch memory. ----
  WT W[3][WN];
  RT S[3][NJ];
  int kk[3] = {-1, -1, -1}, vv[3] = {0, 0, 0};     // offset / pair count of the item living in each set
  // the item whose loads are issued next: offset kq, pair count vq (kq = -1: past the end -> dummy loads of offset kl)
  int kq, vq, kl = kb;
  auto issue = [&](auto T_) __attribute__((always_inline)) {   // prologue: plain back-to-back issue into set T
    constexpr int T = decltype(T_)::value;
    s2_next_offset(act, sCnt, kq, vq, kl, rot, K);
    kk[T] = kq;
    vv[T] = vq;
    int nrow[NJ];
    s2_gather_rows<NIT, PPP>(sIn + kl * TM, pbase, vq, nrow);
    const WT* wnk = reinterpret_cast<const WT*>(wn) + (((size_t)((IRX_S2_ABL & (16 | 64)) ? 0 : kl) * NCS + ((IRX_S2_ABL & 64) ? 0 : cs)) * WN) * 64 + lane;
#pragma unroll
    for (int i = 0; i < WN; ++i) W[T][i] = wnk[(IRX_S2_ABL & 1) ? 0 : (size_t)i * 64];
#pragma unroll
    for (int j = 0; j < NJ; ++j) S[T][j] = s2_ldrow<ST>(x, (size_t)nrow[j] * ldx + c4);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  issue(I0{});
  if (DEPTH == 2) issue(I1{});
#ifdef IRX_S2_PROF
  long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = clock64();
  const long long tstart = tlast;
#endif

  // One item: consumes register set C, prefetches item i + DEPTH into register set T.
  auto item = [&](auto C_, auto T_) __attribute__((always_inline)) {
    constexpr int C = decltype(C_)::value, T = decltype(T_)::value;
    S2_TICK(0);
    const int k = kk[C], vs = vv[C];
    const int vpad = (vs + 15) & ~15;
    const int npass = (vpad + PPP - 1) / PPP;      // block-uniform
    float* sA = sA_all + (DB ? abuf * ABUF : 0);
    if (DB) abuf ^= 1;
    else if (!(IRX_S2_ABL & 128)) __syncthreads(); // previous item's fragment reads are done
    S2_TICK(1);
    s2_wait_vmcnt<(DEPTH - 1) * LPC>();            // this item's rows + weights have landed (later items may be in flight)
    S2_TICK(2);
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      if (it < npass && !(IRX_S2_ABL & 256)) {     // (component-wise: a struct copy out of S[][] keeps the sets in scratch)
        if constexpr (ST)
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(sA) + (pbase + it * PPP) * (CIN + 8) + c4) =
              make_uint2(S[C][it].x, S[C][it].y);
        else if constexpr (BF)
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(sA) + (pbase + it * PPP) * (CIN + 8) + c4) =
              make_uint2(irx_pk_bf16(S[C][it].x, S[C][it].y), irx_pk_bf16(S[C][it].z, S[C][it].w));
        else
          *reinterpret_cast<float4*>(&sA[(pbase + it * PPP) * LDA + c4]) =
              make_float4(S[C][it].x, S[C][it].y, S[C][it].z, S[C][it].w);
      }
    S2_TICK(3);
    if (!(IRX_S2_ABL & 128)) __syncthreads();
    S2_TICK(4);
    // ---- the item to prefetch ----
    s2_next_offset(act, sCnt, kq, vq, kl, rot, K);
    kk[T] = kq;
    vv[T] = vq;
    int nrow[NJ];
    s2_gather_rows<NIT, PPP>(sIn + kl * TM, pbase, vq, nrow);
    const WT* wnk = reinterpret_cast<const WT*>(wn) + (((size_t)((IRX_S2_ABL & (16 | 64)) ? 0 : kl) * NCS + ((IRX_S2_ABL & 64) ? 0 : cs)) * WN) * 64 + lane;
    S2_TICK(5);
    // ---- MFMA over dense 16-pair groups; this wave's channel slice ----
    const unsigned char* lrow = sRow + k * TM;
    int g = (IRX_S2_ABL & 512) ? 1000 : gp;      // (ablation 512: no group work at all, the prefetch branch below runs)
    // bf16, Cin 128: an item with two or more groups for this wave issues half of the chain's loads in each of the first
    // two (128->128: 131.7 -> 125.4 us; no effect in fp32, slightly worse for Cin 64)
    if (IRX_S2_SPLITPF && BF && NJ >= 8 && (g + NGP) * 16 < vpad) {
      s2_group<CIN, COUT, NJ, NT, LDA, LDO, 2, BF, ST, WT, WN, RT>(sA, sOut, lrow, g, vs, m, g4, n_base, W[C], W[T],
                                                                   S[T], nrow, x, c4, wnk, ldx);
      g += NGP;
      s2_group<CIN, COUT, NJ, NT, LDA, LDO, 3, BF, ST, WT, WN, RT>(sA, sOut, lrow, g, vs, m, g4, n_base, W[C], W[T],
                                                                   S[T], nrow, x, c4, wnk, ldx);
      for (g += NGP; g * 16 < vpad; g += NGP)
        s2_group<CIN, COUT, NJ, NT, LDA, LDO, 0, BF, ST, WT, WN, RT>(sA, sOut, lrow, g, vs, m, g4, n_base, W[C], W[T],
                                                                     S[T], nrow, x, c4, wnk, ldx);
    } else if (!(IRX_S2_ABL & 512) && (NGP == 1 || g * 16 < vpad)) {
      s2_group<CIN, COUT, NJ, NT, LDA, LDO, 1, BF, ST, WT, WN, RT>(sA, sOut, lrow, g, vs, m, g4, n_base, W[C], W[T],
                                                                   S[T], nrow, x, c4, wnk, ldx);
      for (g += NGP; g * 16 < vpad; g += NGP)
        s2_group<CIN, COUT, NJ, NT, LDA, LDO, 0, BF, ST, WT, WN, RT>(sA, sOut, lrow, g, vs, m, g4, n_base, W[C], W[T],
                                                                     S[T], nrow, x, c4, wnk, ldx);
    } else {                                       // a wave without a group in this item still prefetches its share
#pragma unroll
      for (int i = 0; i < WN; ++i) W[T][i] = wnk[(IRX_S2_ABL & 1) ? 0 : (size_t)i * 64];
#pragma unroll
      for (int j = 0; j < NJ; ++j) S[T][j] = s2_ldrow<ST>(x, (size_t)nrow[j] * ldx + c4);
    }
    S2_TICK(6);
  };
  if (kk[0] >= 0 && !(IRX_S2_ABL & 4096)) {   // (ablation 4096: no item loop at all)
    if (DEPTH == 1) {
      while (true) {
        item(I0{}, I1{});
        if (kk[1] < 0) break;
        item(I1{}, I0{});
        if (kk[0] < 0) break;
      }
    } else {
      while (true) {
        item(I0{}, I2{});
        if (kk[1] < 0) break;
        item(I1{}, I0{});
        if (kk[2] < 0) break;
        item(I2{}, I1{});
        if (kk[0] < 0) break;
      }
    }
  }
#ifdef IRX_S2_PROF
  if (lane == 0) {
    for (int i = 0; i < 7; ++i) atomicAdd(&g_s2_prof[i], (unsigned long long)prof[i]);
    atomicAdd(&g_s2_prof[8], (unsigned long long)(clock64() - tstart));
    atomicAdd(&g_s2_prof[9], 1ull);
  }
#endif
  __syncthreads();
  // ---- write the tile: COUT/4 float4 per row ----
  constexpr int F4 = COUT / 4;
  for (int f = tid; f < TM * F4; f += 256) {
    const int row = f / F4, cc = (f % F4) * 4;
    if (q0 + row < n_out && !(IRX_S2_ABL & 2048)) {
      float4 o = *reinterpret_cast<const float4*>(&sOut[row * LDO + cc]);
      const size_t off = (size_t)(q0 + row) * COUT + cc;
      if (accumulate) {
        const float4 e = irx_ld4(y, off, y_bf);
        o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
      }
      irx_st4(y, off, y_bf, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// part[s][k][c][n] = sum over valid pairs of split s:  x[in][c] * dy[out][n]
template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void k_spconv2_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const int32_t* __restrict__ nbr, int ld, int n_out,
                                                          int K, int rows_per_split, float* __restrict__ part, int ldx,
                                                          int dy_bf) {
  constexpr int TC = CIN / 16, TN = COUT / 16;
  constexpr int CW = (TC >= 4) ? TC / 4 : 1;             // c-tiles per wave
  constexpr int NW = (TC >= 4) ? TN : TN / (4 / TC);     // n-tiles per wave
  constexpr int LDX = CIN + 16, LDD = COUT + 16;         // consecutive pairs 16 banks apart
  constexpr int LPX = CIN / 4, LPD = COUT / 4;
  __shared__ __attribute__((aligned(16))) float sX[S2_TM * LDX];
  __shared__ __attribute__((aligned(16))) float sD[S2_TM * LDD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, g4 = lane >> 4;
  const int s = blockIdx.x, k = blockIdx.y;
  const int ct0 = (TC >= 4) ? wave * CW : (wave % TC);
  const int nt0 = (TC >= 4) ? 0 : (wave / TC) * NW;
  const int qbeg = s * rows_per_split;
  int qend = qbeg + rows_per_split;
  if (qend > n_out) qend = n_out;

  f32x4 acc[CW][NW];
#pragma unroll
  for (int a = 0; a < CW; ++a)
#pragma unroll
    for (int b = 0; b < NW; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int q0 = qbeg; q0 < qend; q0 += S2_TM) {
    int my = -1;
    if (q0 + lane < qend) my = nbr[(size_t)k * ld + q0 + lane];
    const PairList pl = compact_pairs(my, lane);
    if (pl.v == 0) continue;
    const int vpad = (pl.v + 3) & ~3;
    __syncthreads();
    {  // x rows of the pairs
      constexpr int PPI = 64 / LPX;
      const int sub = lane / LPX, c4 = (lane % LPX) * 4;
      for (int p0 = wave * PPI; p0 < vpad; p0 += 4 * PPI) {
        const int p = p0 + sub;
        const int idx = __shfl(pl.in_of_pair, p & 63);
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < pl.v) val = *reinterpret_cast<const float4*>(x + (size_t)idx * ldx + c4);
        if (p < vpad) *reinterpret_cast<float4*>(&sX[p * LDX + c4]) = val;
      }
    }
    {  // dy rows of the pairs
      constexpr int PPI = 64 / LPD;
      const int sub = lane / LPD, c4 = (lane % LPD) * 4;
      for (int p0 = wave * PPI; p0 < vpad; p0 += 4 * PPI) {
        const int p = p0 + sub;
        const int orow = __shfl(pl.row_of_pair, p & 63);
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < pl.v) val = irx_ld4(dy, (size_t)(q0 + orow) * COUT + c4, dy_bf);
        if (p < vpad) *reinterpret_cast<float4*>(&sD[p * LDD + c4]) = val;
      }
    }
    __syncthreads();
    for (int ks = 0; ks * 4 < vpad; ++ks) {
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

// Conv3d.kernel w -> fragment-major image for k_spconv2.  Kernel-relative dims (cin_k = reduction, cout_k = outputs):
//   forward      : W[k][c][n] = w[(k*cin_k + c)*cout_k + n]
//   data-gradient: W[k][c][n] = w[(k*cout_k + n)*cin_k + c]   (w is the forward [K][conv Cin][conv Cout] tensor)
// bf16 image: fragments 2p and 2p + 1 of a lane share one 16-byte element -> [K*NCS][NF/2][64 lanes][2][4 bf16]
__device__ __forceinline__ void s2_store_frag_bf16(float* wf, size_t kcs, int nf, int f, int lane, float4 v) {
  uint2 o;
  o.x = irx_pk_bf16(v.x, v.y);
  o.y = irx_pk_bf16(v.z, v.w);
  reinterpret_cast<uint2*>(wf)[((kcs * (nf / 2) + (f >> 1)) * 64 + lane) * 2 + (f & 1)] = o;
}

__global__ void k_permute_w(const float* __restrict__ w, int K, int cin, int cout, int trans_w, float* __restrict__ wf,
                            int bf16, int src_cin) {
  // src_cin: channel count of the SOURCE tensor per offset (forward only): the image covers its first `cin` channels
  const int NT = cout >= 128 ? 2 : 1;
  const int NCS = cout / (16 * NT);
  const int NJ = cin / 16;
  const size_t total = (size_t)K * cin * cout / 4;          // float4 elements
  size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= total) return;
  const int lane = (int)(f & 63);
  size_t r = f >> 6;
  const int t = (int)(r % NT); r /= NT;
  const int j = (int)(r % NJ); r /= NJ;
  const int cs = (int)(r % NCS); r /= NCS;
  const int k = (int)r;
  const int n = cs * 16 * NT + 16 * t + (lane & 15);
  const int c0 = 16 * j + 4 * (lane >> 4);
  float4 v;
  float* pv = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    pv[i] = trans_w ? w[((size_t)k * cout + n) * cin + c0 + i] : w[((size_t)k * src_cin + c0 + i) * cout + n];
  if (bf16)
    s2_store_frag_bf16(wf, (size_t)k * NCS + cs, NJ * NT, j * NT + t, lane, v);
  else
    reinterpret_cast<float4*>(wf)[f] = v;
}

// Dev: resident workgroups per CU the runtime reports for the main kernels (tools/micro/occupancy.py)
extern "C" int irx_debug_occupancy(int which) {
  int n = -1;
  hipError_t e = hipSuccess;
  switch (which) {
    case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spconv2<128, 128, false, false>, 256, 0); break;
    case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spconv2<64, 64, false, false>, 256, 0); break;
    case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spconv2<128, 128, true, true>, 256, 0); break;
    case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spconv2<64, 64, true, true>, 256, 0); break;
    default: return -2;
  }
  return e == hipSuccess ? n : -1;
}

// ---------------------------------------------------------------------------- host dispatch ---
static inline hipStream_t S(void* s) { return (hipStream_t)s; }

// A/B switch for tests and profiling: IRX_SPCONV_V1 = any of "f" (forward), "d" (data-gradient), "w" (weight-
// gradient) or "1"/"all" -> those passes use the first-generation kernels.
bool irx_spconv2_enabled(char pass) {
  static const char* e = getenv("IRX_SPCONV_V1");
  if (!e) return true;
  if (strchr(e, '1') || strchr(e, 'a')) return false;
  return strchr(e, pass) == nullptr;
}

bool irx_spconv2_supported(int cin, int cout) {
  return (cin == 32 || cin == 64 || cin == 128) && (cout == 32 || cout == 64 || cout == 128);
}

// dev knob: IRX_S2_EXTRA_LDS = bytes of dynamic LDS added to every k_spconv2 launch — fewer resident workgroups per CU, to
// measure how much a resident workgroup is worth
static size_t s2_extra_lds() {
  static const size_t v = getenv("IRX_S2_EXTRA_LDS") ? (size_t)atol(getenv("IRX_S2_EXTRA_LDS")) : 0;
  return v;
}

template <int CIN, bool BF, bool ST>
static void launch_fwd2(int cout, dim3 grid, hipStream_t st, const float* x, const float* wn, const int32_t* nbr,
                        int ld, int n_out, int K, int flip_k, float* y, int kps, int acc, int ldx, int y_bf,
                        const int32_t* ord) {
  if (cout == 128) k_spconv2<CIN, 128, BF, ST><<<grid, 256, s2_extra_lds(), st>>>(x, wn, nbr, ld, n_out, K, flip_k, y, kps, acc, ldx, y_bf, ord);
  else if (cout == 64) k_spconv2<CIN, 64, BF, ST><<<grid, 256, s2_extra_lds(), st>>>(x, wn, nbr, ld, n_out, K, flip_k, y, kps, acc, ldx, y_bf, ord);
  else k_spconv2<CIN, 32, BF, ST><<<grid, 256, s2_extra_lds(), st>>>(x, wn, nbr, ld, n_out, K, flip_k, y, kps, acc, ldx, y_bf, ord);
}

// Output rows per workgroup: 64.  128-row tiles stream half the weight bytes per useful FLOP but MEASURED SLOWER twice:
// round 1, 4 waves (N = 81 k, 128->128: 498 vs 441 us: half the waves per CU); round 2, 8 waves x 16 columns, double-
// buffered A tile, prefetch distance 2 (429 vs 379 us; tools/micro/k_spconv3_tall_tile.hip.txt): the 68 KB fp32 tile
// leaves room for ONE workgroup per CU, every wave is then in the same phase of the item (one barrier domain) and the
// MFMA pipe idles through each gather-write / barrier / look-ahead phase -- MFMA utilisation 48 % against 73 % here.
int irx_spconv2_tile(int n_out) {
  (void)n_out;
  return S2_TM;
}

// (Threshold: a layer with >= 400 tiles is left whole. In the training step the other encoder's stream fills the CUs
// a 400..768-tile launch leaves idle, so splitting those only added slab traffic and a reduce launch: 768 -> 256
// measured +1 % end to end, 128 equal, 64 worse (round 2). Round 3, four alternating 80-step runs on one box: 256 -> 400,
// which splits the 317-tile stride-8 level of the scene encoder four ways (150 -> ~60 us alone): 9.537 -> 9.505 ms/step.)
// Offset splits for latency-bound (small) layers: a tile's 27 offsets form a serial chain of ~4 us each, so
// when there are too few tiles to fill the chip the offsets are spread over `splits` workgroups per tile.
int irx_spconv2_splits(int n_out, int K) {
  static const char* e = getenv("IRX_SPCONV_KSPLIT");
  if (e) { int s = atoi(e); return s < 1 ? 1 : (s > K ? K : s); }
  const int tiles = irx_cdiv(n_out, irx_spconv2_tile(n_out));
  static const int full = getenv("IRX_SPCONV_SPLIT_BELOW") ? atoi(getenv("IRX_SPCONV_SPLIT_BELOW")) : 400;
  if (tiles >= full || K < 4) return 1;
  static const int target = getenv("IRX_SPCONV_SPLIT_TARGET") ? atoi(getenv("IRX_SPCONV_SPLIT_TARGET")) : 1024;   // dev A/B knob
  int s = irx_cdiv(target, tiles);
  if (s > 9) s = 9;
  if (s > K) s = K;
  const int kps = irx_cdiv(K, s);
  return irx_cdiv(K, kps);                       // no empty splits
}

// y: result (splits == 1; accumulate != 0 adds to it) or `splits` slabs of [n_out][cout] partial sums
int irx_spconv2_launch(const float* x, const float* wn, const int32_t* nbr, int ld, int n_out, int K, int cin,
                       int cout, int flip_k, float* y, int splits, int accumulate, hipStream_t st, int ldx, IrxStore ty) {
  if (ldx <= 0) ldx = cin;
  IRX_REQUIRE(!ty.x || irx_mode_now(), "irx_spconv_fwd: a bf16 input needs the bf16 compute mode");
  const int y_bf = (splits == 1) ? ty.y : 0;
  const int acc = (splits == 1) ? accumulate : 0;
  IRX_REQUIRE(K <= 27, "irx_spconv_fwd: K = %d > 27 unsupported by the fast path", K);
  dim3 grid(irx_cdiv(n_out, S2_TM), splits);
  const int kps = irx_cdiv(K, splits);
  irx_bracket_begin(st);
  if (irx_mode_now() && ty.x) {
    if (cin == 128) launch_fwd2<128, true, true>(cout, grid, st, x, wn, nbr, ld, n_out, K, flip_k, y, kps, acc, ldx, y_bf, ty.order);
    else if (cin == 64) launch_fwd2<64, true, true>(cout, grid, st, x, wn, nbr, ld, n_out, K, flip_k, y, kps, acc, ldx, y_bf, ty.order);
    else launch_fwd2<32, true, true>(cout, grid, st, x, wn, nbr, ld, n_out, K, flip_k, y, kps, acc, ldx, y_bf, ty.order);
  } else if (irx_mode_now()) {
    if (cin == 128) launch_fwd2<128, true, false>(cout, grid, st, x, wn, nbr, ld, n_out, K, flip_k, y, kps, acc, ldx, y_bf, ty.order);
    else if (cin == 64) launch_fwd2<64, true, false>(cout, grid, st, x, wn, nbr, ld, n_out, K, flip_k, y, kps, acc, ldx, y_bf, ty.order);
    else launch_fwd2<32, true, false>(cout, grid, st, x, wn, nbr, ld, n_out, K, flip_k, y, kps, acc, ldx, y_bf, ty.order);
  } else {
    if (cin == 128) launch_fwd2<128, false, false>(cout, grid, st, x, wn, nbr, ld, n_out, K, flip_k, y, kps, acc, ldx, y_bf, ty.order);
    else if (cin == 64) launch_fwd2<64, false, false>(cout, grid, st, x, wn, nbr, ld, n_out, K, flip_k, y, kps, acc, ldx, y_bf, ty.order);
    else launch_fwd2<32, false, false>(cout, grid, st, x, wn, nbr, ld, n_out, K, flip_k, y, kps, acc, ldx, y_bf, ty.order);
  }
  irx_bracket_end(st);
  IRX_CHECK_LAUNCH("irx_spconv_fwd(v2)");
  return IRX_OK;
}

// ---- data-gradient of a stride-2 (2^3) convolution, tiled by PARENT rows (round 3) ------------------------------------
// dx[q] = dy[parent(q)] . W[koff(q)]^T: every fine row q has exactly one (parent, offset) pair. Through k_spconv2 (tiles of 64
// fine rows over the transposed child table) a tile's 8 offsets hold ~8 pairs each: eight half-empty 16-row MFMA groups and
// eight weight slices per 64 output rows, and an LDS output tile that is only ever written once per row.
// Here a workgroup owns 64 consecutive PARENT rows: their dy rows are one contiguous block (loaded once, coalesced, into the
// LDS A tile — no gather), offset k pairs the 15-32 parents that have a child k with that child's row (one ballot compaction
// of child[k][p0 .. p0+63]), the MFMA groups read their A fragments straight from the tile through the compacted parent
// list, and the results go from the accumulators to dx (each row written exactly once: no LDS output tile, no
// read-modify-write, one barrier per workgroup). Per output row: 2.5 x fewer weight-slice loads and ~30 % fewer MFMA rows.
// Same weight image, same reduction order as k_spconv2: bit-identical results. fp32 only (the bf16 modes keep the old path).
template <int CR, int CO, int NJ, int NT, int LDA, int WN, bool PF>
__device__ __forceinline__ void ud_group(const float* __restrict__ sA, const unsigned char* __restrict__ sel,
                                         const int* __restrict__ dst, int g, int cnt, int m, int g4, int n_base,
                                         const float4 (&wc)[WN], float4 (&wx)[WN], const float4* __restrict__ wnk,
                                         float* __restrict__ dx) {
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int pi = 16 * g + m;
  const int row = pi < cnt ? (int)sel[pi] : 0;                     // padded pairs re-read row 0; their results are dropped
  const float4* pa = reinterpret_cast<const float4*>(&sA[row * LDA + 4 * g4]);
  int orow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) orow[r] = (16 * g + 4 * g4 + r < cnt) ? dst[16 * g + 4 * g4 + r] : -1;
  float4 a_nxt = pa[0];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float4 a4 = a_nxt;
    if (j + 1 < NJ) a_nxt = pa[4 * (j + 1)];
    if (PF) {                                                        // this step's share of the next offset's weight slice
#pragma unroll
      for (int i = j * NT; i < (j + 1) * NT; ++i) wx[i] = wnk[(size_t)i * 64];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wc[j * NT + t].x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wc[j * NT + t].y, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wc[j * NT + t].z, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wc[j * NT + t].w, acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  // D layout: col = lane & 15, row = 4 * (lane >> 4) + r -> pair 16 g + 4 g4 + r -> fine row orow[r]
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (orow[r] >= 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) dx[(size_t)orow[r] * CO + n_base + 16 * t + m] = acc[t][r];
    }
}

template <int CR, int CO>
__global__ __launch_bounds__(256, 2) void k_updgrad(const float* __restrict__ dy, const float* __restrict__ wn,
                                                    const int32_t* __restrict__ child, int ldc, int n_parent,
                                                    float* __restrict__ dx) {
  constexpr int NT = (CO >= 128) ? 2 : 1;
  constexpr int NCS = CO / (16 * NT);
  constexpr int NGP = 4 / NCS;
  constexpr int NJ = CR / 16;
  constexpr int WN = NJ * NT;
  constexpr int LDA = CR + 8;
  constexpr int LPR = CR / 4;                     // lanes (float4) per dy row
  constexpr int RPP = 256 / LPR;                  // rows per load pass
  static_assert(NCS * NGP == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) float sA[64 * LDA];
  __shared__ int sDst[8 * 64];
  __shared__ unsigned char sSel[8 * 64];
  __shared__ int sCnt[8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g4 = lane >> 4;
  const int cs = wave % NCS, gp = wave / NCS;
  const int n_base = cs * 16 * NT;
  const int p0 = blockIdx.x * 64;
  // pairs of offset k: the parents of this tile that have a child k (wave w compacts offsets w and w + 4)
  {
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = wave + 4 * i;
      const int e = (p0 + lane < n_parent) ? child[(size_t)k * ldc + p0 + lane] : -1;
      const unsigned long long valid = __ballot(e >= 0);
      if (e >= 0) {
        const int d = __popcll(valid & lt);
        sSel[k * 64 + d] = (unsigned char)lane;
        sDst[k * 64 + d] = e;
      }
      if (lane == 0) sCnt[k] = __popcll(valid);
    }
  }
  // the tile's dy rows: one contiguous block
#pragma unroll
  for (int it = 0; it < 64 / RPP; ++it) {
    const int row = tid / LPR + it * RPP, c4 = (tid % LPR) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p0 + row < n_parent) v = *reinterpret_cast<const float4*>(dy + (size_t)(p0 + row) * CR + c4);
    *reinterpret_cast<float4*>(&sA[row * LDA + c4]) = v;
  }
  __syncthreads();
  unsigned act;
  {
    const int c = (lane < 8) ? sCnt[lane] : 0;
    act = (unsigned)__ballot(c > 0) & 0xFFu;
  }
  if (!act) return;
  const float4* wn4 = reinterpret_cast<const float4*>(wn);
  float4 W[2][WN];
  int kk[2] = {-1, -1};
  auto slice = [&](int k) __attribute__((always_inline)) { return wn4 + (((size_t)k * NCS + cs) * WN) * 64 + lane; };
  kk[0] = __builtin_ctz(act);
  act &= act - 1;
  {
    const float4* w0 = slice(kk[0]);
#pragma unroll
    for (int i = 0; i < WN; ++i) W[0][i] = w0[(size_t)i * 64];
  }
  auto item = [&](auto C_, auto T_) __attribute__((always_inline)) {
    constexpr int C = decltype(C_)::value, T = decltype(T_)::value;
    const int k = kk[C];
    const int cnt = __builtin_amdgcn_readfirstlane(sCnt[k]);
    int kn = -1;
    if (act) {
      kn = __builtin_ctz(act);
      act &= act - 1;
    }
    kk[T] = kn;
    const float4* wnk = slice(kn < 0 ? k : kn);
    const unsigned char* sel = sSel + k * 64;
    const int* dst = sDst + k * 64;
    int g = gp;
    if (g * 16 < cnt) {
      if (kn >= 0) ud_group<CR, CO, NJ, NT, LDA, WN, true>(sA, sel, dst, g, cnt, m, g4, n_base, W[C], W[T], wnk, dx);
      else ud_group<CR, CO, NJ, NT, LDA, WN, false>(sA, sel, dst, g, cnt, m, g4, n_base, W[C], W[T], wnk, dx);
      for (g += NGP; g * 16 < cnt; g += NGP)
        ud_group<CR, CO, NJ, NT, LDA, WN, false>(sA, sel, dst, g, cnt, m, g4, n_base, W[C], W[T], wnk, dx);
    } else if (kn >= 0) {                          // a wave without a group in this item still fetches its next slice
#pragma unroll
      for (int i = 0; i < WN; ++i) W[T][i] = wnk[(size_t)i * 64];
    }
  };
  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;
  while (true) {
    item(J0{}, J1{});
    if (kk[1] < 0) break;
    item(J1{}, J0{});
    if (kk[0] < 0) break;
  }
}

bool irx_updgrad_supported(int cr, int co) { return (cr == 64 || cr == 128) && (co == 32 || co == 64 || co == 128); }

// dx [n_child][co] = data-gradient of a 2^3 / stride-2 convolution from dy [n_parent][cr]; child: the forward table
// [8][ldc] (child row of parent p at offset k, or -1); wn: the data-gradient weight image (k_permute_w, transposed).
int irx_updgrad_launch(const float* dy, const float* wn, const int32_t* child, int ldc, int n_parent, int cr, int co,
                       float* dx, hipStream_t st) {
  IRX_REQUIRE(irx_updgrad_supported(cr, co), "irx_updgrad: channels (%d, %d) unsupported", cr, co);
  if (n_parent <= 0) return IRX_OK;
  const dim3 grid(irx_cdiv(n_parent, 64));
  irx_bracket_begin(st);
#define UD(CR_, CO_) k_updgrad<CR_, CO_><<<grid, 256, 0, st>>>(dy, wn, child, ldc, n_parent, dx)
  if (cr == 128 && co == 128) UD(128, 128);
  else if (cr == 128 && co == 64) UD(128, 64);
  else if (cr == 128 && co == 32) UD(128, 32);
  else if (cr == 64 && co == 128) UD(64, 128);
  else if (cr == 64 && co == 64) UD(64, 64);
  else UD(64, 32);
#undef UD
  irx_bracket_end(st);
  IRX_CHECK_LAUNCH("irx_updgrad");
  return IRX_OK;
}

// All layers of an encoder in ONE launch (the per-layer permutes were 54 launches of ~5 us per training step).
__global__ void k_permute_w_multi(IrxPermuteJobs J, int trans_w, int bf16) {
  const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int j = 0;
  while (j < J.n && f >= J.end4[j]) ++j;
  if (j >= J.n) return;
  const size_t fl = f - (j ? J.end4[j - 1] : 0);
  const int K = J.K[j], cin = J.cin[j], cout = J.cout[j];
  const float* __restrict__ w = J.w[j];
  const int NT = cout >= 128 ? 2 : 1;
  const int NCS = cout / (16 * NT);
  const int NJ = cin / 16;
  const int lane = (int)(fl & 63);
  size_t r = fl >> 6;
  const int t = (int)(r % NT); r /= NT;
  const int jj = (int)(r % NJ); r /= NJ;
  const int cs = (int)(r % NCS); r /= NCS;
  const int k = (int)r;
  (void)K;
  const int n = cs * 16 * NT + 16 * t + (lane & 15);
  const int c0 = 16 * jj + 4 * (lane >> 4);
  float4 v;
  float* pv = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    pv[i] = trans_w ? w[((size_t)k * cout + n) * cin + c0 + i] : w[((size_t)k * cin + c0 + i) * cout + n];
  if (bf16)
    s2_store_frag_bf16(J.dst[j], (size_t)k * NCS + cs, NJ * NT, jj * NT + t, lane, v);
  else
    reinterpret_cast<float4*>(J.dst[j])[fl] = v;
}

int irx_permute_w_multi_launch(const IrxPermuteJobs& jobs, int trans_w, hipStream_t st) {
  if (jobs.n == 0) return IRX_OK;
  const size_t total = jobs.end4[jobs.n - 1];
  k_permute_w_multi<<<irx_cdiv((long long)total, 256), 256, 0, st>>>(jobs, trans_w, irx_mode_now());
  IRX_CHECK_LAUNCH("irx_encoder(permute)");
  return IRX_OK;
}

int irx_permute_w_launch(const float* w, int K, int cin, int cout, int trans_w, float* wf, hipStream_t st, int src_cin) {
  const size_t total = (size_t)K * cin * cout / 4;
  k_permute_w<<<irx_cdiv((long long)total, 256), 256, 0, st>>>(w, K, cin, cout, trans_w, wf, irx_mode_now(),
                                                              src_cin > 0 ? src_cin : cin);
  IRX_CHECK_LAUNCH("irx_spconv_fwd(permute)");
  return IRX_OK;
}

template <int CIN>
static void launch_wg2(int cout, dim3 grid, hipStream_t st, const float* x, const float* dy, const int32_t* nbr,
                       int ld, int n_out, int K, int rps, float* part, int ldx, int dy_bf) {
  if (cout == 128) k_spconv2_wgrad<CIN, 128><<<grid, 256, 0, st>>>(x, dy, nbr, ld, n_out, K, rps, part, ldx, dy_bf);
  else if (cout == 64) k_spconv2_wgrad<CIN, 64><<<grid, 256, 0, st>>>(x, dy, nbr, ld, n_out, K, rps, part, ldx, dy_bf);
  else k_spconv2_wgrad<CIN, 32><<<grid, 256, 0, st>>>(x, dy, nbr, ld, n_out, K, rps, part, ldx, dy_bf);
}

int irx_spconv2_wgrad_launch(const float* x, const float* dy, const int32_t* nbr, int ld, int n_out, int K,
                             int cin, int cout, int splits, int rps, float* part, hipStream_t st, int ldx, int dy_bf) {
  if (ldx <= 0) ldx = cin;
  dim3 grid(splits, K);
  irx_bracket_begin(st);
  if (cin == 128) launch_wg2<128>(cout, grid, st, x, dy, nbr, ld, n_out, K, rps, part, ldx, dy_bf);
  else if (cin == 64) launch_wg2<64>(cout, grid, st, x, dy, nbr, ld, n_out, K, rps, part, ldx, dy_bf);
  else launch_wg2<32>(cout, grid, st, x, dy, nbr, ld, n_out, K, rps, part, ldx, dy_bf);
  irx_bracket_end(st);
  IRX_CHECK_LAUNCH("irx_spconv_wgrad(v2)");
  return IRX_OK;
}
