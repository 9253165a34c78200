// irx_sched.hip — launch order of the output tiles of k_spconv2 (round 3).
//
// k_spconv2 gives every 64 consecutive (Morton-ordered) output rows to one workgroup and the hardware starts workgroups in
// blockIdx order as slots free up. A tile's duration follows its work — per active offset a fixed part plus one part per
// 16-pair group — and on ScanNet-like scenes that varies 2-3x between tiles (flat surface vs clutter vs boundary). With only
// 2.5 tiles per slot (1270 tiles on 512 slots for the largest 128-channel level) list scheduling in Morton order ends 39 %
// above the balanced makespan in a replay of the measured per-tile costs; longest-tile-first ends 18 % above it (and 9 %
// instead of 23 % for the 64-channel level with 4045 tiles on 768 slots). So: one pass over the neighbour table computes a
// cost class per tile, and a stable counting sort (one workgroup; <= a few thousand tiles) lists the tiles heaviest class first.
// The order changes WHEN a tile is computed, never what is computed: results are bit-identical with and without it.
#include "irx_common.h"

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

#define TS_TM 64          // rows per tile: k_spconv2's S2_TM
#define TS_NCLS 64        // cost classes
#define TS_FIX 3          // cost of an active offset (weight slice fetch, barriers, look-ahead) ...
#define TS_GRP 5          // ... and of each 16-pair group (MFMA chain + epilogue), in the units of the replay (1.0 / 1.7 us)

// one wave per tile: cost = sum over offsets [v > 0] * TS_FIX + ceil(v / 16) * TS_GRP, v = valid entries of the tile's column block
__global__ __launch_bounds__(256) void k_tile_cost(const int32_t* __restrict__ nbr, int ld, int n_out, int K, int ntiles,
                                                   int shift, unsigned char* __restrict__ cls) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= ntiles) return;
  const int q = t * TS_TM + lane;
  int cost = 0;
  for (int k = 0; k < K; ++k) {
    const int e = (q < n_out) ? nbr[(size_t)k * ld + q] : -1;
    const int v = __popcll(__ballot(e >= 0));
    cost += (v > 0 ? TS_FIX : 0) + ((v + 15) >> 4) * TS_GRP;
  }
  int c = cost >> shift;
  if (c > TS_NCLS - 1) c = TS_NCLS - 1;
  if (lane == 0) cls[t] = (unsigned char)(TS_NCLS - 1 - c);      // class 0 = heaviest
}

// stable counting sort of the tiles by class: one workgroup of TO_WAVES waves, wave w owns the w-th contiguous segment of the
// tile list (whole 64-tile chunks); per-wave class histograms in LDS, then position = (tiles of lighter-numbered classes) +
// (same class in earlier waves) + (same class earlier in my wave). The single-wave version took 35-110 us per level.
#define TO_WAVES 16
__global__ __launch_bounds__(64 * TO_WAVES) void k_tile_order(const unsigned char* __restrict__ cls, int ntiles,
                                                              int32_t* __restrict__ order) {
  __shared__ int cnt[TO_WAVES][TS_NCLS];
  __shared__ int cls_base[TS_NCLS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  cnt[wave][lane] = 0;                                 // TS_NCLS == 64 == lanes
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  auto peers_of = [&](unsigned c, bool valid) {
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      const unsigned long long vote = __ballot((c >> b) & 1u);
      m &= ((c >> b) & 1u) ? vote : ~vote;
    }
    return valid ? m : 0ull;
  };
  const int seg = ((ntiles + 64 * TO_WAVES - 1) / (64 * TO_WAVES)) * 64;
  const int t0 = wave * seg, t1 = (t0 + seg < ntiles) ? t0 + seg : ntiles;
  for (int base = t0; base < t1; base += 64) {
    const int t = base + lane;
    const bool valid = t < t1;
    const unsigned c = valid ? cls[t] : 0u;
    const unsigned long long peers = peers_of(c, valid);
    if (valid && (peers & lt) == 0ull) cnt[wave][c] += __popcll(peers);
  }
  __syncthreads();
  // thread (wave, lane = class): tiles of my class in earlier waves; wave 0 also scans the class totals
  int before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < TO_WAVES; ++w) {
    const int h = cnt[w][lane];
    before += (w < wave) ? h : 0;
    total += h;
  }
  if (wave == 0) {
    int incl = total;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int up = __shfl_up(incl, off);
      if (lane >= off) incl += up;
    }
    cls_base[lane] = incl - total;
  }
  __syncthreads();
  cnt[wave][lane] = cls_base[lane] + before;
  __builtin_amdgcn_wave_barrier();
  for (int base = t0; base < t1; base += 64) {
    const int t = base + lane;
    const bool valid = t < t1;
    const unsigned c = valid ? cls[t] : 0u;
    const unsigned long long peers = peers_of(c, valid);
    int pos = 0;
    if (valid) pos = cnt[wave][c] + __popcll(peers & lt);
    __builtin_amdgcn_wave_barrier();
    if (valid && (peers & lt) == 0ull) cnt[wave][c] += __popcll(peers);
    __builtin_amdgcn_wave_barrier();
    if (valid) order[pos] = t;
  }
}

extern "C" size_t irx_tile_order_workspace_bytes(int n_out) { return (size_t)irx_cdiv(n_out > 0 ? n_out : 1, TS_TM) + 256; }

// order[i] = i-th tile (of 64 output rows) to launch: heaviest cost class first, ties in tile order. Deterministic.
extern "C" int irx_tile_order(const int32_t* nbr, int ld, int n_out, int K, int32_t* order, void* workspace,
                              size_t workspace_bytes, void* stream) {
  IRX_REQUIRE(n_out >= 0 && K >= 1 && K <= 27 && ld >= n_out, "irx_tile_order: bad sizes");
  if (n_out == 0) return IRX_OK;
  IRX_REQUIRE(nbr && order && workspace && workspace_bytes >= irx_tile_order_workspace_bytes(n_out), "irx_tile_order: bad arguments");
  const int ntiles = irx_cdiv(n_out, TS_TM);
  // largest possible cost K * (FIX + 4 * GRP) mapped onto the classes
  int shift = 0;
  while (((K * (TS_FIX + 4 * TS_GRP)) >> shift) >= TS_NCLS) ++shift;
  unsigned char* cls = (unsigned char*)workspace;
  k_tile_cost<<<irx_cdiv(ntiles, 4), 256, 0, S(stream)>>>(nbr, ld, n_out, K, ntiles, shift, cls);
  k_tile_order<<<1, 64 * TO_WAVES, 0, S(stream)>>>(cls, ntiles, order);
  IRX_CHECK_LAUNCH("irx_tile_order");
  return IRX_OK;
}
