// irx_sort.hip — stable LSD radix sort of 64-bit keys (Morton keys of the voxeliser, 16-bit slot ids of the input
// pipeline) with the permutation it applies: the ordering step inside "hash-based voxelisation".
// Replaces what torchsparse does with `np.unique(key, return_index=True)` on the host (sorted unique hashes: reference
// models/attribute_module.py:65-69, lib/dataset.py:229-233,256-260 reach it through sparse_quantize) — the build keys
// voxels by Morton code, so the sort IS the row order every later kernel relies on.
//
// 8-bit digits, three launches per pass:
//   k_rs_count   : workgroup = 4 x per_wave consecutive elements, wave w owns the w-th quarter of them; per 64-element chunk the
//                  lanes holding the same digit find each other with 8 ballots (one per digit bit) and the lowest lane of
//                  each peer group adds the group size to the wave's LDS counter -> counts[bin][4 * block + wave]
//   k_rs_rowscan : one workgroup per digit value: exclusive scan of its row of the table + the bin total (the scatter
//                  kernel scans the 256 totals itself) — a single-workgroup scan of the whole table was 200 us per pass
//   k_rs_scatter : same walk; position = base[digit] + (peers below me); stable because waves, chunks and lanes are
//                  visited in element order and the table is scanned bin-major, (block, wave)-minor
// No atomics on global memory, no float: the result is deterministic and bit-exact by construction.
#include "irx_common.h"

#define RS_BITS 8
#define RS_BINS 256
// elements per wave (a workgroup of 4 waves sorts 4 x this many consecutive elements). Round 6: sized from n instead of 1024 — at
// 1024 the scene's 490 k keys were 120 workgroups on 256 CUs, one 4-wave workgroup per CU walking 16 chunks in sequence
// (IRX_SORT_PER_WAVE: dev A/B, a multiple of 64)
static int rs_per_wave(int n) {
  static const int forced = [] {
    const char* e = getenv("IRX_SORT_PER_WAVE");
    int x = e ? atoi(e) : 0;
    return x > 0 ? (x < 64 ? 64 : (x + 63) / 64 * 64) : 0;
  }();
  if (forced) return forced;
  // 120-400 workgroups: fewer leave CUs idle, more make the 256 x (4 x workgroups) digit table the larger stream. Measured (us per
  // 7-pass sort, tools/micro/sort_bench.py) at 1024 / 512 / 256 / 128 / 64 elements per wave: 61 k keys 218 / 147 / 113 / 99 / 103,
  // 489 k keys 240 / 182 / 194 / 253 / 484, 800 k keys 262 / 236 / 294 / 496 / 733
  return n <= 150000 ? 128 : (n <= 300000 ? 256 : 512);
}

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

// lanes of this wave whose `digit` equals mine (including me); inactive lanes (valid == false) match nobody
__device__ __forceinline__ unsigned long long rs_peers(unsigned digit, bool valid) {
  unsigned long long m = __ballot(valid);
#pragma unroll
  for (int b = 0; b < RS_BITS; ++b) {
    const unsigned long long vote = __ballot((digit >> b) & 1u);
    m &= ((digit >> b) & 1u) ? vote : ~vote;
  }
  return valid ? m : 0ull;
}

// n_dev (optional): the number of meaningful elements lives on the device (sync-free voxeliser); elements at and beyond
// it take the key `pad` (which sorts behind every real key) so that they end up at the tail in their input order
__device__ __forceinline__ uint64_t rs_key(const uint64_t* __restrict__ keys, int i, int n_real, uint64_t pad) {
  return i < n_real ? keys[i] : pad;
}

__global__ __launch_bounds__(256) void k_rs_count(const uint64_t* __restrict__ keys, int n, const int32_t* __restrict__ n_dev,
                                                  uint64_t pad, int shift, int nslots, int32_t* __restrict__ counts, int per_wave) {
  __shared__ int hist[4][RS_BINS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 4 * RS_BINS; i += 256) (&hist[0][0])[i] = 0;
  __syncthreads();
  const int n_real = n_dev ? (*n_dev < n ? (*n_dev < 0 ? 0 : *n_dev) : n) : n;
  const int base = (blockIdx.x * 4 + wave) * per_wave;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int c = 0; c < per_wave; c += 64) {
    const int i = base + c + lane;
    const bool valid = i < n;
    const unsigned digit = valid ? (unsigned)((rs_key(keys, i, n_real, pad) >> shift) & (RS_BINS - 1)) : 0u;
    const unsigned long long peers = rs_peers(digit, valid);
    if (valid && (peers & lt) == 0ull) hist[wave][digit] += __popcll(peers);     // lowest lane of the group
  }
  __syncthreads();
  for (int i = tid; i < 4 * RS_BINS; i += 256) {
    const int w = i / RS_BINS, bin = i % RS_BINS;
    counts[(size_t)bin * nslots + 4 * blockIdx.x + w] = hist[w][bin];
  }
}

// One workgroup per digit value: exclusive scan of that bin's row counts[bin][0 .. nslots) in place (coalesced loads, LDS
// scan in chunks of 1024) and the bin's total -> totals[bin]. The scatter kernel adds the exclusive scan over the 256 totals.
__global__ __launch_bounds__(1024) void k_rs_rowscan(int32_t* __restrict__ counts, int nslots, int32_t* __restrict__ totals) {
  __shared__ int buf[1024];
  __shared__ int carry_s;
  int32_t* row = counts + (size_t)blockIdx.x * nslots;
  const int tid = threadIdx.x;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nslots; base += 1024) {
    const int i = base + tid;
    const int v = (i < nslots) ? row[i] : 0;
    buf[tid] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {         // Hillis-Steele inclusive scan
      const int t = (tid >= off) ? buf[tid - off] : 0;
      __syncthreads();
      buf[tid] += t;
      __syncthreads();
    }
    const int carry = carry_s;
    if (i < nslots) row[i] = carry + buf[tid] - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + buf[1023];
    __syncthreads();
  }
  if (tid == 0) totals[blockIdx.x] = carry_s;
}

// idx_in == NULL: the incoming order is the identity (first pass)
__global__ __launch_bounds__(256) void k_rs_scatter(const uint64_t* __restrict__ keys, const int32_t* __restrict__ idx_in,
                                                    int n, const int32_t* __restrict__ n_dev, uint64_t pad, int shift,
                                                    int nslots, const int32_t* __restrict__ offsets,
                                                    const int32_t* __restrict__ totals,
                                                    uint64_t* __restrict__ keys_out, int32_t* __restrict__ idx_out, int per_wave) {
  __shared__ int base_s[4][RS_BINS];
  __shared__ int bin_base[RS_BINS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {  // exclusive scan over the 256 bin totals (thread == bin)
    const int v = totals[tid];
    bin_base[tid] = v;
    __syncthreads();
    for (int off = 1; off < RS_BINS; off <<= 1) {
      const int t = (tid >= off) ? bin_base[tid - off] : 0;
      __syncthreads();
      bin_base[tid] += t;
      __syncthreads();
    }
    const int ex = bin_base[tid] - v;
    __syncthreads();
    bin_base[tid] = ex;
    __syncthreads();
  }
  for (int i = tid; i < 4 * RS_BINS; i += 256) {
    const int w = i / RS_BINS, bin = i % RS_BINS;
    base_s[w][bin] = bin_base[bin] + offsets[(size_t)bin * nslots + 4 * blockIdx.x + w];
  }
  __syncthreads();
  const int n_real = n_dev ? (*n_dev < n ? (*n_dev < 0 ? 0 : *n_dev) : n) : n;
  const int base = (blockIdx.x * 4 + wave) * per_wave;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int c = 0; c < per_wave; c += 64) {
    const int i = base + c + lane;
    const bool valid = i < n;
    const uint64_t key = valid ? rs_key(keys, i, n_real, pad) : 0ull;
    const unsigned digit = (unsigned)((key >> shift) & (RS_BINS - 1));
    const unsigned long long peers = rs_peers(digit, valid);
    int pos = 0;
    if (valid) pos = base_s[wave][digit] + __popcll(peers & lt);
    // every lane of a group read the counter before its lowest lane moves it on (wave-synchronous: one instruction stream)
    __builtin_amdgcn_wave_barrier();
    if (valid && (peers & lt) == 0ull) base_s[wave][digit] += __popcll(peers);
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      keys_out[pos] = key;
      idx_out[pos] = idx_in ? idx_in[i] : i;
    }
  }
}

static inline int rs_passes(int begin_bit, int end_bit) { return (end_bit - begin_bit + RS_BITS - 1) / RS_BITS; }

extern "C" size_t irx_sort_workspace_bytes(int n) {
  if (n <= 0) return 256;
  const size_t nslots = 4 * (size_t)irx_cdiv(n, 4 * rs_per_wave(n));
  size_t b = RS_BINS * nslots * sizeof(int32_t);       // digit counts / offsets
  b = (b + 255) & ~(size_t)255;
  b += 1024;                                           // bin totals
  b += (((size_t)n * sizeof(uint64_t) + 255) & ~(size_t)255) + (((size_t)n * sizeof(int32_t) + 255) & ~(size_t)255);   // ping-pong
  return b + 256;
}

// Stable ascending sort of keys[0, n) by their bits [begin_bit, end_bit): keys_out[i] = keys[order_out[i]].
// n_dev != NULL: only the first *n_dev elements are real (device-side count, no host sync); the others are treated as
// the key `pad` (must compare above every real key within the sorted bits) and land behind them.
extern "C" int irx_sort_pairs_u64(const uint64_t* keys, int n, const int32_t* n_dev, uint64_t pad, int begin_bit, int end_bit,
                                  uint64_t* keys_out, int32_t* order_out, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  IRX_REQUIRE(n >= 0 && begin_bit >= 0 && end_bit <= 64 && begin_bit < end_bit, "irx_sort_pairs_u64: bad arguments");
  if (n == 0) return IRX_OK;
  IRX_REQUIRE(keys && keys_out && order_out && keys != keys_out, "irx_sort_pairs_u64: null or aliased pointer");
  IRX_REQUIRE(workspace && workspace_bytes >= irx_sort_workspace_bytes(n), "irx_sort_pairs_u64: workspace %zu < %zu",
              workspace_bytes, irx_sort_workspace_bytes(n));
  const int pw = rs_per_wave(n);
  const int nblk = irx_cdiv(n, 4 * pw), nslots = 4 * nblk;
  char* p = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  int32_t* counts = (int32_t*)p;
  p += ((size_t)RS_BINS * nslots * sizeof(int32_t) + 255) & ~(size_t)255;
  int32_t* totals = (int32_t*)p;
  p += 1024;
  uint64_t* kb = (uint64_t*)p;
  p += ((size_t)n * sizeof(uint64_t) + 255) & ~(size_t)255;
  int32_t* ib = (int32_t*)p;
  const int passes = rs_passes(begin_bit, end_bit);
  // ping-pong so that the LAST pass writes the caller's output buffers
  const uint64_t* kin = keys;
  const int32_t* iin = nullptr;
  for (int ps = 0; ps < passes; ++ps) {
    const bool to_out = ((passes - 1 - ps) % 2) == 0;
    uint64_t* ko = to_out ? keys_out : kb;
    int32_t* io = to_out ? order_out : ib;
    const int shift = begin_bit + ps * RS_BITS;
    // after the first pass the padding has been materialised into the key buffer: no device count needed any more
    const int32_t* nd = (ps == 0) ? n_dev : nullptr;
    k_rs_count<<<nblk, 256, 0, S(stream)>>>(kin, n, nd, pad, shift, nslots, counts, pw);
    k_rs_rowscan<<<RS_BINS, 1024, 0, S(stream)>>>(counts, nslots, totals);
    k_rs_scatter<<<nblk, 256, 0, S(stream)>>>(kin, iin, n, nd, pad, shift, nslots, counts, totals, ko, io, pw);
    kin = ko;
    iin = io;
  }
  IRX_CHECK_LAUNCH("irx_sort_pairs_u64");
  return IRX_OK;
}
