// irx_coords.hip — coordinate keys, hash-based voxelisation, neighbour-table ("rule") generation.
// All kernels are HBM/L2-latency bound integer work: coalesced 16-byte coordinate loads, one
// thread per (voxel[, offset]) probe, wave-wide ballot + mbcnt prefix for compaction.
#include <stdarg.h>
#include <string.h>
#include "irx_common.h"

// ---------------------------------------------------------------- error plumbing (host) ----
static thread_local char g_irx_err[512] = "";
void irx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_irx_err, sizeof(g_irx_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* irx_last_error(void) { return g_irx_err; }
extern "C" int irx_version(void) { return IRX_VERSION_MAJOR * 1000 + IRX_VERSION_MINOR; }

extern "C" int irx_device_props(int device, int* out8) {
  IRX_REQUIRE(out8 != nullptr, "irx_device_props: null output");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev) {
    irx_set_error("irx_device_props: no HIP device %d (count %d)", device, ndev);
    return IRX_ERR_NO_DEVICE;
  }
  hipDeviceProp_t p;
  IRX_CHECK_HIP(hipGetDeviceProperties(&p, device), "hipGetDeviceProperties");
  out8[0] = p.multiProcessorCount;
  out8[1] = p.warpSize;
  out8[2] = (int)p.maxSharedMemoryPerMultiProcessor;
  out8[3] = p.l2CacheSize;
  out8[4] = p.clockRate;
  int arch = 0;
  for (const char* c = p.gcnArchName; *c && *c != ':'; ++c)
    if (*c >= '0' && *c <= '9') arch = arch * 10 + (*c - '0');
  out8[5] = arch;
  out8[6] = (int)(p.totalGlobalMem >> 20);
  out8[7] = 0;
  return IRX_OK;
}

extern "C" size_t irx_hash_capacity(int n) {
  size_t cap = 64;
  while (cap < (size_t)2 * (size_t)(n > 0 ? n : 1)) cap <<= 1;
  return cap;
}

// --------------------------------------------------------------------------- kernels ------
__global__ void k_coords_to_keys(const int4* __restrict__ coords, int n, uint64_t* __restrict__ keys) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = coords[i];
  keys[i] = irx_make_key(c.x, c.y, c.z, c.w);
}

// a key holds its voxel: (x, y, z, batch) back out of the Morton interleave (rows of the sorted key array -> coordinate
// rows, a streaming pass instead of a random 16-byte gather through the sort permutation)
__global__ void k_keys_to_coords(const uint64_t* __restrict__ keys, int n, int4* __restrict__ coords) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t key = keys[i], m = key & 0xFFFFFFFFFFFFull;
  coords[i] = make_int4((int)irx_compact3(m) - IRX_COORD_BIAS, (int)irx_compact3(m >> 1) - IRX_COORD_BIAS,
                        (int)irx_compact3(m >> 2) - IRX_COORD_BIAS, (int)(key >> 48));
}

template <typename T>
__global__ void k_quantize(const T* __restrict__ xyz, const int32_t* __restrict__ batch, int n,
                           double vx, double vy, double vz, int4* __restrict__ coords,
                           uint64_t* __restrict__ keys) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // float64 division then floor — bit-identical to numpy's `np.floor(coords / voxel)`.
  double x = (double)xyz[3 * (size_t)i + 0], y = (double)xyz[3 * (size_t)i + 1],
         z = (double)xyz[3 * (size_t)i + 2];
  const double fx = floor(x / vx), fy = floor(y / vy), fz = floor(z / vz);
  int b = batch ? batch[i] : 0;
  // A key holds 16 biased bits per coordinate and 15 bits of batch index: anything outside would alias onto another
  // voxel's key (silently merged voxels, wrong kernel maps). Such a point gets the poison key instead; k_voxel_select
  // turns it into a NEGATIVE voxel count, which every consumer treats as empty and the host raises on.
  const bool ok = fx >= -(double)IRX_COORD_BIAS && fx < (double)IRX_COORD_BIAS && fy >= -(double)IRX_COORD_BIAS &&
                  fy < (double)IRX_COORD_BIAS && fz >= -(double)IRX_COORD_BIAS && fz < (double)IRX_COORD_BIAS &&
                  b >= 0 && b < 32768;          // (NaN coordinates fail the comparisons too)
  int cx = ok ? (int)fx : 0, cy = ok ? (int)fy : 0, cz = ok ? (int)fz : 0;
  if (coords) coords[i] = make_int4(cx, cy, cz, b);
  keys[i] = ok ? irx_make_key(cx, cy, cz, b) : IRX_POISON_KEY;
}

__global__ void k_fill_table(uint64_t* __restrict__ tk, int32_t* __restrict__ tv, size_t cap) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < cap; i += stride) {
    tk[i] = IRX_EMPTY_KEY;
    tv[i] = 0x7FFFFFFF;
  }
}

// insert with first-occurrence rule: value = min(point index) over points of that voxel.
__global__ void k_voxel_insert(const uint64_t* __restrict__ keys, int n, uint64_t* tk, int32_t* tv,
                               uint64_t mask) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t key = keys[i];
  uint64_t slot = irx_mix64(key) & mask;
  while (true) {
    unsigned long long prev =
        atomicCAS((unsigned long long*)&tk[slot], (unsigned long long)IRX_EMPTY_KEY,
                  (unsigned long long)key);
    if (prev == IRX_EMPTY_KEY || prev == key) {
      atomicMin(&tv[slot], i);
      return;
    }
    slot = (slot + 1) & mask;
  }
}

// A point wins when it is the first occurrence of its voxel. Wave-aggregated append:
// ballot -> one atomicAdd per wave -> lane offset by mbcnt (prefix popcount of lower lanes).
__global__ void k_voxel_select(const uint64_t* __restrict__ keys, int n,
                               const uint64_t* __restrict__ tk, const int32_t* __restrict__ tv,
                               uint64_t mask, int32_t* __restrict__ winners, int32_t* count) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool win = false;
  if (i < n) {
    const uint64_t key = keys[i];
    if (key == IRX_POISON_KEY) atomicOr(count, (int32_t)0x80000000);   // out-of-range point (k_quantize): count < 0
    else win = (irx_hash_lookup(tk, tv, mask, key) == i);
  }
  unsigned long long ballot = __ballot(win);
  if (ballot == 0ull) return;
  int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == (__ffsll((long long)ballot) - 1)) base = atomicAdd(count, __popcll(ballot)) & 0x7fffffff;   // (sign bit =
  base = __shfl(base, __ffsll((long long)ballot) - 1);                                                       //  range error)
  if (win) {
    unsigned long long lower = ballot & ((1ull << lane) - 1ull);
    winners[base + __popcll(lower)] = i;
  }
}

// ---- 27-neighbour tables ("rule generation") of Morton-sorted levels, round 6 -------------------------------------------------
// Rows are sorted by Morton key, so the 26 neighbours of a workgroup's 256 consecutive voxels almost always live a few hundred rows
// away: the workgroup stages the keys of rows [q0 - 896, q0 + 256 + 896) in an LDS hash table (key + window row per slot, 4096
// slots: 40 KB) and resolves every probe whose key falls inside the window's key range there —
// definitive both ways (present -> row, absent -> -1), ~2 ds_read_b64 instead of a dependent random walk through a 16-24 MB hash
// table in HBM. (First form: a binary search over the sorted offsets, 11 ds_read_b32 per probe — 286 LDS reads per voxel made the
// kernel LDS-bandwidth-bound at 225 us for the 845 k voxels of a B = 16 scene pyramid.) Only probes
// outside the window's key range (measured on 50 k-point scenes: ~20 % of the probes, 2-5 % of the neighbours that exist) go to
// the level's global hash table; a level of <= 2048 rows is its own window and never does. All 27 columns are written by the
// thread that owns the row: coalesced 4-byte stores per column, no mirrored scatter, no pre-fill (round 5's k_kmap_s1 probed 13
// offsets per voxel in the hash table and scattered the mirrored half: 40 launches and 567 us per step for the two pyramids).
// One launch covers EVERY level of a pyramid (block -> level through a prefix table in the kernel arguments).
#define KM_T 256
#define KM_W 896
#define KM_WIN (KM_T + 2 * KM_W)
#define KM_MAXLEV 8

struct KmLevels {
  int nlev;
  int n[KM_MAXLEV], stride[KM_MAXLEV], ld[KM_MAXLEV];
  int blk0[KM_MAXLEV + 1];                 // first workgroup of each level in the fused grid (fill / insert / kmap grids differ)
  const uint64_t* keys[KM_MAXLEV];         // may be NULL: keys are then derived from the coordinate rows
  const int4* coords[KM_MAXLEV];
  uint64_t* tk[KM_MAXLEV];
  int32_t* tv[KM_MAXLEV];
  uint64_t mask[KM_MAXLEV];
  int32_t* nbr[KM_MAXLEV];
};

__device__ __forceinline__ int km_level_of(const KmLevels& L, int& b) {
  int l = 0;
  while (l + 1 < L.nlev && b >= L.blk0[l + 1]) ++l;
  b -= L.blk0[l];
  return l;
}

__global__ __launch_bounds__(256) void k_tables_fill_multi(KmLevels L) {
  int b = blockIdx.x;
  const int l = km_level_of(L, b);
  const int nb = L.blk0[l + 1] - L.blk0[l];
  const size_t cap = (size_t)L.mask[l] + 1;
  uint64_t* tk = L.tk[l];
  int32_t* tv = L.tv[l];
  for (size_t i = (size_t)b * 256 + threadIdx.x; i < cap; i += (size_t)nb * 256) {
    tk[i] = IRX_EMPTY_KEY;
    tv[i] = 0x7FFFFFFF;
  }
}

__global__ __launch_bounds__(256) void k_insert_multi(KmLevels L) {
  int b = blockIdx.x;
  const int l = km_level_of(L, b);
  const int i = b * 256 + threadIdx.x;
  if (i >= L.n[l]) return;
  const uint64_t key = L.keys[l] ? L.keys[l][i] : ({ int4 c = L.coords[l][i]; irx_make_key(c.x, c.y, c.z, c.w); });
  uint64_t* tk = L.tk[l];
  const uint64_t mask = L.mask[l];
  uint64_t slot = irx_mix64(key) & mask;
  while (true) {
    unsigned long long prev = atomicCAS((unsigned long long*)&tk[slot], (unsigned long long)IRX_EMPTY_KEY, (unsigned long long)key);
    if (prev == IRX_EMPTY_KEY || prev == key) {
      atomicMin(&L.tv[l][slot], i);
      return;
    }
    slot = (slot + 1) & mask;
  }
}

// window hash in LDS: 4096 slots of (64-bit key, 16-bit window row); load factor <= 0.5
#define KM_SLOTS 4096
#define KM_EMPTY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ uint32_t km_hash(uint64_t k) {
  uint32_t d = (uint32_t)k ^ (uint32_t)(k >> 29) ^ (uint32_t)(k >> 47);
  d ^= d >> 15; d *= 0x2C1B3C6Du; d ^= d >> 12; d *= 0x297A2D39u; d ^= d >> 15;
  return d & (KM_SLOTS - 1);
}

__global__ __launch_bounds__(256) void k_kmap_win(KmLevels L) {
  __shared__ unsigned long long s_tab[KM_SLOTS];
  __shared__ unsigned short s_idx[KM_SLOTS];
  __shared__ int s_unsorted;
  int b = blockIdx.x;
  const int l = km_level_of(L, b);
  const int n = L.n[l], s = L.stride[l], ld = L.ld[l];
  const uint64_t* __restrict__ keys = L.keys[l];
  const int4* __restrict__ coords = L.coords[l];
  const int q0 = b * KM_T;
  int lo = q0 - KM_W, hi = q0 + KM_T + KM_W;
  if (n <= KM_WIN) { lo = 0; hi = n; }      // a small level is its own window: every probe is definitive, no global table involved
  if (lo < 0) lo = 0;
  if (hi > n) hi = n;
  const int cnt = hi - lo;
  uint64_t klo, khi;
  if (keys) { klo = keys[lo]; khi = keys[hi - 1]; }
  else {
    int4 c = coords[lo]; klo = irx_make_key(c.x, c.y, c.z, c.w);
    c = coords[hi - 1]; khi = irx_make_key(c.x, c.y, c.z, c.w);
  }
  if (threadIdx.x == 0) s_unsorted = 0;
#pragma unroll
  for (int i = threadIdx.x; i < KM_SLOTS; i += 256) s_tab[i] = KM_EMPTY;
  __syncthreads();
  for (int i = threadIdx.x; i < cnt; i += 256) {
    uint64_t k, prev = 0;
    if (keys) {
      k = keys[lo + i];
      if (i > 0) prev = keys[lo + i - 1];
    } else {
      const int4 c = coords[lo + i];
      k = irx_make_key(c.x, c.y, c.z, c.w);
      if (i > 0) { const int4 p = coords[lo + i - 1]; prev = irx_make_key(p.x, p.y, p.z, p.w); }
    }
    // rows that are not strictly ascending in Morton order (a caller of the one-level entry that did not sort): the window's key
    // range would mean nothing, so the whole workgroup resolves its probes in the global hash table, as round 5's kernel did
    if (i > 0 && k <= prev) s_unsorted = 1;
    uint32_t h = km_hash(k);
    while (atomicCAS(&s_tab[h], KM_EMPTY, (unsigned long long)k) != KM_EMPTY) h = (h + 1) & (KM_SLOTS - 1);
    s_idx[h] = (unsigned short)i;
  }
  __syncthreads();
  const int q = q0 + threadIdx.x;
  if (q >= n) return;
  const bool unsorted = s_unsorted != 0;
  const bool open_lo = lo == 0, open_hi = hi == n;
  const uint64_t* __restrict__ tk = L.tk[l];
  const int32_t* __restrict__ tv = L.tv[l];
  const uint64_t mask = L.mask[l];
  int32_t* __restrict__ nbr = L.nbr[l];
  const int4 c = coords[q];
  // the three spread coordinates per axis once: a probe's key is an OR of three of them
  uint64_t sx[3], sy[3], sz[3];
  bool okx[3], oky[3], okz[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int x = c.x + (d - 1) * s, y = c.y + (d - 1) * s, z = c.z + (d - 1) * s;
    okx[d] = x >= -IRX_COORD_BIAS && x < IRX_COORD_BIAS;
    oky[d] = y >= -IRX_COORD_BIAS && y < IRX_COORD_BIAS;
    okz[d] = z >= -IRX_COORD_BIAS && z < IRX_COORD_BIAS;
    sx[d] = irx_spread3((uint32_t)(x + IRX_COORD_BIAS));
    sy[d] = irx_spread3((uint32_t)(y + IRX_COORD_BIAS)) << 1;
    sz[d] = irx_spread3((uint32_t)(z + IRX_COORD_BIAS)) << 2;
  }
  const uint64_t kb = (uint64_t)(uint32_t)c.w << 48;
  // fully unrolled: 26 independent probe chains per thread (x fastest: odd kernel)
#pragma unroll
  for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int k = dx + 3 * dy + 9 * dz;
        int r = -1;
        if (k == 13) {
          r = q;
        } else if (okx[dx] && oky[dy] && okz[dz]) {
          const uint64_t nk = kb | ((sx[dx] | sy[dy] | sz[dz]) & 0xFFFFFFFFFFFFull);
          // definitive in LDS: inside the window's key range, or beyond an end of the level that the window reaches
          const bool inside = !unsorted && ((nk >= klo || open_lo) && (nk <= khi || open_hi));
          if (inside) {
            if (nk >= klo && nk <= khi) {
              uint32_t h = km_hash(nk);
              while (true) {
                const unsigned long long e = s_tab[h];
                if (e == nk) { r = lo + (int)s_idx[h]; break; }
                if (e == KM_EMPTY) break;
                h = (h + 1) & (KM_SLOTS - 1);
              }
            }
          } else {
            r = irx_hash_lookup(tk, tv, mask, nk);
          }
        }
        nbr[(size_t)k * ld + q] = r;
      }
    }
  }
}

// ---- octree descent: a level's 27-neighbour table from the NEXT COARSER level's table (round 6) ----------------------------------
// The pyramid irx_pyramid_build makes IS an octree: row q of level l has a parent row p at level l + 1 and a child slot koff
// (x * 4 + y * 2 + z of its position inside the parent's 2 x 2 x 2 cell). The voxel at q + d (d in {-1, 0, 1}^3, in units of the
// level's stride) lies in the parent cell P + D with D = floor((b + d) / 2) per axis (b = the child-position bit) at child position
// (b + d) mod 2 — so nbr_l[d][q] = child_l[(b + d) mod 2][ nbr_{l+1}[D][p] ]: two dependent reads of tables indexed by rows that are
// NEIGHBOURS IN MORTON ORDER (cache-resident), no hash table, no key arithmetic, and an absent parent neighbour answers all of its
// (up to 8) children at once. Each voxel needs 8 entries of the parent's table (2 values of D per axis) and <= 26 child-table
// reads; the coarsest level (a few thousand rows) comes from k_kmap_win. Measured against the windowed / hashed search at B = 16:
// see DESIGN.md section 12.
__global__ __launch_bounds__(256) void k_kmap_descend(const int32_t* __restrict__ parent, const uint8_t* __restrict__ koff, int n,
                                                      const int32_t* __restrict__ nbrc, int ldc, const int32_t* __restrict__ child,
                                                      int ldch, int32_t* __restrict__ nbr, int ld) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  const int p = parent[q];
  const int ko = koff[q];
  const int bx = (ko >> 2) & 1, by = (ko >> 1) & 1, bz = ko & 1;
  // the 8 parent-level cells this voxel's neighbourhood touches: per axis D in {b - 1, b}
  int pn[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const int Dx = bx - 1 + (a & 1), Dy = by - 1 + ((a >> 1) & 1), Dz = bz - 1 + ((a >> 2) & 1);
    const int Dk = (Dx + 1) + 3 * (Dy + 1) + 9 * (Dz + 1);
    pn[a] = (Dk == 13) ? p : nbrc[(size_t)Dk * ldc + p];
  }
#pragma unroll
  for (int dz = -1; dz <= 1; ++dz) {
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int k = (dx + 1) + 3 * (dy + 1) + 9 * (dz + 1);
        int r;
        if (k == 13) {
          r = q;
        } else {
          const int tx = bx + dx, ty = by + dy, tz = bz + dz;                       // -1 .. 2
          // index of the parent cell among the 8 fetched: a = D - (b - 1), D = floor(t / 2)
          const int ax = ((tx + 2) >> 1) - bx, ay = ((ty + 2) >> 1) - by, az = ((tz + 2) >> 1) - bz;     // ((t + 2) >> 1) - 1 = floor(t / 2)
          const int sel = ax | (ay << 1) | (az << 2);
          int pp = pn[0];
#pragma unroll
          for (int a = 1; a < 8; ++a) pp = (sel == a) ? pn[a] : pp;
          const int kc = ((tx & 1) << 2) | ((ty & 1) << 1) | (tz & 1);
          r = pp >= 0 ? child[(size_t)kc * ldch + pp] : -1;
        }
        nbr[(size_t)k * ld + q] = r;
      }
    }
  }
}

// ---- down-sampling by segmented scan over Morton-sorted keys ----------------------------
#define DS_BLOCK 256
#define DS_ITEMS 8
#define DS_TILE (DS_BLOCK * DS_ITEMS)

__device__ static inline uint64_t parent_key(uint64_t key, int level) {
  return key & ~(7ull << (3 * level));
}

// pass 1: per-tile count of segment heads (parent key differs from predecessor's).
__global__ void k_ds_count(const uint64_t* __restrict__ keys, int n, int level,
                           int32_t* __restrict__ tile_counts, const int32_t* __restrict__ n_dev) {
  __shared__ int s_wave[DS_BLOCK / 64];
  if (n_dev) n = *n_dev < 0 ? 0 : *n_dev;   // (negative = the voxeliser's range-error flag: an empty level)                           // row count produced on the device by the previous level
  int base = blockIdx.x * DS_TILE;
  int cnt = 0;
  for (int it = 0; it < DS_ITEMS; ++it) {
    int i = base + it * DS_BLOCK + threadIdx.x;
    if (i < n) {
      uint64_t pk = parent_key(keys[i], level);
      bool head = (i == 0) || (parent_key(keys[i - 1], level) != pk);
      cnt += head ? 1 : 0;
    }
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
  if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < DS_BLOCK / 64; ++w) t += s_wave[w];
    tile_counts[blockIdx.x] = t;
  }
}

// pass 2: every tile sums the counts of the tiles before it (<= a few hundred values), then an
// in-tile ballot/prefix scan assigns parent rows; heads write the parent's coord/key.
__global__ void k_ds_write(const uint64_t* __restrict__ keys, const int4* __restrict__ coords, int n,
                           int level, int s2, const int32_t* __restrict__ tile_counts, int ntiles,
                           int32_t* __restrict__ parent, uint8_t* __restrict__ koff,
                           int4* __restrict__ out_coords, uint64_t* __restrict__ out_keys,
                           int32_t* __restrict__ child, int ld, int32_t* __restrict__ n_out,
                           const int32_t* __restrict__ n_dev) {
  __shared__ int s_red[DS_BLOCK / 64];
  if (n_dev) n = *n_dev < 0 ? 0 : *n_dev;   // (negative = the voxeliser's range-error flag: an empty level)
  __shared__ int s_wave_base[DS_BLOCK / 64];
  __shared__ int s_tile_base;
  // exclusive prefix of tile counts for this tile
  int acc = 0;
  for (int t = threadIdx.x; t < (int)blockIdx.x; t += DS_BLOCK) acc += tile_counts[t];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < DS_BLOCK / 64; ++w) t += s_red[w];
    s_tile_base = t;
    if (blockIdx.x == (unsigned)ntiles - 1) *n_out = t + tile_counts[ntiles - 1];
  }
  __syncthreads();
  int running = s_tile_base;  // heads seen before the current DS_BLOCK-wide strip
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int base = blockIdx.x * DS_TILE;
  for (int it = 0; it < DS_ITEMS; ++it) {
    int i = base + it * DS_BLOCK + threadIdx.x;
    bool valid = i < n;
    uint64_t key = valid ? keys[i] : 0ull;
    uint64_t pk = parent_key(key, level);
    bool head = valid && ((i == 0) || (parent_key(keys[i - 1], level) != pk));
    unsigned long long ballot = __ballot(head);
    if (lane == 0) s_wave_base[wave] = __popcll(ballot);
    __syncthreads();
    int wbase = 0, total = 0;
    for (int w = 0; w < DS_BLOCK / 64; ++w) {
      int c = s_wave_base[w];
      if (w < wave) wbase += c;
      total += c;
    }
    // inclusive count of heads up to and including this lane, minus 1 = parent row
    unsigned long long upto = ballot & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull));
    int p = running + wbase + __popcll(upto) - 1;
    if (valid) {
      parent[i] = p;
      int bits = (int)((key >> (3 * level)) & 7ull);  // bit0 = x, bit1 = y, bit2 = z
      int k = ((bits & 1) << 2) | (bits & 2) | ((bits >> 2) & 1);  // even kernel: z fastest
      koff[i] = (uint8_t)k;
      child[(size_t)k * ld + p] = i;
      if (head) {
        int4 c = coords[i];
        int m = ~(s2 - 1);
        // floor(c / s2) * s2 for two's-complement ints and power-of-two s2
        out_coords[p] = make_int4(c.x & m, c.y & m, c.z & m, c.w);
        out_keys[p] = pk;
      }
    }
    running += total;
    __syncthreads();
  }
}

__global__ void k_fill_i32(int32_t* __restrict__ p, size_t n, int32_t v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

__global__ void k_down_transpose(const int32_t* __restrict__ parent, const uint8_t* __restrict__ koff,
                                 int n, int32_t* __restrict__ tbl, int ld) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int p = parent[i];
  int kk = koff[i];
#pragma unroll
  for (int k = 0; k < 8; ++k) tbl[(size_t)k * ld + i] = (k == kk) ? p : -1;
}

__global__ void k_bev_table(int s, int batch_size, int nx, int ny, int nz,
                            const uint64_t* __restrict__ tk, const int32_t* __restrict__ tv,
                            uint64_t mask, int32_t* __restrict__ tbl, int ld) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  int k = blockIdx.y;
  int ncell = batch_size * nx * ny;
  if (c >= ncell) return;
  int b = c / (nx * ny);
  int r = c - b * nx * ny;
  int ix = r / ny, iy = r - ix * ny;
  tbl[(size_t)k * ld + c] = irx_hash_lookup(tk, tv, mask, irx_make_key(ix * s, iy * s, k * s, b));
}

__global__ void k_bev_rows(const int4* __restrict__ coords, int n, int s, int batch_size, int nx,
                           int ny, int nz, int32_t* __restrict__ cell_of_row,
                           uint8_t* __restrict__ zbin_of_row) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = coords[i];
  bool in = c.x >= 0 && c.y >= 0 && c.z >= 0 && c.x < nx * s && c.y < ny * s && c.z < nz * s &&
            c.w >= 0 && c.w < batch_size;
  int ix = c.x / s, iy = c.y / s, iz = c.z / s;
  cell_of_row[i] = in ? (c.w * nx * ny + ix * ny + iy) : -1;
  zbin_of_row[i] = (uint8_t)(in ? iz : 0);
}

__global__ void k_batch_offsets(const int4* __restrict__ coords, int n, int nseg,
                                int32_t* __restrict__ offsets) {
  // offsets[b] = first row with batch >= b. One thread per row boundary.
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  int prev = (i == 0) ? -1 : coords[i - 1].w;
  int cur = (i == n) ? nseg : coords[i].w;
  if (cur > nseg) cur = nseg;
  for (int b = prev + 1; b <= cur; ++b) offsets[b] = i;
}

// ------------------------------------------------------------------------- C entry points --
static inline hipStream_t S(void* s) { return (hipStream_t)s; }
static inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

extern "C" int irx_coords_to_keys(const int32_t* coords, int n, uint64_t* keys, void* stream) {
  IRX_REQUIRE(n >= 0, "irx_coords_to_keys: n < 0");
  if (n == 0) return IRX_OK;
  IRX_REQUIRE(coords && keys, "irx_coords_to_keys: null pointer");
  k_coords_to_keys<<<irx_cdiv(n, 256), 256, 0, S(stream)>>>((const int4*)coords, n, keys);
  IRX_CHECK_LAUNCH("irx_coords_to_keys");
  return IRX_OK;
}

extern "C" int irx_keys_to_coords(const uint64_t* keys, int n, int32_t* coords, void* stream) {
  IRX_REQUIRE(n >= 0, "irx_keys_to_coords: n < 0");
  if (n == 0) return IRX_OK;
  IRX_REQUIRE(coords && keys, "irx_keys_to_coords: null pointer");
  k_keys_to_coords<<<irx_cdiv(n, 256), 256, 0, S(stream)>>>(keys, n, (int4*)coords);
  IRX_CHECK_LAUNCH("irx_keys_to_coords");
  return IRX_OK;
}

extern "C" int irx_quantize(const void* xyz, int xyz_is_f64, const int32_t* batch, int n, double vx,
                            double vy, double vz, int32_t* coords, uint64_t* keys, void* stream) {
  IRX_REQUIRE(n >= 0, "irx_quantize: n < 0");
  if (n == 0) return IRX_OK;
  IRX_REQUIRE(xyz && keys, "irx_quantize: null pointer");
  IRX_REQUIRE(vx > 0 && vy > 0 && vz > 0, "irx_quantize: voxel size must be > 0");
  if (xyz_is_f64)
    k_quantize<double><<<irx_cdiv(n, 256), 256, 0, S(stream)>>>((const double*)xyz, batch, n, vx, vy,
                                                               vz, (int4*)coords, keys);
  else
    k_quantize<float><<<irx_cdiv(n, 256), 256, 0, S(stream)>>>((const float*)xyz, batch, n, vx, vy,
                                                              vz, (int4*)coords, keys);
  IRX_CHECK_LAUNCH("irx_quantize");
  return IRX_OK;
}

static int check_table(const char* who, const void* tk, const void* tv, size_t cap, int n) {
  IRX_REQUIRE(tk && tv, "%s: null hash table", who);
  IRX_REQUIRE(cap >= 64 && (cap & (cap - 1)) == 0, "%s: capacity %zu is not a power of two", who, cap);
  IRX_REQUIRE(cap >= (size_t)n + (size_t)n / 2 || cap >= 2 * (size_t)n,
              "%s: capacity %zu too small for %d keys", who, cap, n);
  return IRX_OK;
}

extern "C" int irx_voxel_insert(const uint64_t* keys, int n, uint64_t* tk, int32_t* tv, size_t cap,
                                void* stream) {
  IRX_REQUIRE(n >= 0, "irx_voxel_insert: n < 0");
  int rc = check_table("irx_voxel_insert", tk, tv, cap, n);
  if (rc) return rc;
  int fb = irx_cdiv((long long)cap, 256);
  if (fb > 2048) fb = 2048;
  k_fill_table<<<fb, 256, 0, S(stream)>>>(tk, tv, cap);
  IRX_CHECK_LAUNCH("irx_voxel_insert(fill)");
  if (n == 0) return IRX_OK;
  IRX_REQUIRE(keys, "irx_voxel_insert: null keys");
  k_voxel_insert<<<irx_cdiv(n, 256), 256, 0, S(stream)>>>(keys, n, tk, tv, (uint64_t)cap - 1);
  IRX_CHECK_LAUNCH("irx_voxel_insert");
  return IRX_OK;
}

extern "C" int irx_voxel_select(const uint64_t* keys, int n, const uint64_t* tk, const int32_t* tv,
                                size_t cap, int32_t* winners, int32_t* count, void* stream) {
  IRX_REQUIRE(n >= 0 && count, "irx_voxel_select: bad arguments");
  IRX_CHECK_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), S(stream)), "irx_voxel_select(memset)");
  if (n == 0) return IRX_OK;
  int rc = check_table("irx_voxel_select", tk, tv, cap, n);
  if (rc) return rc;
  IRX_REQUIRE(keys && winners, "irx_voxel_select: null pointer");
  k_voxel_select<<<irx_cdiv(n, 256), 256, 0, S(stream)>>>(keys, n, tk, tv, (uint64_t)cap - 1, winners,
                                                         count);
  IRX_CHECK_LAUNCH("irx_voxel_select");
  return IRX_OK;
}

extern "C" int irx_hash_build(const uint64_t* keys, int n, uint64_t* tk, int32_t* tv, size_t cap,
                              void* stream) {
  IRX_REQUIRE(n >= 0, "irx_hash_build: n < 0");
  int rc = check_table("irx_hash_build", tk, tv, cap, n);
  if (rc) return rc;
  int fb = irx_cdiv((long long)cap, 256);
  if (fb > 2048) fb = 2048;
  k_fill_table<<<fb, 256, 0, S(stream)>>>(tk, tv, cap);
  IRX_CHECK_LAUNCH("irx_hash_build(fill)");
  if (n == 0) return IRX_OK;
  IRX_REQUIRE(keys, "irx_hash_build: null keys");
  // unique keys: the first-occurrence insert degenerates to key -> row
  k_voxel_insert<<<irx_cdiv(n, 256), 256, 0, S(stream)>>>(keys, n, tk, tv, (uint64_t)cap - 1);
  IRX_CHECK_LAUNCH("irx_hash_build");
  return IRX_OK;
}

static int km_launch(const KmLevels& base, bool build_tables, void* stream) {
  KmLevels L = base;
  if (build_tables) {
    int nb = 0;
    for (int l = 0; l < L.nlev; ++l) {
      L.blk0[l] = nb;
      int fb = irx_cdiv((long long)(L.mask[l] + 1), 256 * 8);       // 8 slots per thread
      nb += fb < 1 ? 1 : fb;
    }
    L.blk0[L.nlev] = nb;
    k_tables_fill_multi<<<nb, 256, 0, S(stream)>>>(L);
    IRX_CHECK_LAUNCH("irx_kmaps_build_multi(fill)");
    nb = 0;
    for (int l = 0; l < L.nlev; ++l) {
      L.blk0[l] = nb;
      nb += irx_cdiv(L.n[l], 256);
    }
    L.blk0[L.nlev] = nb;
    if (nb > 0) {
      k_insert_multi<<<nb, 256, 0, S(stream)>>>(L);
      IRX_CHECK_LAUNCH("irx_kmaps_build_multi(insert)");
    }
  }
  int nb = 0;
  for (int l = 0; l < L.nlev; ++l) {
    L.blk0[l] = nb;
    nb += irx_cdiv(L.n[l], KM_T);
  }
  L.blk0[L.nlev] = nb;
  if (nb > 0) {
    k_kmap_win<<<nb, 256, 0, S(stream)>>>(L);
    IRX_CHECK_LAUNCH("irx_kmaps_build_multi(kmap)");
  }
  return IRX_OK;
}

extern "C" int irx_kmap_build_s1(const int32_t* coords, int n, int tensor_stride, const uint64_t* tk,
                                 const int32_t* tv, size_t cap, int32_t* nbr, int ld, void* stream) {
  IRX_REQUIRE(n >= 0 && ld >= n, "irx_kmap_build_s1: bad n/ld");
  IRX_REQUIRE(tensor_stride >= 1 && (tensor_stride & (tensor_stride - 1)) == 0,
              "irx_kmap_build_s1: tensor stride %d is not a power of two", tensor_stride);
  if (n == 0) return IRX_OK;
  int rc = check_table("irx_kmap_build_s1", tk, tv, cap, n);
  if (rc) return rc;
  IRX_REQUIRE(coords && nbr, "irx_kmap_build_s1: null pointer");
  KmLevels L;
  memset(&L, 0, sizeof(L));
  L.nlev = 1;
  L.n[0] = n; L.stride[0] = tensor_stride; L.ld[0] = ld;
  L.keys[0] = nullptr;                         // derived from the coordinate rows (Morton order is the caller's contract)
  L.coords[0] = (const int4*)coords;
  L.tk[0] = (uint64_t*)tk; L.tv[0] = (int32_t*)tv; L.mask[0] = (uint64_t)cap - 1;
  L.nbr[0] = nbr;
  return km_launch(L, false, stream);
}

extern "C" int irx_kmaps_build_multi(int nlev, const uint64_t* const* keys, const int32_t* const* coords, const int* n,
                                     const int* tensor_stride, uint64_t* const* tk, int32_t* const* tv, const size_t* cap,
                                     int32_t* const* nbr, const int* ld, void* stream) {
  IRX_REQUIRE(nlev >= 0 && nlev <= KM_MAXLEV, "irx_kmaps_build_multi: %d levels (at most %d)", nlev, KM_MAXLEV);
  if (nlev == 0) return IRX_OK;
  IRX_REQUIRE(keys && coords && n && tensor_stride && tk && tv && cap && nbr && ld, "irx_kmaps_build_multi: null argument array");
  KmLevels L;
  memset(&L, 0, sizeof(L));
  int m = 0;
  for (int l = 0; l < nlev; ++l) {
    IRX_REQUIRE(n[l] >= 0 && ld[l] >= n[l], "irx_kmaps_build_multi: bad n / ld of level %d", l);
    IRX_REQUIRE(tensor_stride[l] >= 1 && (tensor_stride[l] & (tensor_stride[l] - 1)) == 0,
                "irx_kmaps_build_multi: tensor stride %d is not a power of two", tensor_stride[l]);
    if (n[l] == 0) continue;                   // an empty level has no tables
    int rc = check_table("irx_kmaps_build_multi", tk[l], tv[l], cap[l], n[l]);
    if (rc) return rc;
    IRX_REQUIRE(keys[l] && coords[l] && nbr[l], "irx_kmaps_build_multi: null pointer at level %d", l);
    L.n[m] = n[l]; L.stride[m] = tensor_stride[l]; L.ld[m] = ld[l];
    L.keys[m] = keys[l]; L.coords[m] = (const int4*)coords[l];
    L.tk[m] = tk[l]; L.tv[m] = tv[l]; L.mask[m] = (uint64_t)cap[l] - 1;
    L.nbr[m] = nbr[l];
    ++m;
  }
  L.nlev = m;
  if (m == 0) return IRX_OK;
  return km_launch(L, true, stream);
}

extern "C" int irx_kmaps_build_pyramid(int nlev, const int* n, const int* tensor_stride, const uint64_t* keys_top,
                                       const int32_t* coords_top, uint64_t* tk_top, int32_t* tv_top, size_t cap_top,
                                       const int32_t* const* parent, const uint8_t* const* koff, const int32_t* const* child,
                                       const int* child_ld, int32_t* const* nbr, const int* ld, void* stream) {
  IRX_REQUIRE(nlev >= 1 && nlev <= KM_MAXLEV, "irx_kmaps_build_pyramid: %d levels (1 .. %d)", nlev, KM_MAXLEV);
  IRX_REQUIRE(n && tensor_stride && nbr && ld, "irx_kmaps_build_pyramid: null argument array");
  IRX_REQUIRE(nlev == 1 || (parent && koff && child && child_ld), "irx_kmaps_build_pyramid: null map arrays");
  for (int l = 0; l < nlev; ++l) {
    IRX_REQUIRE(n[l] >= 0 && ld[l] >= n[l] && (n[l] == 0 || nbr[l]), "irx_kmaps_build_pyramid: bad n / ld / table of level %d", l);
    IRX_REQUIRE(l == 0 || n[l] <= n[l - 1], "irx_kmaps_build_pyramid: level %d has more rows than the finer level", l);
  }
  const int top = nlev - 1;
  if (n[top] > 0) {
    IRX_REQUIRE(coords_top, "irx_kmaps_build_pyramid: null coordinates of the coarsest level");
    KmLevels L;
    memset(&L, 0, sizeof(L));
    L.nlev = 1;
    L.n[0] = n[top]; L.stride[0] = tensor_stride[top]; L.ld[0] = ld[top];
    L.keys[0] = keys_top; L.coords[0] = (const int4*)coords_top;
    L.nbr[0] = nbr[top];
    const bool need_table = n[top] > KM_WIN;           // a level of <= KM_WIN rows is its own window: no hash table involved
    if (need_table) {
      int rc = check_table("irx_kmaps_build_pyramid", tk_top, tv_top, cap_top, n[top]);
      if (rc) return rc;
      L.tk[0] = tk_top; L.tv[0] = tv_top; L.mask[0] = (uint64_t)cap_top - 1;
    }
    int rc = km_launch(L, need_table, stream);
    if (rc) return rc;
  }
  for (int l = top - 1; l >= 0; --l) {
    if (n[l] == 0) continue;
    IRX_REQUIRE(parent[l] && koff[l] && child[l] && child_ld[l] >= n[l + 1] && n[l + 1] > 0,
                "irx_kmaps_build_pyramid: bad down-sampling map of level %d", l);
    k_kmap_descend<<<irx_cdiv(n[l], 256), 256, 0, S(stream)>>>(parent[l], koff[l], n[l], nbr[l + 1], ld[l + 1], child[l], child_ld[l],
                                                               nbr[l], ld[l]);
    IRX_CHECK_LAUNCH("irx_kmaps_build_pyramid(descend)");
  }
  return IRX_OK;
}

extern "C" size_t irx_downsample_workspace_bytes(int n) {
  return (size_t)(irx_cdiv(n > 0 ? n : 1, DS_TILE)) * sizeof(int32_t) + 64;
}

extern "C" int irx_downsample(const uint64_t* keys, const int32_t* coords, int n, int tensor_stride,
                              int32_t* parent, uint8_t* koff, int32_t* out_coords, uint64_t* out_keys,
                              int32_t* child, int ld, int32_t* n_out, void* workspace,
                              size_t workspace_bytes, void* stream) {
  IRX_REQUIRE(n >= 0 && n_out, "irx_downsample: bad arguments");
  IRX_REQUIRE(tensor_stride >= 1 && (tensor_stride & (tensor_stride - 1)) == 0 && tensor_stride <= 8192,
              "irx_downsample: tensor stride %d unsupported", tensor_stride);
  if (n == 0) {
    IRX_CHECK_HIP(hipMemsetAsync(n_out, 0, sizeof(int32_t), S(stream)), "irx_downsample(memset)");
    return IRX_OK;
  }
  IRX_REQUIRE(keys && coords && parent && koff && out_coords && out_keys && child,
              "irx_downsample: null pointer");
  IRX_REQUIRE(ld >= 1, "irx_downsample: ld < 1");
  if (workspace_bytes < irx_downsample_workspace_bytes(n) || !workspace) {
    irx_set_error("irx_downsample: workspace %zu < %zu", workspace_bytes,
                  irx_downsample_workspace_bytes(n));
    return IRX_ERR_WORKSPACE;
  }
  int ntiles = irx_cdiv(n, DS_TILE);
  int level = ilog2(tensor_stride);
  int32_t* tile_counts = (int32_t*)workspace;
  size_t fill = (size_t)8 * ld;
  int fb = irx_cdiv((long long)fill, 256);
  if (fb > 2048) fb = 2048;
  k_fill_i32<<<fb, 256, 0, S(stream)>>>(child, fill, -1);
  IRX_CHECK_LAUNCH("irx_downsample(fill)");
  k_ds_count<<<ntiles, DS_BLOCK, 0, S(stream)>>>(keys, n, level, tile_counts, nullptr);
  IRX_CHECK_LAUNCH("irx_downsample(count)");
  k_ds_write<<<ntiles, DS_BLOCK, 0, S(stream)>>>(keys, (const int4*)coords, n, level,
                                                2 * tensor_stride, tile_counts, ntiles, parent, koff,
                                                (int4*)out_coords, out_keys, child, ld, n_out, nullptr);
  IRX_CHECK_LAUNCH("irx_downsample(write)");
  return IRX_OK;
}

// child[k][p] = -1 for p < n (n on the device when n_dev != NULL): only the columns a level can use are touched
__global__ void k_fill_child(int32_t* __restrict__ child, int ld, int n, const int32_t* __restrict__ n_dev) {
  if (n_dev) n = *n_dev < 0 ? 0 : *n_dev;   // (negative = the voxeliser's range-error flag: an empty level)
  const size_t total = (size_t)8 * n;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) child[(i / n) * ld + (i % n)] = -1;
}

// The whole `levels`-deep pyramid of one tensor in ONE call and with NO host sync in between: level l reads its row
// count from counts[l-1] on the device (grids are sized for the upper bound n0; surplus workgroups find nothing to
// do), so the caller needs a single D2H copy of counts[] instead of one per level (each cost ~0.12 ms of blocked
// host time in the training loop). n0_dev != NULL: the finest level's own row count is on the device too (a voxeliser
// that has not been synchronised yet) and n0 is only its upper bound. Every per-level buffer is caller-allocated for n0 rows; child tables have ld = ld.
extern "C" int irx_pyramid_build(const uint64_t* keys0, const int32_t* coords0, int n0, int stride0, int levels,
                                 int32_t* const* parent, uint8_t* const* koff, int32_t* const* out_coords,
                                 uint64_t* const* out_keys, int32_t* const* child, int ld, int32_t* counts,
                                 const int32_t* n0_dev, void* workspace, size_t workspace_bytes, void* stream) {
  IRX_REQUIRE(n0 >= 0 && levels >= 1 && levels <= 8 && counts, "irx_pyramid_build: bad arguments");
  IRX_REQUIRE(stride0 >= 1 && (stride0 & (stride0 - 1)) == 0 && (stride0 << levels) <= 16384,
              "irx_pyramid_build: tensor stride %d unsupported", stride0);
  if (n0 == 0) {
    IRX_CHECK_HIP(hipMemsetAsync(counts, 0, levels * sizeof(int32_t), S(stream)), "irx_pyramid_build(memset)");
    return IRX_OK;
  }
  IRX_REQUIRE(keys0 && coords0 && parent && koff && out_coords && out_keys && child, "irx_pyramid_build: null pointer");
  IRX_REQUIRE(ld >= n0, "irx_pyramid_build: ld %d < n0 %d", ld, n0);
  if (workspace_bytes < irx_downsample_workspace_bytes(n0) || !workspace) {
    irx_set_error("irx_pyramid_build: workspace %zu < %zu", workspace_bytes, irx_downsample_workspace_bytes(n0));
    return IRX_ERR_WORKSPACE;
  }
  const int ntiles = irx_cdiv(n0, DS_TILE);
  int32_t* tile_counts = (int32_t*)workspace;
  const uint64_t* keys = keys0;
  const int32_t* coords = coords0;
  int stride = stride0;
  for (int l = 0; l < levels; ++l) {
    IRX_REQUIRE(parent[l] && koff[l] && out_coords[l] && out_keys[l] && child[l], "irx_pyramid_build: null level %d", l);
    const int32_t* n_dev = l ? counts + (l - 1) : n0_dev;   // level 0: host n0, or (n0_dev != NULL) n0 = upper bound
    int fb = irx_cdiv((long long)8 * n0, 256);
    if (fb > 2048) fb = 2048;
    k_fill_child<<<fb, 256, 0, S(stream)>>>(child[l], ld, n0, n_dev);
    IRX_CHECK_LAUNCH("irx_pyramid_build(fill)");
    k_ds_count<<<ntiles, DS_BLOCK, 0, S(stream)>>>(keys, n0, ilog2(stride), tile_counts, n_dev);
    IRX_CHECK_LAUNCH("irx_pyramid_build(count)");
    k_ds_write<<<ntiles, DS_BLOCK, 0, S(stream)>>>(keys, (const int4*)coords, n0, ilog2(stride), 2 * stride, tile_counts,
                                                  ntiles, parent[l], koff[l], (int4*)out_coords[l], out_keys[l],
                                                  child[l], ld, counts + l, n_dev);
    IRX_CHECK_LAUNCH("irx_pyramid_build(write)");
    keys = out_keys[l];
    coords = out_coords[l];
    stride *= 2;
  }
  return IRX_OK;
}

extern "C" int irx_kmap_down_transpose(const int32_t* parent, const uint8_t* koff, int n, int32_t* tbl,
                                       int ld, void* stream) {
  IRX_REQUIRE(n >= 0 && ld >= n, "irx_kmap_down_transpose: bad n/ld");
  if (n == 0) return IRX_OK;
  IRX_REQUIRE(parent && koff && tbl, "irx_kmap_down_transpose: null pointer");
  k_down_transpose<<<irx_cdiv(n, 256), 256, 0, S(stream)>>>(parent, koff, n, tbl, ld);
  IRX_CHECK_LAUNCH("irx_kmap_down_transpose");
  return IRX_OK;
}

extern "C" int irx_bev_table(const int32_t* coords, int n, int tensor_stride, int batch_size, int nx,
                             int ny, int nz, const uint64_t* tk, const int32_t* tv, size_t cap,
                             int32_t* tbl, int ld, int32_t* cell_of_row, uint8_t* zbin_of_row,
                             void* stream) {
  IRX_REQUIRE(n >= 0 && batch_size >= 1 && nx >= 1 && ny >= 1 && nz >= 1 && nz <= 255,
              "irx_bev_table: bad sizes");
  int ncell = batch_size * nx * ny;
  IRX_REQUIRE(ld >= ncell, "irx_bev_table: ld %d < cells %d", ld, ncell);
  int rc = check_table("irx_bev_table", tk, tv, cap, n);
  if (rc) return rc;
  IRX_REQUIRE(tbl, "irx_bev_table: null table");
  dim3 grid(irx_cdiv(ncell, 256), nz);
  k_bev_table<<<grid, 256, 0, S(stream)>>>(tensor_stride, batch_size, nx, ny, nz, tk, tv,
                                          (uint64_t)cap - 1, tbl, ld);
  IRX_CHECK_LAUNCH("irx_bev_table");
  if (n > 0 && cell_of_row && zbin_of_row) {
    IRX_REQUIRE(coords, "irx_bev_table: null coords");
    k_bev_rows<<<irx_cdiv(n, 256), 256, 0, S(stream)>>>((const int4*)coords, n, tensor_stride,
                                                       batch_size, nx, ny, nz, cell_of_row,
                                                       zbin_of_row);
    IRX_CHECK_LAUNCH("irx_bev_table(rows)");
  }
  return IRX_OK;
}

extern "C" int irx_batch_offsets(const int32_t* coords, int n, int nseg, int32_t* offsets,
                                 void* stream) {
  IRX_REQUIRE(n >= 0 && nseg >= 0 && offsets, "irx_batch_offsets: bad arguments");
  IRX_REQUIRE(n == 0 || coords, "irx_batch_offsets: null coords");
  k_batch_offsets<<<irx_cdiv(n + 1, 256), 256, 0, S(stream)>>>((const int4*)coords, n, nseg, offsets);
  IRX_CHECK_LAUNCH("irx_batch_offsets");
  return IRX_OK;
}
