// irx_input.hip — device side of the per-sample input pipeline (reference lib/dataset.py:124,154-181,207-232):
// sub-sample a resident scan, apply the augmentation transform, split it into instances (bounding box + fixed-size
// resample). All of it is row gathers and min/max reductions: HBM/latency-bound, no arithmetic beyond the 3x3
// rotations. Element type follows the scan on disk (float32 for ScanNet `_aligned_vert.npy`; float64 supported).
#include "irx_common.h"

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

struct IrxAug {
  int flip_x, flip_y, n_rot, has_shift;
  double rot[3][9];   // applied in order; row-major R, new = R * p (numpy: dot(p, R.T))
  double shift[3];
};

// dst[r][:] = src[choices[r]][:], xyz (first three columns) transformed. Every assignment of the reference rounds
// to the storage type (the numpy array is modified in place), so does this.
template <typename T>
__global__ __launch_bounds__(256) void k_scene_sample(const T* __restrict__ src, const int32_t* __restrict__ choices,
                                                      int n, int c, IrxAug aug, T* __restrict__ dst) {
#pragma clang fp contract(off)
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const T* s = src + (size_t)choices[r] * c;
  T* d = dst + (size_t)r * c;
  T x = s[0], y = s[1], z = s[2];
  if (aug.flip_x) x = -x;
  if (aug.flip_y) y = -y;
  for (int m = 0; m < aug.n_rot; ++m) {
    const double* R = aug.rot[m];
    const double px = (double)x, py = (double)y, pz = (double)z;
    x = (T)((px * R[0] + py * R[1]) + pz * R[2]);
    y = (T)((px * R[3] + py * R[4]) + pz * R[5]);
    z = (T)((px * R[6] + py * R[7]) + pz * R[8]);
  }
  if (aug.has_shift) {
    x = (T)((double)x + aug.shift[0]);
    y = (T)((double)y + aug.shift[1]);
    z = (T)((double)z + aug.shift[2]);
  }
  d[0] = x;
  d[1] = y;
  d[2] = z;
  for (int j = 3; j < c; ++j) d[j] = s[j];
}

// One workgroup per instance (blockIdx.x < n_inst) + one for the whole cloud (blockIdx.x == n_inst, if extent != 0):
// min / max of xyz over the segment's rows (exact, order-free), then centre = 0.5 * (lo + hi) and size = hi - lo in
// the storage type T like numpy does, widened to float64 (np.concatenate with an int array, lib/dataset.py:223).
template <typename T>
__global__ __launch_bounds__(256) void k_instance_box(const T* __restrict__ pts, int n, int c,
                                                      const int32_t* __restrict__ order,
                                                      const int32_t* __restrict__ seg, int n_inst,
                                                      double* __restrict__ obbs, T* __restrict__ extent) {
#pragma clang fp contract(off)
  __shared__ T slo[3][256], shi[3][256];
  const int i = blockIdx.x;
  const bool whole = (i == n_inst);
  const int beg = whole ? 0 : seg[i], end = whole ? n : seg[i + 1];
  if (end <= beg) {                                  // empty segment (an instance that lost all its points): zeros
    if (!whole && threadIdx.x < 7) obbs[(size_t)i * 7 + threadIdx.x] = 0.0;
    return;                                          // block-uniform
  }
  T lo[3], hi[3];
  bool any = false;
  for (int p = beg + threadIdx.x; p < end; p += 256) {
    const T* row = pts + (size_t)(whole ? p : order[p]) * c;
    for (int a = 0; a < 3; ++a) {
      const T v = row[a];
      lo[a] = any ? (v < lo[a] ? v : lo[a]) : v;
      hi[a] = any ? (v > hi[a] ? v : hi[a]) : v;
    }
    any = true;
  }
  // threads without rows take the segment's first row
  if (!any) {
    const T* row = pts + (size_t)(whole ? beg : order[beg]) * c;
    for (int a = 0; a < 3; ++a) lo[a] = hi[a] = row[a];
  }
  for (int a = 0; a < 3; ++a) {
    slo[a][threadIdx.x] = lo[a];
    shi[a][threadIdx.x] = hi[a];
  }
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
      for (int a = 0; a < 3; ++a) {
        const T l2 = slo[a][threadIdx.x + w], h2 = shi[a][threadIdx.x + w];
        if (l2 < slo[a][threadIdx.x]) slo[a][threadIdx.x] = l2;
        if (h2 > shi[a][threadIdx.x]) shi[a][threadIdx.x] = h2;
      }
    __syncthreads();
  }
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    const T l = slo[a][0], h = shi[a][0];
    if (whole) {
      extent[a] = l;
      extent[3 + a] = h;
    } else {
      obbs[(size_t)i * 7 + a] = (double)((T)0.5 * (T)(l + h));
      obbs[(size_t)i * 7 + 3 + a] = (double)(T)(h - l);
      if (a == 0) obbs[(size_t)i * 7 + 6] = 0.0;
    }
  }
}

// out[i][s][:] = pts[rows[i][s]][:]  (the 1024-point resample; rows index the sampled cloud)
template <typename T>
__global__ __launch_bounds__(256) void k_instance_gather(const T* __restrict__ pts, int c,
                                                         const int32_t* __restrict__ rows, size_t total,
                                                         T* __restrict__ out) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per output element
  if (e >= total) return;
  const size_t r = e / c;
  const int j = (int)(e % c);
  out[e] = pts[(size_t)rows[r] * c + j];
}

extern "C" int irx_scene_sample(const void* src, int n_src, int c, const int32_t* choices, int n, int flip_x,
                                int flip_y, const double* rot, int n_rot, const double* shift, void* dst,
                                int elem_bytes, void* stream) {
  IRX_REQUIRE(elem_bytes == 4 || elem_bytes == 8, "irx_scene_sample: elem_bytes must be 4 or 8 (got %d)", elem_bytes);
  IRX_REQUIRE(c >= 3 && n >= 0 && n_src >= 0 && n_rot >= 0 && n_rot <= 3, "irx_scene_sample: bad sizes");
  IRX_REQUIRE(n_rot == 0 || rot, "irx_scene_sample: rot is NULL");
  if (n == 0) return 0;
  IRX_REQUIRE(src && choices && dst && n_src > 0, "irx_scene_sample: NULL argument");
  IrxAug aug;
  aug.flip_x = flip_x;
  aug.flip_y = flip_y;
  aug.n_rot = n_rot;
  aug.has_shift = shift != nullptr;
  for (int m = 0; m < 3; ++m)
    for (int k = 0; k < 9; ++k) aug.rot[m][k] = (m < n_rot) ? rot[m * 9 + k] : 0.0;
  for (int k = 0; k < 3; ++k) aug.shift[k] = shift ? shift[k] : 0.0;
  const int grid = (n + 255) / 256;
  if (elem_bytes == 4)
    k_scene_sample<float><<<grid, 256, 0, S(stream)>>>((const float*)src, choices, n, c, aug, (float*)dst);
  else
    k_scene_sample<double><<<grid, 256, 0, S(stream)>>>((const double*)src, choices, n, c, aug, (double*)dst);
  IRX_CHECK_LAUNCH("k_scene_sample");
  return 0;
}

extern "C" int irx_instance_split(const void* pts, int n, int c, const int32_t* order, const int32_t* seg, int n_inst,
                                  const int32_t* rows, int n_sample, void* inst_points, double* obbs, void* extent,
                                  int elem_bytes, void* stream) {
  IRX_REQUIRE(elem_bytes == 4 || elem_bytes == 8, "irx_instance_split: elem_bytes must be 4 or 8 (got %d)", elem_bytes);
  IRX_REQUIRE(c >= 3 && n >= 0 && n_inst >= 0 && n_sample >= 0, "irx_instance_split: bad sizes");
  if (n == 0) return 0;
  IRX_REQUIRE(pts, "irx_instance_split: pts is NULL");
  IRX_REQUIRE(n_inst == 0 || (order && seg && obbs), "irx_instance_split: NULL instance argument");
  const int blocks = n_inst + (extent ? 1 : 0);
  if (blocks > 0) {
    if (elem_bytes == 4)
      k_instance_box<float><<<blocks, 256, 0, S(stream)>>>((const float*)pts, n, c, order, seg, n_inst, obbs,
                                                           (float*)extent);
    else
      k_instance_box<double><<<blocks, 256, 0, S(stream)>>>((const double*)pts, n, c, order, seg, n_inst, obbs,
                                                            (double*)extent);
    IRX_CHECK_LAUNCH("k_instance_box");
  }
  const size_t total = (size_t)n_inst * n_sample * c;
  if (total > 0) {
    IRX_REQUIRE(rows && inst_points, "irx_instance_split: rows / inst_points is NULL");
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (elem_bytes == 4)
      k_instance_gather<float><<<grid, 256, 0, S(stream)>>>((const float*)pts, c, rows, total, (float*)inst_points);
    else
      k_instance_gather<double><<<grid, 256, 0, S(stream)>>>((const double*)pts, c, rows, total, (double*)inst_points);
    IRX_CHECK_LAUNCH("k_instance_gather");
  }
  return 0;
}

// gslot[r] = slot_src[choices[r]] + slot_base, sem[r] = sem_src[choices[r]]  (labels of the sampled points, as int64
// for torch's bincount / scatter_reduce / sort)
__global__ __launch_bounds__(256) void k_sample_labels(const int32_t* __restrict__ slot_src,
                                                       const int32_t* __restrict__ sem_src,
                                                       const int64_t* __restrict__ choices, int n, int64_t slot_base,
                                                       int64_t* __restrict__ gslot, int64_t* __restrict__ sem) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int64_t v = choices[r];
  gslot[r] = (int64_t)slot_src[v] + slot_base;
  sem[r] = (int64_t)sem_src[v];
}

template <typename T>
__global__ __launch_bounds__(256) void k_scene_sample64(const T* __restrict__ src, const int64_t* __restrict__ choices,
                                                        int n, int c, IrxAug aug, T* __restrict__ dst) {
#pragma clang fp contract(off)
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const T* s = src + (size_t)choices[r] * c;
  T* d = dst + (size_t)r * c;
  T x = s[0], y = s[1], z = s[2];
  if (aug.flip_x) x = -x;
  if (aug.flip_y) y = -y;
  for (int m = 0; m < aug.n_rot; ++m) {
    const double* R = aug.rot[m];
    const double px = (double)x, py = (double)y, pz = (double)z;
    x = (T)((px * R[0] + py * R[1]) + pz * R[2]);
    y = (T)((px * R[3] + py * R[4]) + pz * R[5]);
    z = (T)((px * R[6] + py * R[7]) + pz * R[8]);
  }
  if (aug.has_shift) {
    x = (T)((double)x + aug.shift[0]);
    y = (T)((double)y + aug.shift[1]);
    z = (T)((double)z + aug.shift[2]);
  }
  d[0] = x;
  d[1] = y;
  d[2] = z;
  for (int j = 3; j < c; ++j) d[j] = s[j];
}

extern "C" int irx_scene_sample_batch(int n_samples, const void* const* src, const int* n_src, int c,
                                      const int64_t* const* choices, int n, const int* flip_xy, const double* rot,
                                      const int* n_rot, const double* shift, const int* has_shift, void* dst,
                                      const int32_t* const* slot_src, const int32_t* const* sem_src,
                                      const int64_t* slot_base, int64_t* gslot, int64_t* sem, int elem_bytes,
                                      void* stream) {
  IRX_REQUIRE(elem_bytes == 4 || elem_bytes == 8, "irx_scene_sample_batch: elem_bytes must be 4 or 8 (got %d)", elem_bytes);
  IRX_REQUIRE(n_samples >= 0 && c >= 3 && n >= 0, "irx_scene_sample_batch: bad sizes");
  if (n_samples == 0 || n == 0) return 0;
  IRX_REQUIRE(src && n_src && choices && flip_xy && n_rot && has_shift && dst, "irx_scene_sample_batch: NULL argument");
  const int grid = (n + 255) / 256;
  for (int b = 0; b < n_samples; ++b) {
    IRX_REQUIRE(src[b] && choices[b] && n_src[b] > 0 && n_rot[b] >= 0 && n_rot[b] <= 3,
                "irx_scene_sample_batch: sample %d: NULL pointer or bad sizes", b);
    IRX_REQUIRE((n_rot[b] == 0 || rot) && (!has_shift[b] || shift), "irx_scene_sample_batch: rot / shift is NULL");
    IrxAug aug;
    aug.flip_x = flip_xy[2 * b];
    aug.flip_y = flip_xy[2 * b + 1];
    aug.n_rot = n_rot[b];
    aug.has_shift = has_shift[b];
    for (int m = 0; m < 3; ++m)
      for (int k = 0; k < 9; ++k) aug.rot[m][k] = (m < n_rot[b]) ? rot[(size_t)b * 27 + m * 9 + k] : 0.0;
    for (int k = 0; k < 3; ++k) aug.shift[k] = has_shift[b] ? shift[(size_t)b * 3 + k] : 0.0;
    char* d = (char*)dst + (size_t)b * n * c * elem_bytes;
    if (elem_bytes == 4)
      k_scene_sample64<float><<<grid, 256, 0, S(stream)>>>((const float*)src[b], choices[b], n, c, aug, (float*)d);
    else
      k_scene_sample64<double><<<grid, 256, 0, S(stream)>>>((const double*)src[b], choices[b], n, c, aug, (double*)d);
    if (gslot) {
      IRX_REQUIRE(slot_src && sem_src && slot_base && sem && slot_src[b] && sem_src[b],
                  "irx_scene_sample_batch: label arrays of sample %d are NULL", b);
      k_sample_labels<<<grid, 256, 0, S(stream)>>>(slot_src[b], sem_src[b], choices[b], n, slot_base[b],
                                                   gslot + (size_t)b * n, sem + (size_t)b * n);
    }
  }
  IRX_CHECK_LAUNCH("irx_scene_sample_batch");
  return 0;
}

// ---- counter-based randomness for the fully device-side mode: no sort, no generator state ---------------------------
__device__ __forceinline__ uint32_t irx_mix32(uint32_t x) {   // murmur3 finaliser
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}
// Pseudo-random PERMUTATION of [0, N): 4-round Feistel network on the smallest even-bit domain >= N, cycle-walked back
// into range (a bijection of the domain restricted to [0, N) stays a bijection; < 4 walks expected). perm(0..n-1) is
// therefore a size-n subset WITHOUT replacement, computed per element with no sort and no state.
__device__ __forceinline__ uint32_t irx_prp(uint32_t x, uint32_t N, uint64_t key) {
  if (N <= 1) return 0;
  const int bits = 32 - __clz(N - 1);
  const int h = (bits + 1) >> 1;
  const uint32_t mask = (h >= 32) ? 0xFFFFFFFFu : ((1u << h) - 1u);
  do {
    uint32_t L = x >> h, R = x & mask;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t F = irx_mix32(R ^ (uint32_t)(key >> (16 * r)) ^ (0x9E3779B9u * (uint32_t)(r + 1))) & mask;
      const uint32_t t = L ^ F;
      L = R;
      R = t;
    }
    x = (L << h) | R;
  } while (x >= N);
  return x;
}

struct IrxSubsetJobs {
  int n_src[64];
  unsigned long long seed[64];
};
// out[b][r]: V >= n -> prp_b(r) (subset without replacement); V < n -> a hashed uniform draw (with replacement)
__global__ __launch_bounds__(256) void k_random_subset(IrxSubsetJobs J, int n, int64_t* __restrict__ out) {
  const int b = blockIdx.y;
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint32_t V = (uint32_t)J.n_src[b];
  const uint64_t key = J.seed[b];
  uint32_t v;
  if (V >= (uint32_t)n) {
    v = irx_prp((uint32_t)r, V, key);
  } else {
    const uint64_t hsh = ((uint64_t)irx_mix32((uint32_t)r ^ (uint32_t)key) << 32) | irx_mix32((uint32_t)r * 0x9E3779B9u ^ (uint32_t)(key >> 32));
    v = (uint32_t)(hsh % V);
  }
  out[(size_t)b * n + r] = (int64_t)v;
}

// rows[i][s] = order[seg[i] + j]: j = prp_i(s) when the instance has >= n_sample points (distinct rows), a hashed
// uniform draw otherwise (with replacement), 0 for an empty slot (its rows are dropped by the caller)
__global__ __launch_bounds__(256) void k_resample_rows(const int32_t* __restrict__ order, const int32_t* __restrict__ seg,
                                                       int n_slots, int n_sample, unsigned long long seed,
                                                       int32_t* __restrict__ rows) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)n_slots * n_sample) return;
  const int i = (int)(e / n_sample), s = (int)(e % n_sample);
  const int beg = seg[i], cnt = seg[i + 1] - beg;
  if (cnt <= 0) { rows[e] = 0; return; }
  const uint64_t key = seed ^ (0xD6E8FEB86659FD93ull * (uint64_t)(i + 1));
  uint32_t j;
  if (cnt >= n_sample) {
    j = irx_prp((uint32_t)s, (uint32_t)cnt, key);
  } else {
    const uint64_t hsh = ((uint64_t)irx_mix32((uint32_t)s ^ (uint32_t)key) << 32) | irx_mix32((uint32_t)s * 0x9E3779B9u ^ (uint32_t)(key >> 32));
    j = (uint32_t)(hsh % (uint32_t)cnt);
  }
  rows[e] = order[beg + (int)j];
}

extern "C" int irx_random_subset(int n_samples, const int* n_src, int n, const uint64_t* seeds, int64_t* out, void* stream) {
  IRX_REQUIRE(n_samples >= 0 && n_samples <= 64 && n >= 0, "irx_random_subset: n_samples %d outside [0, 64] or n < 0", n_samples);
  if (n_samples == 0 || n == 0) return 0;
  IRX_REQUIRE(n_src && seeds && out, "irx_random_subset: NULL argument");
  IrxSubsetJobs J;
  for (int b = 0; b < n_samples; ++b) {
    IRX_REQUIRE(n_src[b] > 0, "irx_random_subset: sample %d has no vertices", b);
    J.n_src[b] = n_src[b];
    J.seed[b] = seeds[b];
  }
  dim3 grid((n + 255) / 256, n_samples);
  k_random_subset<<<grid, 256, 0, S(stream)>>>(J, n, out);
  IRX_CHECK_LAUNCH("irx_random_subset");
  return 0;
}

extern "C" int irx_resample_rows(const int32_t* order, const int32_t* seg, int n_slots, int n_sample, uint64_t seed,
                                 int32_t* rows, void* stream) {
  IRX_REQUIRE(n_slots >= 0 && n_sample >= 0, "irx_resample_rows: bad sizes");
  const size_t total = (size_t)n_slots * n_sample;
  if (total == 0) return 0;
  IRX_REQUIRE(order && seg && rows, "irx_resample_rows: NULL argument");
  k_resample_rows<<<(unsigned)((total + 255) / 256), 256, 0, S(stream)>>>(order, seg, n_slots, n_sample, seed, rows);
  IRX_CHECK_LAUNCH("irx_resample_rows");
  return 0;
}
