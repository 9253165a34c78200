// irx_norm.hip — BatchNorm over voxel rows with fused residual add and ReLU, forward + backward.
// HBM-bound streaming kernels: 16-byte loads, one pass for statistics, one pass for apply.
// Replaces spnn.BatchNorm + spnn.ReLU + the SparseTensor `+` of ResidualBlock
// (reference models/basic_blocks.py:20-21,37-38,44,52,55).
#include <stdlib.h>
#include "irx_common.h"

// voxel rows per statistics workgroup: ~256 KB of one fp32 tensor per workgroup (2048 rows at 32 channels, 512 at 128),
// halved while that leaves fewer than 128 workgroups, never below 256 rows. Measured (tools/bn_microbench.py, stats /
// backward): 489 k x 32: 24.6 / 76.8 us with 512 rows -> 16.6 / 70.6 with 2048; 259 k x 64: 24.0 / 80.7 (256) -> 16.7 / 73.5
// (1024); smaller blocks are slower at every size (64 rows: 91 us at 489 k x 32 — the float64 fold over the partials grows).
static inline int bn_rows(int n, int c) {
  static const int forced = getenv("IRX_BN_ROWS") ? atoi(getenv("IRX_BN_ROWS")) : 0;   // dev A/B knob
  if (forced > 0) return forced;
  int rows = 65536 / (c < 32 ? 32 : c);
  if (rows > 2048) rows = 2048;
  // at least 64 rows per statistics workgroup (256 until round 5's second session: the passes that fold the offset-split slabs of a
  // small level read S x 4 bytes per element from 36 workgroups — k_bn_partial1_slabs 32.6 -> 25.1 us, k_bn_partial_slabs 17.5 ->
  // 11.8 us average at B = 16, 2.05 -> 1.90 ms of BatchNorm kernel time per step; tools/micro/bn_rows_stats.sh).  IRX_BN_ROWS_MIN: dev knob
  static const int floor_ = getenv("IRX_BN_ROWS_MIN") ? atoi(getenv("IRX_BN_ROWS_MIN")) : 64;
  while (rows > floor_ && n / rows < 128) rows >>= 1;
  if (rows < floor_) rows = floor_;
  return rows;
}

static inline hipStream_t S(void* s) { return (hipStream_t)s; }
static inline hipStream_t S_(void* s) { return (hipStream_t)s; }   // (where a parameter named S shadows the helper)
#include <atomic>

// Element types of the tensors of one BatchNorm call (0 = float32, 1 = bf16; bf16 needs the 4-wide path): the encoder
// executor's bf16 storage mode keeps conv outputs, layer outputs and the gradients in flight as bf16 while the
// statistics, the arithmetic and the parameter gradients stay fp32. The C-ABI entry points pass all zeros.
struct BnTy {
  int x, y, dy, dx, res;
};

// TY == false: every tensor is fp32 and the loads are the plain float4 loads the compiler batches (a run-time type
// flag in front of each load kept it from issuing the rows' loads together: k_bn_partial<1> 14.7 -> 27 us);
// TY == true (V == 4 only): mixed element types, run-time flags — the executor's LAST layer in bf16 storage mode.
template <bool TY>
__device__ __forceinline__ float4 bn_ld4(const float* p, size_t off, int bf) {
  if constexpr (TY) return irx_ld4(p, off, bf);
  else return *reinterpret_cast<const float4*>(p + off);
}
template <bool TY>
__device__ __forceinline__ void bn_st4(float* p, size_t off, int bf, float4 v) {
  if constexpr (TY) irx_st4(p, off, bf, v);
  else *reinterpret_cast<float4*>(p + off) = v;
}

// V == 8: every tensor of the call is bf16 and a thread owns 8 channels = one 16-byte load / store per tensor and row
// (with V == 4 a bf16 row quad is only 8 bytes per lane and the streaming kernels run at half their bytes in flight)
__device__ __forceinline__ void bn_ld8(const float* p, size_t off, float (&v)[8]) {
  const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p) + off);
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
  v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
  v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ void bn_st8(float* p, size_t off, const float (&v)[8]) {
  *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(p) + off) =
      make_uint4(irx_pk_bf16(v[0], v[1]), irx_pk_bf16(v[2], v[3]), irx_pk_bf16(v[4], v[5]), irx_pk_bf16(v[6], v[7]));
}

// finalize-in-the-last-workgroup arguments of k_bn_partial (counter == NULL: the separate k_bn_finalize launch does it)
struct BnFin {
  unsigned* counter = nullptr;
  float eps = 0.f, momentum = 0.f;
  float *out0 = nullptr, *out1 = nullptr, *running_mean = nullptr, *running_var = nullptr;
};
// ticket counters of the launches in flight: zero at module load, left at zero by every kernel that used one; handed out round
// robin (two launches share one only if more than BN_NCOUNTERS statistics kernels are in flight at once)
#define BN_NCOUNTERS 4096
__device__ unsigned g_bn_counters[BN_NCOUNTERS];

__host__ __device__ static inline int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// Per-workgroup partial sums over BN_ROWS rows.
//   MODE 0: (sum x, sum x^2)
//   MODE 1: (sum g, sum g*xhat), g = dy * (RELU ? y > 0 : 1), xhat = (x - mean) * invstd
// V = vector width (4 when c % 4 == 0 else 1). part layout [blk][2][c].
// threads per statistics workgroup: 512 (round 3, rocprofv3 same box: k_bn_partial<1> 15.7 / 14.5 / 16.2 us and k_bn_partial<0>
// 7.2 / 7.0 / 8.3 us at 256 / 512 / 1024 — these launches have fewer workgroups than the chip has CUs, so more loads in
// flight per workgroup help until the LDS fold grows); -DBN_PT=... for A/B builds
#ifndef BN_PT
#define BN_PT 512
#endif
// rows whose loads a thread has in flight per trip (-DBN_U=... for A/B builds)
#ifndef BN_U
#define BN_U 4
#endif
template <int MODE, int V, bool TY, bool RM = false, bool RL = true>
__global__ __launch_bounds__(BN_PT) void k_bn_partial(const float* __restrict__ x,
                                                    const float* __restrict__ y,
                                                    const float* __restrict__ dy, int n, int c,
                                                    const float* __restrict__ mean,
                                                    const float* __restrict__ invstd, int relu_arg,
                                                    int qpad, int rows_per_block, float* __restrict__ part, BnTy ty,
                                                    const float* __restrict__ mk_gamma, const float* __restrict__ mk_beta,
                                                    BnFin fin = BnFin()) {
  // RM (compile time: a run-time flag in front of the loads keeps the compiler from batching a row group's loads, measured
  // again in round 3: 15.8 -> 21.8 us) with mk_gamma / mk_beta (MODE 1, relu, a layer WITHOUT a shortcut): the ReLU mask is recomputed from x —
  // y > 0  <=>  fma(x, invstd * gamma, fma(-mean, invstd * gamma, beta)) > 0, the very expression k_bn_apply evaluated — so y
  // is never read (a third of this pass's bytes)
  __shared__ __align__(16) float s0[BN_PT * V];
  __shared__ __align__(16) float s1[BN_PT * V];
  const int cq = c / V;
  const int qd = threadIdx.x % qpad;
  const int rg = threadIdx.x / qpad;
  const int nrg = BN_PT / qpad;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  float a0[V], a1[V];
#pragma unroll
  for (int j = 0; j < V; ++j) a0[j] = a1[j] = 0.f;
  if (qd < cq) {
    float mu[V], is[V], msc[V], msh[V];
    constexpr bool remask = RM;
    const bool relu = RL;                      // compile time (shadows the argument): no run-time flag in front of the y loads
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        mu[j] = mean[qd * V + j];
        is[j] = invstd[qd * V + j];
        msc[j] = msh[j] = 0.f;
        if (remask) {
          msc[j] = is[j] * mk_gamma[qd * V + j];
          msh[j] = fmaf(-mu[j], msc[j], mk_beta[qd * V + j]);
        }
      }
    }
    // U rows per trip: all their loads are issued before the first add (memory-level parallelism; a load per add left
    // the kernel latency-bound), the adds keep the row order (bit-identical to the one-row loop)
    constexpr int U = BN_U;
    auto load_row = [&](int r, float (&xv)[V], float (&yv)[V], float (&dv)[V]) __attribute__((always_inline)) {
      const size_t off = (size_t)r * c + (size_t)qd * V;
      if constexpr (V == 8) {
        bn_ld8(x, off, xv);
        if (MODE == 1) {
          bn_ld8(dy, off, dv);
          if (relu && !remask) bn_ld8(y, off, yv);
        }
      } else if constexpr (V == 4) {
        *reinterpret_cast<float4*>(xv) = bn_ld4<TY>(x, off, ty.x);
        if (MODE == 1) {
          *reinterpret_cast<float4*>(dv) = bn_ld4<TY>(dy, off, ty.dy);
          if (relu && !remask) *reinterpret_cast<float4*>(yv) = bn_ld4<TY>(y, off, ty.y);
        }
      } else {
        xv[0] = x[off];
        if (MODE == 1) {
          dv[0] = dy[off];
          if (relu && !remask) yv[0] = y[off];
        }
      }
    };
    auto add_row = [&](const float (&xv)[V], const float (&yv)[V], const float (&dv)[V]) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        if (MODE == 0) {
          a0[j] += xv[j];
          a1[j] += xv[j] * xv[j];
        } else {
          float gval = dv[j];
          // (the recomputed mask belongs HERE, behind all the loads of the row group: inside load_row it put a wait for x
          // between the rows' loads — 15.7 -> 18.3 us)
          const float yy = remask ? fmaf(xv[j], msc[j], msh[j]) : yv[j];
          if (relu && !(yy > 0.f)) gval = 0.f;
          a0[j] += gval;
          a1[j] += gval * ((xv[j] - mu[j]) * is[j]);
        }
      }
    };
    int r = r0 + rg;
    for (; r + (U - 1) * nrg < r1; r += U * nrg) {
      float xv[U][V], yv[U][V], dv[U][V];
#pragma unroll
      for (int u = 0; u < U; ++u) load_row(r + u * nrg, xv[u], yv[u], dv[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) add_row(xv[u], yv[u], dv[u]);
    }
    for (; r < r1; r += nrg) {
      float xv[V], yv[V], dv[V];
      load_row(r, xv, yv, dv);
      add_row(xv, yv, dv);
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) {
    s0[threadIdx.x * V + j] = a0[j];
    s1[threadIdx.x * V + j] = a1[j];
  }
  __syncthreads();
  // threads (rg < V, qd) fold the row groups, one channel each (same sequential order per channel as one thread walking all V:
  // bit-identical; with V = 8 and 32 channels that one thread read 2 x 8 x 128 LDS words in a row — k_bn_partial<0, 8> 21.8 us on
  // 61 k x 32 against 8.2 us for the fp32 kernel with twice the bytes)
  const int jw = nrg < V ? nrg : V;
  if (rg < jw && qd < cq) {
    for (int j = rg; j < V; j += jw) {
      float t0 = 0.f, t1 = 0.f;
      for (int g2 = 0; g2 < nrg; ++g2) {
        t0 += s0[(g2 * qpad + qd) * V + j];
        t1 += s1[(g2 * qpad + qd) * V + j];
      }
      if (V >= 4 && MODE <= 1 && fin.counter != nullptr) {
        // published for the last workgroup (below): agent-scope relaxed atomic stores go THROUGH the L2 to memory (sc1), so that
        // no cache write-back is needed to make them visible to a workgroup on another XCD
        __hip_atomic_store(&part[((size_t)blockIdx.x * 2 + 0) * c + qd * V + j], t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&part[((size_t)blockIdx.x * 2 + 1) * c + qd * V + j], t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        part[((size_t)blockIdx.x * 2 + 0) * c + qd * V + j] = t0;
        part[((size_t)blockIdx.x * 2 + 1) * c + qd * V + j] = t1;
      }
    }
  }
  if constexpr (V >= 4 && MODE <= 1) {
    // Finalize folded into the LAST workgroup to finish (round 5; fin.counter != NULL): every workgroup publishes its partials
    // (agent-scope release fence), takes a ticket, and the holder of the last ticket folds all partials in float64 — what
    // k_bn_finalize did in a launch of its own (26 layers x 2 directions x ~5 us of dependent launch latency per step). Nobody
    // waits for anybody: no spinning, no co-residency requirement. The counter is left at 0 for its next user.
    if (fin.counter == nullptr) return;
    __shared__ int s_last;
    // Round 5, second attempt. The first (and round 2's) used __threadfence() on both sides: an agent-scope release / acquire is
    // a write-back / invalidate of the XCD's whole L2 on gfx950 and cost more than the launch it replaced (39.9 -> 57.8 us at
    // 489 k x 32). Here nothing is flushed: the partials are written and read with agent-scope RELAXED atomics (memory-side, sc1),
    // and the only ordering needed — "my partial stores are acknowledged before my ticket" — is a wait for this wave's stores
    // (workgroup-scope release = s_waitcnt, no cache operation) followed by the workgroup barrier.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0)
      s_last = (__hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    const int nblk = gridDim.x;
    const int cq4 = c / 4, nsl = BN_PT / cq4;            // float4 columns x slices of the partial blocks
    const int col = threadIdx.x % cq4, sl = threadIdx.x / cq4;
    double t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = 0.0;
    if (sl < nsl) {
      for (int b = sl; b < nblk; b += nsl) {
        float pv[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pv[j] = __hip_atomic_load(part + ((size_t)b * 2 + 0) * c + col * 4 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          pv[4 + j] = __hip_atomic_load(part + ((size_t)b * 2 + 1) * c + col * 4 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] += (double)pv[j];
      }
    }
    double* sd = reinterpret_cast<double*>(s0);           // BN_PT doubles fit: V >= 4
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __syncthreads();
      sd[threadIdx.x] = t[j];
      __syncthreads();
      if (sl == 0) {
        double acc = t[j];
        for (int q = 1; q < nsl; ++q) acc += sd[q * cq4 + col];
        t[j] = acc;
      }
    }
    if (sl == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ch = col * 4 + j;
        const double t0 = t[j], t1 = t[4 + j];
        if (MODE == 0) {
          const double mu = t0 / (double)n;
          double var = t1 / (double)n - mu * mu;
          if (var < 0.0) var = 0.0;
          fin.out0[ch] = (float)mu;
          fin.out1[ch] = (float)(1.0 / sqrt(var + (double)fin.eps));
          if (fin.running_mean) {
            const double unbiased = (n > 1) ? var * (double)n / (double)(n - 1) : var;
            fin.running_mean[ch] = (float)((1.0 - fin.momentum) * (double)fin.running_mean[ch] + fin.momentum * mu);
            fin.running_var[ch] = (float)((1.0 - fin.momentum) * (double)fin.running_var[ch] + fin.momentum * unbiased);
          }
        } else {
          fin.out0[ch] = (float)t0;
          fin.out1[ch] = (float)t1;
        }
      }
    }
    if (threadIdx.x == 0) __hip_atomic_store(fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Fold partials in float64. One workgroup per 32 channels: 32 channels x 32 slices (1024 threads), so even the
// largest level (~2000 partial blocks) is ~60 dependent loads deep.
//   MODE 0: mean / invstd (+ running stats);  MODE 1: out0 = sum g (dbeta), out1 = sum g*xhat (dgamma)
#define BN_FIN_SLICES 32
template <int MODE>
__global__ __launch_bounds__(1024) void k_bn_finalize(const float* __restrict__ part, int nblk, int n,
                                                     int c, float eps, float momentum,
                                                     float* __restrict__ out0, float* __restrict__ out1,
                                                     float* __restrict__ running_mean,
                                                     float* __restrict__ running_var) {
  __shared__ double d0[32 * BN_FIN_SLICES];
  __shared__ double d1[32 * BN_FIN_SLICES];
  const int ch = blockIdx.x * 32 + (threadIdx.x & 31);
  const int slice = threadIdx.x >> 5;
  double t0 = 0.0, t1 = 0.0;
  if (ch < c) {
    for (int b = slice; b < nblk; b += BN_FIN_SLICES) {
      t0 += (double)part[((size_t)b * 2 + 0) * c + ch];
      t1 += (double)part[((size_t)b * 2 + 1) * c + ch];
    }
  }
  d0[threadIdx.x] = t0;
  d1[threadIdx.x] = t1;
  __syncthreads();
  if (slice == 0 && ch < c) {
    for (int s2 = 1; s2 < BN_FIN_SLICES; ++s2) {
      t0 += d0[s2 * 32 + (threadIdx.x & 31)];
      t1 += d1[s2 * 32 + (threadIdx.x & 31)];
    }
    if (MODE == 0) {
      const double mu = t0 / (double)n;
      double var = t1 / (double)n - mu * mu;
      if (var < 0.0) var = 0.0;
      out0[ch] = (float)mu;
      out1[ch] = (float)(1.0 / sqrt(var + (double)eps));
      if (running_mean) {
        const double unbiased = (n > 1) ? var * (double)n / (double)(n - 1) : var;
        running_mean[ch] = (float)((1.0 - momentum) * (double)running_mean[ch] + momentum * mu);
        running_var[ch] = (float)((1.0 - momentum) * (double)running_var[ch] + momentum * unbiased);
      }
    } else if (MODE == 2) {                       // raw float64 sums (sync BatchNorm: summed over the ranks before use)
      reinterpret_cast<double*>(out0)[ch] = t0;
      reinterpret_cast<double*>(out1)[ch] = t1;
    } else {
      out0[ch] = (float)t0;
      out1[ch] = (float)t1;
    }
  }
}

// mean / invstd (+ running statistics) from (sum x, sum x^2, count) — the tail of k_bn_finalize<0> for sums that were
// folded over the ranks first
__global__ void k_bn_from_sums(const double* __restrict__ sums, double count, int c, float eps, float momentum,
                               float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ running_mean,
                               float* __restrict__ running_var) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  if (count < 1.0) count = sums[2 * c];          // the folded row count travels behind the sums (no host round trip)
  const double mu = sums[ch] / count;
  double var = sums[c + ch] / count - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[ch] = (float)mu;
  invstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = (count > 1.0) ? var * count / (count - 1.0) : var;
    running_mean[ch] = (float)((1.0 - momentum) * (double)running_mean[ch] + momentum * mu);
    running_var[ch] = (float)((1.0 - momentum) * (double)running_var[ch] + momentum * unbiased);
  }
}

// Apply kernels: every thread owns ONE channel group of V channels (its scale / shift live in registers) and walks
// rows with a fixed stride, so the inner loop is load - fma - store with no index arithmetic beyond an add.
// qpad = power of two >= c / V threads per row; rows_per_pass = 256 / qpad per workgroup.
// RES / RL (shortcut operand present, ReLU) are compile-time: a run-time flag in front of a load keeps the compiler from
// batching the loads of a row (measured on the backward kernels: +25-40 %)
template <int V, bool TY, bool RES, bool RL>
__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ x, int n, int c, int qpad,
                                                  const float* __restrict__ mean,
                                                  const float* __restrict__ invstd,
                                                  const float* __restrict__ gamma,
                                                  const float* __restrict__ beta,
                                                  const float* __restrict__ res, int relu,
                                                  float* __restrict__ y, BnTy ty) {
  const int cq = c / V;
  const int qd = threadIdx.x % qpad;
  if (qd >= cq) return;
  const int rpp = 256 / qpad;
  float sc[V], sh[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    sc[j] = invstd[qd * V + j] * gamma[qd * V + j];
    sh[j] = fmaf(-mean[qd * V + j], sc[j], beta[qd * V + j]);    // (k_bn_partial<1> / k_bn_bwd_apply re-evaluate exactly this)
  }
  const int row_stride = gridDim.x * rpp;
  for (int r = blockIdx.x * rpp + threadIdx.x / qpad; r < n; r += row_stride) {
    const size_t off = (size_t)r * c + (size_t)qd * V;
    float xv[V], rv[V], ov[V];
    if constexpr (V == 8) {
      bn_ld8(x, off, xv);
      if constexpr (RES) bn_ld8(res, off, rv);
    } else if constexpr (V == 4) {
      *reinterpret_cast<float4*>(xv) = bn_ld4<TY>(x, off, ty.x);
      if constexpr (RES) *reinterpret_cast<float4*>(rv) = bn_ld4<TY>(res, off, ty.res);
    } else {
      xv[0] = x[off];
      if constexpr (RES) rv[0] = res[off];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float o = fmaf(xv[j], sc[j], sh[j]);
      if constexpr (RES) o += rv[j];
      if constexpr (RL) o = o > 0.f ? o : 0.f;
      ov[j] = o;
    }
    if constexpr (V == 8)
      bn_st8(y, off, ov);
    else if constexpr (V == 4)
      bn_st4<TY>(y, off, ty.y, *reinterpret_cast<float4*>(ov));
    else
      y[off] = ov[0];
  }
}

// dx = gamma*invstd*(g - sum_g/n - xhat*sum_gx/n);  dres = g
template <int V, bool TY, bool RM, bool RL, bool DR>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ x,
                                                      const float* __restrict__ y,
                                                      const float* __restrict__ dy, int n, int c, int qpad,
                                                      const float* __restrict__ mean,
                                                      const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ sum_g,
                                                      const float* __restrict__ sum_gx, int relu,
                                                      float* __restrict__ dx, float* __restrict__ dres, BnTy ty,
                                                      float inv_count, const double* __restrict__ count_dev,
                                                      const float* __restrict__ mk_beta) {
  // inv_count: 1 / (rows the sums were taken over) — 1 / n, or 1 / (rows of ALL ranks) for sync BatchNorm
  const int cq = c / V;
  const int qd = threadIdx.x % qpad;
  if (qd >= cq) return;
  const int rpp = 256 / qpad;
  const float inv_n = count_dev ? (float)(1.0 / *count_dev) : inv_count;
  float mu[V], is[V], gi[V], sg[V], sgx[V], msc[V], msh[V];
  constexpr bool remask = RM;                              // see k_bn_partial
#pragma unroll
  for (int j = 0; j < V; ++j) {
    mu[j] = mean[qd * V + j];
    is[j] = invstd[qd * V + j];
    gi[j] = gamma[qd * V + j] * is[j];
    sg[j] = sum_g[qd * V + j] * inv_n;
    sgx[j] = sum_gx[qd * V + j] * inv_n;
    msc[j] = msh[j] = 0.f;
    if constexpr (RM) {
      msc[j] = is[j] * gamma[qd * V + j];
      msh[j] = fmaf(-mu[j], msc[j], mk_beta[qd * V + j]);
    }
  }
  const int row_stride = gridDim.x * rpp;
  for (int r = blockIdx.x * rpp + threadIdx.x / qpad; r < n; r += row_stride) {
    const size_t off = (size_t)r * c + (size_t)qd * V;
    float xv[V], yv[V], dv[V], ox[V], og[V];
    if constexpr (V == 8) {
      bn_ld8(x, off, xv);
      bn_ld8(dy, off, dv);
      if constexpr (RL && !RM) bn_ld8(y, off, yv);
    } else if constexpr (V == 4) {
      *reinterpret_cast<float4*>(xv) = bn_ld4<TY>(x, off, ty.x);
      *reinterpret_cast<float4*>(dv) = bn_ld4<TY>(dy, off, ty.dy);
      if constexpr (RL && !RM) *reinterpret_cast<float4*>(yv) = bn_ld4<TY>(y, off, ty.y);
    } else {
      xv[0] = x[off];
      dv[0] = dy[off];
      if constexpr (RL && !RM) yv[0] = y[off];
    }
    if constexpr (RM) {
#pragma unroll
      for (int j = 0; j < V; ++j) yv[j] = fmaf(xv[j], msc[j], msh[j]);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float gval = dv[j];
      if constexpr (RL) { if (!(yv[j] > 0.f)) gval = 0.f; }
      const float xh = (xv[j] - mu[j]) * is[j];
      ox[j] = gi[j] * (gval - sg[j] - xh * sgx[j]);
      og[j] = gval;
    }
    if constexpr (V == 8) {
      bn_st8(dx, off, ox);
      if constexpr (DR) bn_st8(dres, off, og);
    } else if constexpr (V == 4) {
      bn_st4<TY>(dx, off, ty.dx, *reinterpret_cast<float4*>(ox));
      if constexpr (DR) bn_st4<TY>(dres, off, ty.res, *reinterpret_cast<float4*>(og));
    } else {
      dx[off] = ox[0];
      if constexpr (DR) dres[off] = og[0];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Single-launch BatchNorm for tensors that stay on-die ("slice" kernels, round 5).
// The three-launch form above (partials -> float64 fold -> apply) costs a small level three dependent launches of 4-8 us each
// for a tensor of a few hundred KB: at B = 16 that was 168 launches and 1.5 ms per step, as much as the convolutions. Here one
// workgroup owns V consecutive channels (one 16-byte column of every row) and walks ALL rows twice: statistics, then apply
// (forward) / the two gradient sums, then dx (backward). Nothing crosses workgroups: no partials, no finalize launch, no fence,
// no counter. The second pass re-reads the rows from L2 / Infinity Cache (the host only takes this path below
// bn_slice_max_bytes()). C / V workgroups of 1024 threads: 16 at 128 bf16 channels, 32 at 128 fp32 channels.
// Arithmetic: per-thread fp32 sums over its <= n / 1024 rows, fp32 butterfly inside the wave, float64 across the 16 waves and for
// mean / variance; scale / shift and the backward expressions are the ones k_bn_apply / k_bn_bwd_apply evaluate, so the ReLU mask
// recomputed from x (RM) stays exact.
#define BN_SL_PT 1024
template <int V, int BF>
__device__ __forceinline__ void sl_ld(const float* __restrict__ p, size_t off, float (&v)[V]) {
  if constexpr (BF) {
    if constexpr (V == 8) {
      bn_ld8(p, off, v);
    } else {
      const float4 t = irx_bf4_to_f4(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p) + off));
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
  } else {
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(p + off + 4 * q);
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  }
}
template <int V, int BF>
__device__ __forceinline__ void sl_st(float* __restrict__ p, size_t off, const float (&v)[V]) {
  if constexpr (BF) {
    if constexpr (V == 8) {
      bn_st8(p, off, v);
    } else {
      *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p) + off) = irx_f4_to_bf4(make_float4(v[0], v[1], v[2], v[3]));
    }
  } else {
#pragma unroll
    for (int q = 0; q < V / 4; ++q)
      *reinterpret_cast<float4*>(p + off + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  }
}

// fold the per-thread sums a0 / a1 over the workgroup -> tot[0..V) / tot[V..2V) as float64 in LDS (valid after the call's barrier)
template <int V>
__device__ __forceinline__ void sl_fold(float (&a0)[V], float (&a1)[V], float* s_w, double* tot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < V; ++j) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      a0[j] += __shfl_xor(a0[j], m, 64);
      a1[j] += __shfl_xor(a1[j], m, 64);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      s_w[wave * 2 * V + j] = a0[j];
      s_w[wave * 2 * V + V + j] = a1[j];
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * V) {
    double t = 0.0;
    for (int w = 0; w < BN_SL_PT / 64; ++w) t += (double)s_w[w * 2 * V + threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
}

template <int V, int XB, int YB, int RB, bool RES>
__global__ __launch_bounds__(BN_SL_PT) void k_bn_slice_fwd(const float* __restrict__ x, int n, int c, float eps, float momentum,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ res, float* __restrict__ y,
                                                           float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var) {
  __shared__ float s_w[(BN_SL_PT / 64) * 2 * V];
  __shared__ double tot[2 * V];
  __shared__ float s_sc[V], s_sh[V];
  const size_t ch0 = (size_t)blockIdx.x * V;
  float a0[V], a1[V];
#pragma unroll
  for (int j = 0; j < V; ++j) a0[j] = a1[j] = 0.f;
  constexpr int U = 4;
  int r = threadIdx.x;
  for (; r + (U - 1) * BN_SL_PT < n; r += U * BN_SL_PT) {
    float xv[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) sl_ld<V, XB>(x, (size_t)(r + u * BN_SL_PT) * c + ch0, xv[u]);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < V; ++j) {
        a0[j] += xv[u][j];
        a1[j] += xv[u][j] * xv[u][j];
      }
  }
  for (; r < n; r += BN_SL_PT) {
    float xv[V];
    sl_ld<V, XB>(x, (size_t)r * c + ch0, xv);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      a0[j] += xv[j];
      a1[j] += xv[j] * xv[j];
    }
  }
  sl_fold<V>(a0, a1, s_w, tot);
  if (threadIdx.x < V) {
    const int ch = (int)ch0 + threadIdx.x;
    const double mu = tot[threadIdx.x] / (double)n;
    double var = tot[V + threadIdx.x] / (double)n - mu * mu;
    if (var < 0.0) var = 0.0;
    const float mf = (float)mu, isf = (float)(1.0 / sqrt(var + (double)eps));
    mean_out[ch] = mf;
    invstd_out[ch] = isf;
    if (running_mean) {
      const double unbiased = (n > 1) ? var * (double)n / (double)(n - 1) : var;
      running_mean[ch] = (float)((1.0 - momentum) * (double)running_mean[ch] + momentum * mu);
      running_var[ch] = (float)((1.0 - momentum) * (double)running_var[ch] + momentum * unbiased);
    }
    const float sc = isf * gamma[ch];                    // (k_bn_apply's own expressions, on the stored float statistics)
    s_sc[threadIdx.x] = sc;
    s_sh[threadIdx.x] = fmaf(-mf, sc, beta[ch]);
  }
  __syncthreads();
  float sc[V], sh[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    sc[j] = s_sc[j];
    sh[j] = s_sh[j];
  }
  r = threadIdx.x;
  for (; r + (U - 1) * BN_SL_PT < n; r += U * BN_SL_PT) {
    float xv[U][V], rv[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t off = (size_t)(r + u * BN_SL_PT) * c + ch0;
      sl_ld<V, XB>(x, off, xv[u]);
      if constexpr (RES) sl_ld<V, RB>(res, off, rv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float ov[V];
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float o = fmaf(xv[u][j], sc[j], sh[j]);
        if constexpr (RES) o += rv[u][j];
        ov[j] = o > 0.f ? o : 0.f;
      }
      sl_st<V, YB>(y, (size_t)(r + u * BN_SL_PT) * c + ch0, ov);
    }
  }
  for (; r < n; r += BN_SL_PT) {
    const size_t off = (size_t)r * c + ch0;
    float xv[V], rv[V], ov[V];
    sl_ld<V, XB>(x, off, xv);
    if constexpr (RES) sl_ld<V, RB>(res, off, rv);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float o = fmaf(xv[j], sc[j], sh[j]);
      if constexpr (RES) o += rv[j];
      ov[j] = o > 0.f ? o : 0.f;
    }
    sl_st<V, YB>(y, off, ov);
  }
}

// backward (ReLU layers): RM = mask recomputed from x (a layer without a shortcut; y is not read), DR = the shortcut's gradient
// dres = g is written too. XB: x, dx, dres;  YB: y and dy (the executor's last layer keeps both fp32 while x / dx / dres are bf16)
template <int V, int XB, int YB, bool RM, bool DR>
__global__ __launch_bounds__(BN_SL_PT) void k_bn_slice_bwd(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ dy, int n, int c,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ mk_beta,
                                                           float* __restrict__ dx, float* __restrict__ dres,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float s_w[(BN_SL_PT / 64) * 2 * V];
  __shared__ double tot[2 * V];
  const size_t ch0 = (size_t)blockIdx.x * V;
  float mu[V], is[V], gi[V], msc[V], msh[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    mu[j] = mean[ch0 + j];
    is[j] = invstd[ch0 + j];
    const float g = gamma ? gamma[ch0 + j] : 0.f;
    gi[j] = g * is[j];
    msc[j] = msh[j] = 0.f;
    if constexpr (RM) {
      msc[j] = is[j] * g;
      msh[j] = fmaf(-mu[j], msc[j], mk_beta[ch0 + j]);
    }
  }
  float a0[V], a1[V];
#pragma unroll
  for (int j = 0; j < V; ++j) a0[j] = a1[j] = 0.f;
  constexpr int U = 2;
  auto grad = [&](const float (&xv)[V], const float (&yv)[V], const float (&dv)[V], float (&g)[V]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float yy = RM ? fmaf(xv[j], msc[j], msh[j]) : yv[j];
      g[j] = (yy > 0.f) ? dv[j] : 0.f;
    }
  };
  int r = threadIdx.x;
  for (; r + (U - 1) * BN_SL_PT < n; r += U * BN_SL_PT) {
    float xv[U][V], yv[U][V], dv[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t off = (size_t)(r + u * BN_SL_PT) * c + ch0;
      sl_ld<V, XB>(x, off, xv[u]);
      sl_ld<V, YB>(dy, off, dv[u]);
      if constexpr (!RM) sl_ld<V, YB>(y, off, yv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float g[V];
      grad(xv[u], yv[u], dv[u], g);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        a0[j] += g[j];
        a1[j] += g[j] * ((xv[u][j] - mu[j]) * is[j]);
      }
    }
  }
  for (; r < n; r += BN_SL_PT) {
    const size_t off = (size_t)r * c + ch0;
    float xv[V], yv[V], dv[V], g[V];
    sl_ld<V, XB>(x, off, xv);
    sl_ld<V, YB>(dy, off, dv);
    if constexpr (!RM) sl_ld<V, YB>(y, off, yv);
    grad(xv, yv, dv, g);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      a0[j] += g[j];
      a1[j] += g[j] * ((xv[j] - mu[j]) * is[j]);
    }
  }
  sl_fold<V>(a0, a1, s_w, tot);
  const float inv_n = 1.f / (float)n;
  float sg[V], sgx[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const float t0 = (float)tot[j], t1 = (float)tot[V + j];     // what k_bn_finalize<1> stores and k_bn_bwd_apply reads back
    sg[j] = t0 * inv_n;
    sgx[j] = t1 * inv_n;
  }
  if (threadIdx.x < V) {
    dbeta[ch0 + threadIdx.x] = (float)tot[threadIdx.x];
    dgamma[ch0 + threadIdx.x] = (float)tot[V + threadIdx.x];
  }
  if (dx == nullptr) return;
  r = threadIdx.x;
  for (; r + (U - 1) * BN_SL_PT < n; r += U * BN_SL_PT) {
    float xv[U][V], yv[U][V], dv[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t off = (size_t)(r + u * BN_SL_PT) * c + ch0;
      sl_ld<V, XB>(x, off, xv[u]);
      sl_ld<V, YB>(dy, off, dv[u]);
      if constexpr (!RM) sl_ld<V, YB>(y, off, yv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t off = (size_t)(r + u * BN_SL_PT) * c + ch0;
      float g[V], ox[V];
      grad(xv[u], yv[u], dv[u], g);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float xh = (xv[u][j] - mu[j]) * is[j];
        ox[j] = gi[j] * (g[j] - sg[j] - xh * sgx[j]);
      }
      sl_st<V, XB>(dx, off, ox);
      if constexpr (DR) sl_st<V, XB>(dres, off, g);
    }
  }
  for (; r < n; r += BN_SL_PT) {
    const size_t off = (size_t)r * c + ch0;
    float xv[V], yv[V], dv[V], g[V], ox[V];
    sl_ld<V, XB>(x, off, xv);
    sl_ld<V, YB>(dy, off, dv);
    if constexpr (!RM) sl_ld<V, YB>(y, off, yv);
    grad(xv, yv, dv, g);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float xh = (xv[j] - mu[j]) * is[j];
      ox[j] = gi[j] * (g[j] - sg[j] - xh * sgx[j]);
    }
    sl_st<V, XB>(dx, off, ox);
    if constexpr (DR) sl_st<V, XB>(dres, off, g);
  }
}

// k_bn_partial<0> with the offset-split reduce of the convolution folded in (round 5): x does not exist yet — the conv left S fp32
// slabs [S][n][c] — so this pass sums them (slab order, the adds of k_wgrad_reduce), stores x (bf16: XB, else fp32) for the apply
// pass / the backward, and accumulates the statistics of the STORED value exactly like k_bn_partial<0> (same thread mapping, same
// row order, same fold): bit-identical to reduce-then-statistics, one dependent launch fewer per split layer (~9 per encoder pass).
template <int V, int XB>
__global__ __launch_bounds__(BN_PT) void k_bn_partial_slabs(const float* __restrict__ slabs, int S, size_t elems, int n, int c,
                                                          int qpad, int rows_per_block, float* __restrict__ x_out,
                                                          float* __restrict__ part) {
  __shared__ __align__(16) float s0[BN_PT * V];
  __shared__ __align__(16) float s1[BN_PT * V];
  const int cq = c / V;
  const int qd = threadIdx.x % qpad;
  const int rg = threadIdx.x / qpad;
  const int nrg = BN_PT / qpad;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  float a0[V], a1[V];
#pragma unroll
  for (int j = 0; j < V; ++j) a0[j] = a1[j] = 0.f;
  if (qd < cq) {
    for (int r = r0 + rg; r < r1; r += nrg) {
      const size_t off = (size_t)r * c + (size_t)qd * V;
      float v[V];
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = 0.f;
      int sidx = 0;
      for (; sidx + 4 <= S; sidx += 4) {                // 4 slabs' loads in flight, adds in slab order
        float4 t[4][V / 4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < V / 4; ++q) t[u][q] = *reinterpret_cast<const float4*>(slabs + (size_t)(sidx + u) * elems + off + 4 * q);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < V / 4; ++q) {
            v[4 * q] += t[u][q].x; v[4 * q + 1] += t[u][q].y; v[4 * q + 2] += t[u][q].z; v[4 * q + 3] += t[u][q].w;
          }
      }
      for (; sidx < S; ++sidx)
#pragma unroll
        for (int q = 0; q < V / 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(slabs + (size_t)sidx * elems + off + 4 * q);
          v[4 * q] += t.x; v[4 * q + 1] += t.y; v[4 * q + 2] += t.z; v[4 * q + 3] += t.w;
        }
      if constexpr (XB) {
#pragma unroll
        for (int j = 0; j < V; j += 2) {                  // what the store keeps is what the statistics see
          const unsigned pk = irx_pk_bf16(v[j], v[j + 1]);
          v[j] = __uint_as_float(pk << 16);
          v[j + 1] = __uint_as_float(pk & 0xffff0000u);
        }
      }
      sl_st<V, XB>(x_out, off, v);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        a0[j] += v[j];
        a1[j] += v[j] * v[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) {
    s0[threadIdx.x * V + j] = a0[j];
    s1[threadIdx.x * V + j] = a1[j];
  }
  __syncthreads();
  const int jw = nrg < V ? nrg : V;              // (k_bn_partial's fold: one channel per thread)
  if (rg < jw && qd < cq) {
    for (int j = rg; j < V; j += jw) {
      float t0 = 0.f, t1 = 0.f;
      for (int g2 = 0; g2 < nrg; ++g2) {
        t0 += s0[(g2 * qpad + qd) * V + j];
        t1 += s1[(g2 * qpad + qd) * V + j];
      }
      part[((size_t)blockIdx.x * 2 + 0) * c + qd * V + j] = t0;
      part[((size_t)blockIdx.x * 2 + 1) * c + qd * V + j] = t1;
    }
  }
}

// k_bn_partial<1> (ReLU layers, mask from y) with the offset-split reduce of the data-gradient convolution folded in: dy does not
// exist yet — the conv left S fp32 slabs — so this pass forms dy = (acc ? dy : 0) + sum of the slabs (k_wgrad_reduce's adds, in its
// order), stores it (bf16: XB) for the apply pass, and accumulates the two gradient sums from the STORED value with k_bn_partial<1>'s
// thread mapping, row order and fold: bit-identical to reduce-then-statistics, one dependent launch fewer per split layer.
template <int V, int XB>
__global__ __launch_bounds__(BN_PT) void k_bn_partial1_slabs(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ slabs, int S, size_t elems, int acc,
                                                           float* __restrict__ dy, int n, int c, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, int qpad, int rows_per_block,
                                                           float* __restrict__ part) {
  __shared__ __align__(16) float s0[BN_PT * V];
  __shared__ __align__(16) float s1[BN_PT * V];
  const int cq = c / V;
  const int qd = threadIdx.x % qpad;
  const int rg = threadIdx.x / qpad;
  const int nrg = BN_PT / qpad;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  float a0[V], a1[V];
#pragma unroll
  for (int j = 0; j < V; ++j) a0[j] = a1[j] = 0.f;
  if (qd < cq) {
    float mu[V], is[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      mu[j] = mean[qd * V + j];
      is[j] = invstd[qd * V + j];
    }
    for (int r = r0 + rg; r < r1; r += nrg) {
      const size_t off = (size_t)r * c + (size_t)qd * V;
      float xv[V], yv[V], v[V], old[V];
      sl_ld<V, XB>(x, off, xv);
      sl_ld<V, XB>(y, off, yv);
      if (acc) sl_ld<V, XB>(dy, off, old);
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = 0.f;
      int sidx = 0;
      for (; sidx + 4 <= S; sidx += 4) {
        float4 t[4][V / 4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < V / 4; ++q) t[u][q] = *reinterpret_cast<const float4*>(slabs + (size_t)(sidx + u) * elems + off + 4 * q);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < V / 4; ++q) {
            v[4 * q] += t[u][q].x; v[4 * q + 1] += t[u][q].y; v[4 * q + 2] += t[u][q].z; v[4 * q + 3] += t[u][q].w;
          }
      }
      for (; sidx < S; ++sidx)
#pragma unroll
        for (int q = 0; q < V / 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(slabs + (size_t)sidx * elems + off + 4 * q);
          v[4 * q] += t.x; v[4 * q + 1] += t.y; v[4 * q + 2] += t.z; v[4 * q + 3] += t.w;
        }
      if (acc) {
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = old[j] + v[j];          // (old + sum): k_wgrad_reduce's order
      }
      if constexpr (XB) {
#pragma unroll
        for (int j = 0; j < V; j += 2) {
          const unsigned pk = irx_pk_bf16(v[j], v[j + 1]);
          v[j] = __uint_as_float(pk << 16);
          v[j + 1] = __uint_as_float(pk & 0xffff0000u);
        }
      }
      sl_st<V, XB>(dy, off, v);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float gval = v[j];
        if (!(yv[j] > 0.f)) gval = 0.f;
        a0[j] += gval;
        a1[j] += gval * ((xv[j] - mu[j]) * is[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) {
    s0[threadIdx.x * V + j] = a0[j];
    s1[threadIdx.x * V + j] = a1[j];
  }
  __syncthreads();
  const int jw = nrg < V ? nrg : V;              // (k_bn_partial's fold: one channel per thread)
  if (rg < jw && qd < cq) {
    for (int j = rg; j < V; j += jw) {
      float t0 = 0.f, t1 = 0.f;
      for (int g2 = 0; g2 < nrg; ++g2) {
        t0 += s0[(g2 * qpad + qd) * V + j];
        t1 += s1[(g2 * qpad + qd) * V + j];
      }
      part[((size_t)blockIdx.x * 2 + 0) * c + qd * V + j] = t0;
      part[((size_t)blockIdx.x * 2 + 1) * c + qd * V + j] = t1;
    }
  }
}

__global__ void k_bn_slab_reduce_acc(const float* __restrict__ part, int S, size_t elems, float* __restrict__ out, int out_bf, int acc) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elems) return;
  float s = 0.f;
  for (int j = 0; j < S; ++j) s += part[(size_t)j * elems + i];
  if (out_bf) {
    unsigned short* o = reinterpret_cast<unsigned short*>(out) + i;
    if (acc) s = __uint_as_float((unsigned)*o << 16) + s;
    *o = (unsigned short)(irx_pk_bf16(s, 0.f) & 0xffffu);
  } else {
    out[i] = acc ? out[i] + s : s;
  }
}

// a zeroed ticket counter for one k_bn_partial launch that folds its own partials (see the kernel's tail), or NULL when that
// is switched off (IRX_BN_LASTBLOCK=0: the separate k_bn_finalize launch) or the channel count does not fit its fold
static unsigned* bn_counter(int c) {
  // OFF by default — a measured negative result (tools/bn_microbench.py, MI355X): the two agent-scope fences cost more than the
  // launch they replace (gfx950 writes back / invalidates a whole L2 per fence): bf16 fwd 488 800 x 32: 39.9 -> 57.8 us,
  // 81 261 x 128: 22.2 -> 43.8 us, 2 244 x 128: 12.2 -> 17.3 us. IRX_BN_LASTBLOCK=1 enables it.
  static const bool on = getenv("IRX_BN_LASTBLOCK") && atoi(getenv("IRX_BN_LASTBLOCK")) != 0;
  if (!on || c % 4 != 0 || c / 4 > BN_PT) return nullptr;
  static std::atomic<unsigned*> base[16];
  static std::atomic<unsigned> next{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  unsigned* b = base[dev].load(std::memory_order_acquire);
  if (!b) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_bn_counters)) != hipSuccess || !p) return nullptr;
    b = (unsigned*)p;
    base[dev].store(b, std::memory_order_release);
  }
  return b + (next.fetch_add(1, std::memory_order_relaxed) % BN_NCOUNTERS);
}

// tensors up to this many bytes (n * c * element size of x) take the slice kernels: IRX_BN_SLICE_BYTES (dev knob; 0 switches
// them off). Default 640 KB — measured (tools/bn_microbench.py, MI355X, bf16, us per call, three launches | slice kernel):
//   2 244 x 128: fwd 12.2 | 10.4, bwd 14.2 | 12.0      4 636 x 128: 12.9 | 14.0, 15.0 | 18.3      8 990 x 128: 13.1 | 25.0, 15.6 | 59.4
//   20 267 x 128: 14.7 | 77.5      81 261 x 128: 22.2 | 298
// i.e. a workgroup walking one 16-byte column streams ~12 GB/s (every lane of a load touches its own cache line and uses 16 of its
// 128 bytes), so the single launch only pays below ~3 k rows, where three dependent launches (~4 us each) are all there is.
static size_t bn_slice_max_bytes() {
  static const long v = getenv("IRX_BN_SLICE_BYTES") ? atol(getenv("IRX_BN_SLICE_BYTES")) : (640L << 10);
  return v > 0 ? (size_t)v : 0;
}
static bool bn_slice_ok(int n, int c, int x_bf) {
  const int v = x_bf ? 8 : 4;
  return n > 0 && c % v == 0 && (size_t)n * c * (x_bf ? 2 : 4) <= bn_slice_max_bytes();
}

// ------------------------------------------------------------------------------ C entry ------
extern "C" size_t irx_bn_workspace_bytes(int n, int c) {
  if (n <= 0 || c <= 0) return 0;
  return (size_t)irx_cdiv(n, bn_rows(n, c)) * 2 * (size_t)c * sizeof(float);
}

static int bn_check(const char* who, int n, int c, const void* ws, size_t ws_bytes) {
  IRX_REQUIRE(n >= 0 && c >= 1 && c <= 1024, "%s: unsupported sizes n=%d c=%d", who, n, c);
  const int v = (c % 4 == 0) ? 4 : 1;
  IRX_REQUIRE(c / v <= 256, "%s: c=%d too large for the statistics kernel", who, c);
  if (n > 0 && (ws == nullptr || ws_bytes < irx_bn_workspace_bytes(n, c))) {
    irx_set_error("%s: workspace %zu < %zu", who, ws_bytes, irx_bn_workspace_bytes(n, c));
    return IRX_ERR_WORKSPACE;
  }
  return IRX_OK;
}

static inline int row_grid(int n, int qpad) {
  const int rpp = 256 / qpad;
  long long b = ((long long)n + rpp - 1) / rpp;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int irx_bn_stats(const float* x, int n, int c, float eps, float momentum, float* mean,
                            float* invstd, float* running_mean, float* running_var, void* workspace,
                            size_t workspace_bytes, void* stream) {
  return irx_bn_stats_t(x, n, c, eps, momentum, mean, invstd, running_mean, running_var, workspace, workspace_bytes,
                        stream, 0);
}

static int bn_bf_ok(const char* who, bool any_bf, bool v4) {
  IRX_REQUIRE(!any_bf || v4, "%s: bf16 tensors need c %% 4 == 0 and 16-byte aligned pointers", who);
  return IRX_OK;
}

// Dev-only, TIMING ONLY (results wrong): IRX_BN_ABL bit 0 = no forward statistics pass, 1 = no forward apply pass,
// 2 = no backward statistics pass, 3 = no backward apply pass — what a perfect fusion of that pass into a neighbouring
// kernel could buy at most.
static int bn_abl() {
  static const int v = getenv("IRX_BN_ABL") ? atoi(getenv("IRX_BN_ABL")) : 0;
  return v;
}

int irx_bn_stats_t(const float* x, int n, int c, float eps, float momentum, float* mean, float* invstd,
                   float* running_mean, float* running_var, void* workspace, size_t workspace_bytes, void* stream,
                   int x_bf) {
  int rc = bn_check("irx_bn_stats", n, c, workspace, workspace_bytes);
  if (rc) return rc;
  if (n == 0) return IRX_OK;
  IRX_REQUIRE(x && mean && invstd, "irx_bn_stats: null pointer");
  const int nblk = irx_cdiv(n, bn_rows(n, c));
  float* part = (float*)workspace;
  const bool v4 = (c % 4 == 0) && (((uintptr_t)x & 15) == 0);
  const BnTy ty = {x_bf, 0, 0, 0, 0};
  rc = bn_bf_ok("irx_bn_stats", x_bf != 0, v4);
  if (rc) return rc;
  BnFin fin;
  if (v4 && (fin.counter = bn_counter(c)) != nullptr) {
    fin.eps = eps; fin.momentum = momentum; fin.out0 = mean; fin.out1 = invstd;
    fin.running_mean = running_mean; fin.running_var = running_var;
  }
  if (bn_abl() & 1) {
  } else if (v4 && x_bf && c % 8 == 0)
    k_bn_partial<0, 8, false><<<nblk, BN_PT, 0, S(stream)>>>(x, nullptr, nullptr, n, c, nullptr, nullptr, 0,
                                                   next_pow2(c / 8), bn_rows(n, c), part, ty, nullptr, nullptr, fin);
  else if (v4 && x_bf)
    k_bn_partial<0, 4, true><<<nblk, BN_PT, 0, S(stream)>>>(x, nullptr, nullptr, n, c, nullptr, nullptr, 0,
                                                         next_pow2(c / 4), bn_rows(n, c), part, ty, nullptr, nullptr, fin);
  else if (v4)
    k_bn_partial<0, 4, false><<<nblk, BN_PT, 0, S(stream)>>>(x, nullptr, nullptr, n, c, nullptr, nullptr, 0,
                                                          next_pow2(c / 4), bn_rows(n, c), part, ty, nullptr, nullptr, fin);
  else {
    IRX_REQUIRE(c <= 256, "irx_bn_stats: c=%d needs c %% 4 == 0 or c <= 256", c);
    k_bn_partial<0, 1, false><<<nblk, BN_PT, 0, S(stream)>>>(x, nullptr, nullptr, n, c, nullptr, nullptr, 0,
                                                   next_pow2(c), bn_rows(n, c), part, ty, nullptr, nullptr);
  }
  IRX_CHECK_LAUNCH("irx_bn_stats(partial)");
  if (fin.counter && !(bn_abl() & 1)) return IRX_OK;       // folded by the last workgroup of the launch above
  k_bn_finalize<0><<<irx_cdiv(c, 32), 32 * BN_FIN_SLICES, 0, S(stream)>>>(part, nblk, n, c, eps, momentum, mean,
                                                          invstd, running_mean, running_var);
  IRX_CHECK_LAUNCH("irx_bn_stats(finalize)");
  return IRX_OK;
}

// ---- sync BatchNorm (statistics over the rows of ALL ranks): the same kernels, with the cross-rank fold of the sums left
// to the caller (torch.distributed in instancerefer_amd/sparse/functional.py) between the two halves ----
extern "C" int irx_bn_sums(const float* x, int n, int c, double* sums, void* workspace, size_t workspace_bytes,
                           void* stream) {
  int rc = bn_check("irx_bn_sums", n, c, workspace, workspace_bytes);
  if (rc) return rc;
  IRX_REQUIRE(sums, "irx_bn_sums: null pointer");
  if (n == 0) {
    IRX_CHECK_HIP(hipMemsetAsync(sums, 0, 2 * (size_t)c * sizeof(double), S(stream)), "irx_bn_sums(memset)");
    return IRX_OK;
  }
  IRX_REQUIRE(x, "irx_bn_sums: null pointer");
  const int nblk = irx_cdiv(n, bn_rows(n, c));
  float* part = (float*)workspace;
  const BnTy ty = {0, 0, 0, 0, 0};
  if ((c % 4 == 0) && (((uintptr_t)x & 15) == 0))
    k_bn_partial<0, 4, false><<<nblk, BN_PT, 0, S(stream)>>>(x, nullptr, nullptr, n, c, nullptr, nullptr, 0,
                                                          next_pow2(c / 4), bn_rows(n, c), part, ty, nullptr, nullptr);
  else {
    IRX_REQUIRE(c <= 256, "irx_bn_sums: c=%d needs c %% 4 == 0 or c <= 256", c);
    k_bn_partial<0, 1, false><<<nblk, BN_PT, 0, S(stream)>>>(x, nullptr, nullptr, n, c, nullptr, nullptr, 0,
                                                          next_pow2(c), bn_rows(n, c), part, ty, nullptr, nullptr);
  }
  IRX_CHECK_LAUNCH("irx_bn_sums(partial)");
  k_bn_finalize<2><<<irx_cdiv(c, 32), 32 * BN_FIN_SLICES, 0, S(stream)>>>(
      part, nblk, n, c, 0.f, 0.f, reinterpret_cast<float*>(sums), reinterpret_cast<float*>(sums + c), nullptr, nullptr);
  IRX_CHECK_LAUNCH("irx_bn_sums(finalize)");
  return IRX_OK;
}

// irx_bn_sums with an element type for x (0 = float32, 1 = bf16) — the executor's sync-BatchNorm path; count_slot != NULL:
// the row count n is stored there as a double (the caller folds it with the sums: sums[2 c])
__global__ void k_set_double(double* p, double v) { *p = v; }
__global__ void k_copy2_f32(const float* __restrict__ a, const float* __restrict__ b, int c, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < c) { out[i] = a[i]; out[c + i] = b[i]; }
}
int irx_bn_sums_t(const float* x, int n, int c, double* sums, void* workspace, size_t workspace_bytes, void* stream, int x_bf,
                  double* count_slot) {
  int rc = bn_check("irx_bn_sums", n, c, workspace, workspace_bytes);
  if (rc) return rc;
  IRX_REQUIRE(sums, "irx_bn_sums: null pointer");
  if (count_slot) {
    k_set_double<<<1, 1, 0, S(stream)>>>(count_slot, (double)n);
    IRX_CHECK_LAUNCH("irx_bn_sums(count)");
  }
  if (n == 0) {
    IRX_CHECK_HIP(hipMemsetAsync(sums, 0, 2 * (size_t)c * sizeof(double), S(stream)), "irx_bn_sums(memset)");
    return IRX_OK;
  }
  if (!x_bf) return irx_bn_sums(x, n, c, sums, workspace, workspace_bytes, stream);
  const int nblk = irx_cdiv(n, bn_rows(n, c));
  float* part = (float*)workspace;
  const bool v4 = (c % 4 == 0) && (((uintptr_t)x & 15) == 0);
  const BnTy ty = {1, 0, 0, 0, 0};
  rc = bn_bf_ok("irx_bn_sums", true, v4);
  if (rc) return rc;
  if (c % 8 == 0)
    k_bn_partial<0, 8, false><<<nblk, BN_PT, 0, S(stream)>>>(x, nullptr, nullptr, n, c, nullptr, nullptr, 0,
                                                          next_pow2(c / 8), bn_rows(n, c), part, ty, nullptr, nullptr);
  else
    k_bn_partial<0, 4, true><<<nblk, BN_PT, 0, S(stream)>>>(x, nullptr, nullptr, n, c, nullptr, nullptr, 0,
                                                         next_pow2(c / 4), bn_rows(n, c), part, ty, nullptr, nullptr);
  IRX_CHECK_LAUNCH("irx_bn_sums(partial)");
  k_bn_finalize<2><<<irx_cdiv(c, 32), 32 * BN_FIN_SLICES, 0, S(stream)>>>(
      part, nblk, n, c, 0.f, 0.f, reinterpret_cast<float*>(sums), reinterpret_cast<float*>(sums + c), nullptr, nullptr);
  IRX_CHECK_LAUNCH("irx_bn_sums(finalize)");
  return IRX_OK;
}
int irx_bn_pack_sums(const float* sum_g, const float* sum_gx, int c, float* out, void* stream) {
  k_copy2_f32<<<irx_cdiv(c, 128), 128, 0, S(stream)>>>(sum_g, sum_gx, c, out);
  IRX_CHECK_LAUNCH("irx_bn_backward(pack sums)");
  return IRX_OK;
}

extern "C" int irx_bn_stats_from_sums(const double* sums, double count, int c, float eps, float momentum, float* mean,
                                      float* invstd, float* running_mean, float* running_var, void* stream) {
  IRX_REQUIRE(sums && mean && invstd && c >= 1, "irx_bn_stats_from_sums: bad arguments");
  k_bn_from_sums<<<irx_cdiv(c, 128), 128, 0, S(stream)>>>(sums, count, c, eps, momentum, mean, invstd, running_mean,
                                                         running_var);
  IRX_CHECK_LAUNCH("irx_bn_stats_from_sums");
  return IRX_OK;
}

extern "C" int irx_bn_backward_sums(const float* x, const float* y, const float* dy, int n, int c, const float* mean,
                                    const float* invstd, int relu, float* dgamma, float* dbeta, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  return irx_bn_backward_t(x, y, dy, n, c, mean, invstd, nullptr, relu, nullptr, dgamma, dbeta, nullptr, workspace,
                           workspace_bytes, stream, 0, 0, 0, 0, 0, 1);
}

extern "C" int irx_bn_backward_apply(const float* x, const float* y, const float* dy, int n, int c, const float* mean,
                                     const float* invstd, const float* gamma, int relu, const float* sum_g,
                                     const float* sum_gx, double count, const double* count_dev, float* dx,
                                     float* dresidual, void* stream) {
  return irx_bn_backward_t(x, y, dy, n, c, mean, invstd, gamma, relu, dx, nullptr, nullptr, dresidual, nullptr, 0, stream,
                           0, 0, 0, 0, 0, 2, sum_g, sum_gx, count, count >= 1.0 ? nullptr : count_dev);
}

extern "C" int irx_bn_apply(const float* x, int n, int c, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, const float* residual, int relu,
                            float* y, void* stream) {
  return irx_bn_apply_t(x, n, c, mean, invstd, gamma, beta, residual, relu, y, stream, 0, 0, 0);
}

int irx_bn_apply_t(const float* x, int n, int c, const float* mean, const float* invstd, const float* gamma,
                   const float* beta, const float* residual, int relu, float* y, void* stream, int x_bf, int res_bf,
                   int y_bf) {
  IRX_REQUIRE(n >= 0 && c >= 1, "irx_bn_apply: bad sizes");
  if (n == 0) return IRX_OK;
  IRX_REQUIRE(x && mean && invstd && gamma && beta && y, "irx_bn_apply: null pointer");
  const bool v4 = (c % 4 == 0) && (c / 4 <= 256) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)y & 15) == 0) &&
                  (residual == nullptr || ((uintptr_t)residual & 15) == 0);
  const BnTy ty = {x_bf, y_bf, 0, 0, res_bf};
  int rc = bn_bf_ok("irx_bn_apply", (x_bf | res_bf | y_bf) != 0, v4);
  if (rc) return rc;
  if (bn_abl() & 2) return IRX_OK;
  if (v4 && x_bf && y_bf && (!residual || res_bf) && c % 8 == 0) {
    const int qpad = next_pow2(c / 8);
    if (residual && relu) k_bn_apply<8, false, true, true><<<row_grid(n, qpad), 256, 0, S(stream)>>>(x, n, c, qpad, mean, invstd, gamma, beta, residual, relu, y, ty);
    else if (residual && !relu) k_bn_apply<8, false, true, false><<<row_grid(n, qpad), 256, 0, S(stream)>>>(x, n, c, qpad, mean, invstd, gamma, beta, residual, relu, y, ty);
    else if (!residual && relu) k_bn_apply<8, false, false, true><<<row_grid(n, qpad), 256, 0, S(stream)>>>(x, n, c, qpad, mean, invstd, gamma, beta, residual, relu, y, ty);
    else if (!residual && !relu) k_bn_apply<8, false, false, false><<<row_grid(n, qpad), 256, 0, S(stream)>>>(x, n, c, qpad, mean, invstd, gamma, beta, residual, relu, y, ty);
  } else if (v4 && (x_bf | res_bf | y_bf)) {
    const int qpad = next_pow2(c / 4);
    if (residual && relu) k_bn_apply<4, true, true, true><<<row_grid(n, qpad), 256, 0, S(stream)>>>(x, n, c, qpad, mean, invstd, gamma, beta, residual, relu, y, ty);
    else if (residual && !relu) k_bn_apply<4, true, true, false><<<row_grid(n, qpad), 256, 0, S(stream)>>>(x, n, c, qpad, mean, invstd, gamma, beta, residual, relu, y, ty);
    else if (!residual && relu) k_bn_apply<4, true, false, true><<<row_grid(n, qpad), 256, 0, S(stream)>>>(x, n, c, qpad, mean, invstd, gamma, beta, residual, relu, y, ty);
    else if (!residual && !relu) k_bn_apply<4, true, false, false><<<row_grid(n, qpad), 256, 0, S(stream)>>>(x, n, c, qpad, mean, invstd, gamma, beta, residual, relu, y, ty);
  } else if (v4) {
    const int qpad = next_pow2(c / 4);
    if (residual && relu) k_bn_apply<4, false, true, true><<<row_grid(n, qpad), 256, 0, S(stream)>>>(x, n, c, qpad, mean, invstd, gamma, beta, residual, relu, y, ty);
    else if (residual && !relu) k_bn_apply<4, false, true, false><<<row_grid(n, qpad), 256, 0, S(stream)>>>(x, n, c, qpad, mean, invstd, gamma, beta, residual, relu, y, ty);
    else if (!residual && relu) k_bn_apply<4, false, false, true><<<row_grid(n, qpad), 256, 0, S(stream)>>>(x, n, c, qpad, mean, invstd, gamma, beta, residual, relu, y, ty);
    else if (!residual && !relu) k_bn_apply<4, false, false, false><<<row_grid(n, qpad), 256, 0, S(stream)>>>(x, n, c, qpad, mean, invstd, gamma, beta, residual, relu, y, ty);
  } else {
    IRX_REQUIRE(c <= 256, "irx_bn_apply: c=%d needs c %% 4 == 0 or c <= 256", c);
    const int qpad = next_pow2(c);
    if (residual && relu) k_bn_apply<1, false, true, true><<<row_grid(n, qpad), 256, 0, S(stream)>>>(x, n, c, qpad, mean, invstd, gamma, beta, residual, relu, y, ty);
    else if (residual && !relu) k_bn_apply<1, false, true, false><<<row_grid(n, qpad), 256, 0, S(stream)>>>(x, n, c, qpad, mean, invstd, gamma, beta, residual, relu, y, ty);
    else if (!residual && relu) k_bn_apply<1, false, false, true><<<row_grid(n, qpad), 256, 0, S(stream)>>>(x, n, c, qpad, mean, invstd, gamma, beta, residual, relu, y, ty);
    else if (!residual && !relu) k_bn_apply<1, false, false, false><<<row_grid(n, qpad), 256, 0, S(stream)>>>(x, n, c, qpad, mean, invstd, gamma, beta, residual, relu, y, ty);
  }
  IRX_CHECK_LAUNCH("irx_bn_apply");
  return IRX_OK;
}

// Train-mode BatchNorm (+ shortcut) + ReLU in one call: statistics (mean / invstd written for the backward, running statistics
// updated) and the apply pass. Small tensors take ONE launch (k_bn_slice_fwd), the others irx_bn_stats_t + irx_bn_apply_t.
int irx_bn_forward_t(const float* x, int n, int c, float eps, float momentum, const float* gamma, const float* beta,
                     const float* residual, int relu, float* mean, float* invstd, float* running_mean, float* running_var,
                     float* y, void* workspace, size_t workspace_bytes, void* stream, int x_bf, int res_bf, int y_bf) {
  const bool types_ok = (!x_bf && !y_bf && (!residual || !res_bf)) || (x_bf && (!residual || res_bf));
  const bool aligned = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15) == 0;
  if (relu && types_ok && aligned && bn_slice_ok(n, c, x_bf) && !(bn_abl() & 3)) {
    IRX_REQUIRE(x && y && gamma && beta && mean && invstd, "irx_bn_forward: null pointer");
    const hipStream_t st = S(stream);
#define BN_SL_FWD(V_, XB_, YB_, RB_)                                                                                          \
  do {                                                                                                                        \
    if (residual) k_bn_slice_fwd<V_, XB_, YB_, RB_, true><<<c / V_, BN_SL_PT, 0, st>>>(x, n, c, eps, momentum, gamma, beta,  \
                                                            residual, y, mean, invstd, running_mean, running_var);          \
    else k_bn_slice_fwd<V_, XB_, YB_, RB_, false><<<c / V_, BN_SL_PT, 0, st>>>(x, n, c, eps, momentum, gamma, beta, nullptr, \
                                                            y, mean, invstd, running_mean, running_var);                    \
  } while (0)
    if (!x_bf) BN_SL_FWD(4, 0, 0, 0);
    else if (y_bf) BN_SL_FWD(8, 1, 1, 1);
    else BN_SL_FWD(8, 1, 0, 1);
#undef BN_SL_FWD
    IRX_CHECK_LAUNCH("irx_bn_forward(slice)");
    return IRX_OK;
  }
  int rc = irx_bn_stats_t(x, n, c, eps, momentum, mean, invstd, running_mean, running_var, workspace, workspace_bytes, stream,
                          x_bf);
  if (rc) return rc;
  return irx_bn_apply_t(x, n, c, mean, invstd, gamma, beta, residual, relu, y, stream, x_bf, res_bf, y_bf);
}

extern "C" int irx_bn_forward(const float* x, int n, int c, float eps, float momentum, const float* gamma, const float* beta,
                              const float* residual, int relu, float* mean, float* invstd, float* running_mean,
                              float* running_var, float* y, void* workspace, size_t workspace_bytes, void* stream) {
  return irx_bn_forward_t(x, n, c, eps, momentum, gamma, beta, residual, relu, mean, invstd, running_mean, running_var, y,
                          workspace, workspace_bytes, stream, 0, 0, 0);
}

// irx_bn_forward_t for a convolution output that still is S offset-split slabs (IrxStore::slabs_out): the statistics pass folds them
// and writes x_out (the tensor the backward reads); a tensor small enough for the slice kernel is reduced first and takes that path.
__global__ void k_bn_slab_reduce(const float* __restrict__ part, int S, size_t elems, float* __restrict__ out, int out_bf) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elems) return;
  float s = 0.f;
  for (int j = 0; j < S; ++j) s += part[(size_t)j * elems + i];
  if (out_bf) reinterpret_cast<unsigned short*>(out)[i] = (unsigned short)(irx_pk_bf16(s, 0.f) & 0xffffu);
  else out[i] = s;
}
int irx_bn_forward_slabs_t(const float* slabs, int S, int n, int c, float eps, float momentum, const float* gamma, const float* beta,
                           const float* residual, int relu, float* mean, float* invstd, float* running_mean, float* running_var,
                           float* x_out, float* y, void* workspace, size_t workspace_bytes, void* stream, int x_bf, int res_bf,
                           int y_bf) {
  IRX_REQUIRE(slabs && S >= 1 && x_out, "irx_bn_forward_slabs: bad arguments");
  const size_t elems = (size_t)n * c;
  const int v = x_bf ? 8 : 4;
  const bool fold = n > 0 && c % v == 0 && (((uintptr_t)slabs | (uintptr_t)x_out) & 15) == 0 && !(relu && bn_slice_ok(n, c, x_bf)) &&
                    !(bn_abl() & 1);
  if (!fold) {
    if (n > 0) {
      k_bn_slab_reduce<<<irx_cdiv((long long)elems, 256), 256, 0, S_(stream)>>>(slabs, S, elems, x_out, x_bf);
      IRX_CHECK_LAUNCH("irx_bn_forward_slabs(reduce)");
    }
    return irx_bn_forward_t(x_out, n, c, eps, momentum, gamma, beta, residual, relu, mean, invstd, running_mean, running_var, y,
                            workspace, workspace_bytes, stream, x_bf, res_bf, y_bf);
  }
  int rc = bn_check("irx_bn_forward_slabs", n, c, workspace, workspace_bytes);
  if (rc) return rc;
  const int nblk = irx_cdiv(n, bn_rows(n, c));
  float* part = (float*)workspace;
  if (x_bf) k_bn_partial_slabs<8, 1><<<nblk, BN_PT, 0, S_(stream)>>>(slabs, S, elems, n, c, next_pow2(c / 8), bn_rows(n, c), x_out, part);
  else k_bn_partial_slabs<4, 0><<<nblk, BN_PT, 0, S_(stream)>>>(slabs, S, elems, n, c, next_pow2(c / 4), bn_rows(n, c), x_out, part);
  IRX_CHECK_LAUNCH("irx_bn_forward_slabs(partial)");
  k_bn_finalize<0><<<irx_cdiv(c, 32), 32 * BN_FIN_SLICES, 0, S_(stream)>>>(part, nblk, n, c, eps, momentum, mean, invstd, running_mean,
                                                                          running_var);
  IRX_CHECK_LAUNCH("irx_bn_forward_slabs(finalize)");
  return irx_bn_apply_t(x_out, n, c, mean, invstd, gamma, beta, residual, relu, y, stream, x_bf, res_bf, y_bf);
}

extern "C" int irx_bn_backward(const float* x, const float* y, const float* dy, int n, int c,
                               const float* mean, const float* invstd, const float* gamma, int relu,
                               float* dx, float* dgamma, float* dbeta, float* dresidual,
                               void* workspace, size_t workspace_bytes, void* stream) {
  return irx_bn_backward_t(x, y, dy, n, c, mean, invstd, gamma, relu, dx, dgamma, dbeta, dresidual, workspace,
                           workspace_bytes, stream, 0, 0, 0, 0, 0);
}

// x_bf / y_bf / dy_bf: types of the saved conv output, the layer output and the incoming gradient; dx_bf / dres_bf:
// types of the two gradients written
int irx_bn_backward_t(const float* x, const float* y, const float* dy, int n, int c, const float* mean,
                      const float* invstd, const float* gamma, int relu, float* dx, float* dgamma, float* dbeta,
                      float* dresidual, void* workspace, size_t workspace_bytes, void* stream, int x_bf, int y_bf,
                      int dy_bf, int dx_bf, int dres_bf, int phases, const float* all_sum_g, const float* all_sum_gx,
                      double all_count, const double* count_dev, const float* beta, IrxDySlabs dys) {
  // beta != NULL (with relu): the forward pass had NO shortcut, i.e. y = relu(fma(x, invstd * gamma, fma(-mean, invstd * gamma,
  // beta))): the ReLU mask is recomputed from x in both passes and y is not read (the encoder executor passes it for the 9
  // of 13 layers without a shortcut; the C-ABI entry points below keep reading y)
  // phases: 1 = the reduction only (dbeta = sum g, dgamma = sum g xhat over THESE rows), 2 = the apply pass only, with the
  // sums (all_sum_g, all_sum_gx) and row count (all_count) it is given — sync BatchNorm folds the ranks' sums in between;
  // 3 = both, on the local sums (everything else)
  int rc = (phases & 1) ? bn_check("irx_bn_backward", n, c, workspace, workspace_bytes) : bn_check("irx_bn_backward", 0, c, nullptr, 0);
  if (rc) return rc;
  IRX_REQUIRE(phases >= 1 && phases <= 3, "irx_bn_backward: phases = %d", phases);
  IRX_REQUIRE(!(phases & 1) || (dgamma && dbeta), "irx_bn_backward: null dgamma/dbeta");
  IRX_REQUIRE(phases != 2 || (all_sum_g && all_sum_gx && (all_count >= 1.0 || count_dev)), "irx_bn_backward(apply): sums / count missing");
  if (phases != 2) count_dev = nullptr;
  if (n == 0) {
    if (!(phases & 1)) return IRX_OK;
    IRX_CHECK_HIP(hipMemsetAsync(dgamma, 0, c * sizeof(float), S(stream)), "irx_bn_backward(memset)");
    IRX_CHECK_HIP(hipMemsetAsync(dbeta, 0, c * sizeof(float), S(stream)), "irx_bn_backward(memset)");
    return IRX_OK;
  }
  IRX_REQUIRE(x && dy && mean && invstd && (gamma || phases == 1) && (dx || phases == 1), "irx_bn_backward: null pointer");
  IRX_REQUIRE(!relu || y, "irx_bn_backward: relu needs y");
  bool fold_dy = false;
  if (dys.slabs) {
    // dy arrives as the data-gradient convolution's offset-split slabs: folded by the statistics pass below when the call has the
    // plain shape (local statistics, ReLU, one element type), materialised by a reduce launch of its own otherwise
    const int vv = x_bf ? 8 : 4;
    fold_dy = phases == 3 && relu && x_bf == y_bf && x_bf == dy_bf && c % vv == 0 && !bn_slice_ok(n, c, x_bf) &&
              ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)dy | (uintptr_t)dys.slabs) & 15) == 0) && !(bn_abl() & 4);
    if (!fold_dy) {
      const size_t elems = (size_t)n * c;
      k_bn_slab_reduce_acc<<<irx_cdiv((long long)elems, 256), 256, 0, S_(stream)>>>(dys.slabs, dys.S, elems, const_cast<float*>(dy),
                                                                                  dy_bf, dys.acc);
      IRX_CHECK_LAUNCH("irx_bn_backward(slab reduce)");
    }
  }
  const int nblk = irx_cdiv(n, bn_rows(n, c));
  float* part = (float*)workspace;
  const bool v4 = (c % 4 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)dy & 15) == 0) &&
                  (dx == nullptr || ((uintptr_t)dx & 15) == 0) && (!relu || ((uintptr_t)y & 15) == 0) &&
                  (dresidual == nullptr || ((uintptr_t)dresidual & 15) == 0);
  const BnTy ty = {x_bf, y_bf, dy_bf, dx_bf, dres_bf};
  const float* mk_beta = (relu && beta && gamma && !dresidual) ? beta : nullptr;
  const float* mk_gamma = mk_beta ? gamma : nullptr;
  rc = bn_bf_ok("irx_bn_backward", (x_bf | y_bf | dy_bf | dx_bf | dres_bf) != 0, v4);
  if (rc) return rc;
  {
    // one launch for a tensor that stays on-die (k_bn_slice_bwd): both passes of the local-statistics case
    const bool all_f32 = !(x_bf | y_bf | dy_bf | dx_bf | dres_bf);
    const bool sl_types = all_f32 || (x_bf && dx_bf && (!dresidual || dres_bf) && (dy_bf == y_bf || mk_beta));
    if (phases == 3 && relu && dx && gamma && sl_types && v4 && !dresidual && bn_slice_ok(n, c, x_bf) && !(bn_abl() & 12)) {
      // (with a shortcut gradient to write the slice kernel measured slower even at 2 244 rows: 16.1 vs 14.5 us)
      const hipStream_t st = S(stream);
#define BN_SL_BWD(V_, XB_, YB_)                                                                                               \
  do {                                                                                                                        \
    if (mk_beta) k_bn_slice_bwd<V_, XB_, YB_, true, false><<<c / V_, BN_SL_PT, 0, st>>>(x, y, dy, n, c, mean, invstd, gamma, \
                                                                                        mk_beta, dx, nullptr, dgamma, dbeta); \
    else if (dresidual) k_bn_slice_bwd<V_, XB_, YB_, false, true><<<c / V_, BN_SL_PT, 0, st>>>(x, y, dy, n, c, mean, invstd,  \
                                                                                 gamma, nullptr, dx, dresidual, dgamma, dbeta); \
    else k_bn_slice_bwd<V_, XB_, YB_, false, false><<<c / V_, BN_SL_PT, 0, st>>>(x, y, dy, n, c, mean, invstd, gamma, nullptr, \
                                                                                 dx, nullptr, dgamma, dbeta);                \
  } while (0)
      if (all_f32) BN_SL_BWD(4, 0, 0);
      else if (dy_bf) BN_SL_BWD(8, 1, 1);
      else BN_SL_BWD(8, 1, 0);
#undef BN_SL_BWD
      IRX_CHECK_LAUNCH("irx_bn_backward(slice)");
      return IRX_OK;
    }
  }
  const bool any_bf = (x_bf | y_bf | dy_bf | dx_bf | dres_bf) != 0;
  const bool v8 = v4 && x_bf && (y_bf || !relu) && dy_bf && dx_bf && (!dresidual || dres_bf) && c % 8 == 0;
  const bool rm = mk_beta != nullptr;            // mask recomputed from x: separate instantiations (see k_bn_partial)
  // Measured per kernel (rocprofv3, same box, the 9 shortcut-free layers of both encoders): the apply pass gains from not
  // reading y (14.4 -> 11.7 us average), the statistics pass LOSES (15.7 -> 18.3 us: it is bound by the latency of a row
  // group's loads, not by their bytes, and the recomputation lengthens the dependent chain behind them) — so only the apply
  // pass recomputes the mask; the statistics pass keeps reading y.
  static const bool rm_stats = getenv("IRX_BN_REMASK_STATS") && atoi(getenv("IRX_BN_REMASK_STATS")) != 0;   // dev A/B knob
  BnFin fin;
  if ((phases & 1) && v4 && (fin.counter = bn_counter(c)) != nullptr) {
    fin.out0 = dbeta; fin.out1 = dgamma;
  }
#define BN_PARTIAL1(V_, TY_, QP_)                                                                                          \
  do {                                                                                                                     \
    if (rm && rm_stats) k_bn_partial<1, V_, TY_, true, true><<<nblk, BN_PT, 0, S(stream)>>>(x, y, dy, n, c, mean, invstd, relu, QP_,     \
                                                                        bn_rows(n, c), part, ty, mk_gamma, mk_beta, fin);  \
    else if (relu) k_bn_partial<1, V_, TY_, false, true><<<nblk, BN_PT, 0, S(stream)>>>(x, y, dy, n, c, mean, invstd, relu, QP_, \
                                                                      bn_rows(n, c), part, ty, nullptr, nullptr, fin);     \
    else k_bn_partial<1, V_, TY_, false, false><<<nblk, BN_PT, 0, S(stream)>>>(x, y, dy, n, c, mean, invstd, relu, QP_,      \
                                                                      bn_rows(n, c), part, ty, nullptr, nullptr, fin);     \
  } while (0)
  if (!(phases & 1) || (bn_abl() & 4)) {
  } else if (fold_dy) {
    const size_t elems = (size_t)n * c;
    fin = BnFin();
    if (x_bf) k_bn_partial1_slabs<8, 1><<<nblk, BN_PT, 0, S_(stream)>>>(x, y, dys.slabs, dys.S, elems, dys.acc, const_cast<float*>(dy), n, c,
                                                                      mean, invstd, next_pow2(c / 8), bn_rows(n, c), part);
    else k_bn_partial1_slabs<4, 0><<<nblk, BN_PT, 0, S_(stream)>>>(x, y, dys.slabs, dys.S, elems, dys.acc, const_cast<float*>(dy), n, c,
                                                                  mean, invstd, next_pow2(c / 4), bn_rows(n, c), part);
  } else if (v8)
    BN_PARTIAL1(8, false, next_pow2(c / 8));
  else if (v4 && any_bf)
    BN_PARTIAL1(4, true, next_pow2(c / 4));
  else if (v4)
    BN_PARTIAL1(4, false, next_pow2(c / 4));
  else {
    IRX_REQUIRE(c <= 256, "irx_bn_backward: c=%d needs c %% 4 == 0 or c <= 256", c);
    fin = BnFin();
    BN_PARTIAL1(1, false, next_pow2(c));
  }
#undef BN_PARTIAL1
  if (phases & 1) IRX_CHECK_LAUNCH("irx_bn_backward(partial)");
  if ((phases & 1) && !(fin.counter && !(bn_abl() & 4))) {      // (otherwise: folded by the last workgroup of the launch above)
    k_bn_finalize<1><<<irx_cdiv(c, 32), 32 * BN_FIN_SLICES, 0, S(stream)>>>(part, nblk, n, c, 0.f, 0.f, dbeta, dgamma,
                                                            nullptr, nullptr);
    IRX_CHECK_LAUNCH("irx_bn_backward(finalize)");
  }
  if (!(phases & 2) || (bn_abl() & 8)) return IRX_OK;
  const float* sg = (phases == 2) ? all_sum_g : dbeta;
  const float* sgx = (phases == 2) ? all_sum_gx : dgamma;
  const float inv_count = (phases == 2) ? (all_count >= 1.0 ? (float)(1.0 / all_count) : 0.f) : 1.f / (float)n;
#define BN_BWD_APPLY(V_, TY_, QP_)                                                                                         \
  do {                                                                                                                     \
    const int qpad = QP_;                                                                                                  \
    if (rm) k_bn_bwd_apply<V_, TY_, true, true, false><<<row_grid(n, qpad), 256, 0, S(stream)>>>(       /* no shortcut */ \
        x, y, dy, n, c, qpad, mean, invstd, gamma, sg, sgx, relu, dx, dresidual, ty, inv_count, count_dev, mk_beta);       \
    else if (relu && dresidual) k_bn_bwd_apply<V_, TY_, false, true, true><<<row_grid(n, qpad), 256, 0, S(stream)>>>(      \
        x, y, dy, n, c, qpad, mean, invstd, gamma, sg, sgx, relu, dx, dresidual, ty, inv_count, count_dev, nullptr);       \
    else if (relu) k_bn_bwd_apply<V_, TY_, false, true, false><<<row_grid(n, qpad), 256, 0, S(stream)>>>(                  \
        x, y, dy, n, c, qpad, mean, invstd, gamma, sg, sgx, relu, dx, dresidual, ty, inv_count, count_dev, nullptr);       \
    else if (dresidual) k_bn_bwd_apply<V_, TY_, false, false, true><<<row_grid(n, qpad), 256, 0, S(stream)>>>(             \
        x, y, dy, n, c, qpad, mean, invstd, gamma, sg, sgx, relu, dx, dresidual, ty, inv_count, count_dev, nullptr);       \
    else k_bn_bwd_apply<V_, TY_, false, false, false><<<row_grid(n, qpad), 256, 0, S(stream)>>>(                           \
        x, y, dy, n, c, qpad, mean, invstd, gamma, sg, sgx, relu, dx, dresidual, ty, inv_count, count_dev, nullptr);       \
  } while (0)
  if (v8) BN_BWD_APPLY(8, false, next_pow2(c / 8));
  else if (v4 && any_bf) BN_BWD_APPLY(4, true, next_pow2(c / 4));
  else if (v4) BN_BWD_APPLY(4, false, next_pow2(c / 4));
  else BN_BWD_APPLY(1, false, next_pow2(c));
#undef BN_BWD_APPLY
  IRX_CHECK_LAUNCH("irx_bn_backward(apply)");
  return IRX_OK;
}

// ---- typed entry points: the same operators on tensors whose element type is float32 (0) or bf16 (1) per tensor — what the
// encoder executor's bf16 storage mode calls internally, exported for callers that keep activations in bf16 themselves (and for
// tools/bn_microbench.py). bf16 tensors need c % 4 == 0 (c % 8 == 0 for the fast paths) and 16-byte aligned pointers. beta
// (irx_bn_backward_ex, relu, no dresidual): the layer had no shortcut — the ReLU mask is recomputed from x and y is not read.
extern "C" int irx_bn_forward_ex(const void* x, int n, int c, float eps, float momentum, const float* gamma, const float* beta,
                                 const void* residual, int relu, float* mean, float* invstd, float* running_mean,
                                 float* running_var, void* y, void* workspace, size_t workspace_bytes, void* stream, int x_bf,
                                 int res_bf, int y_bf) {
  return irx_bn_forward_t((const float*)x, n, c, eps, momentum, gamma, beta, (const float*)residual, relu, mean, invstd,
                          running_mean, running_var, (float*)y, workspace, workspace_bytes, stream, x_bf, res_bf, y_bf);
}
extern "C" int irx_bn_backward_ex(const void* x, const void* y, const void* dy, int n, int c, const float* mean,
                                  const float* invstd, const float* gamma, const float* beta, int relu, void* dx, float* dgamma,
                                  float* dbeta, void* dresidual, void* workspace, size_t workspace_bytes, void* stream, int x_bf,
                                  int y_bf, int dy_bf, int dx_bf, int dres_bf) {
  return irx_bn_backward_t((const float*)x, (const float*)y, (const float*)dy, n, c, mean, invstd, gamma, relu, (float*)dx, dgamma,
                           dbeta, (float*)dresidual, workspace, workspace_bytes, stream, x_bf, y_bf, dy_bf, dx_bf, dres_bf, 3,
                           nullptr, nullptr, 0.0, nullptr, beta);
}
