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
  while (rows > 256 && n / rows < 128) rows >>= 1;
  if (rows < 256) rows = 256;
  return rows;
}

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

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
                                                    const float* __restrict__ mk_gamma, const float* __restrict__ mk_beta) {
  // RM (compile time: a run-time flag in front of the loads keeps the compiler from batching a row group's loads, measured
  // again in round 3: 15.8 -> 21.8 us) with mk_gamma / mk_beta (MODE 1, relu, a layer WITHOUT a shortcut): the ReLU mask is recomputed from x —
  // y > 0  <=>  fma(x, invstd * gamma, fma(-mean, invstd * gamma, beta)) > 0, the very expression k_bn_apply evaluated — so y
  // is never read (a third of this pass's bytes)
  __shared__ float s0[BN_PT * V];
  __shared__ float s1[BN_PT * V];
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
  // thread (rg == 0, qd) folds the row groups
  if (rg == 0 && qd < cq) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
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
  if (bn_abl() & 1) {
  } else if (v4 && x_bf && c % 8 == 0)
    k_bn_partial<0, 8, false><<<nblk, BN_PT, 0, S(stream)>>>(x, nullptr, nullptr, n, c, nullptr, nullptr, 0,
                                                   next_pow2(c / 8), bn_rows(n, c), part, ty, nullptr, nullptr);
  else if (v4 && x_bf)
    k_bn_partial<0, 4, true><<<nblk, BN_PT, 0, S(stream)>>>(x, nullptr, nullptr, n, c, nullptr, nullptr, 0,
                                                         next_pow2(c / 4), bn_rows(n, c), part, ty, nullptr, nullptr);
  else if (v4)
    k_bn_partial<0, 4, false><<<nblk, BN_PT, 0, S(stream)>>>(x, nullptr, nullptr, n, c, nullptr, nullptr, 0,
                                                          next_pow2(c / 4), bn_rows(n, c), part, ty, nullptr, nullptr);
  else {
    IRX_REQUIRE(c <= 256, "irx_bn_stats: c=%d needs c %% 4 == 0 or c <= 256", c);
    k_bn_partial<0, 1, false><<<nblk, BN_PT, 0, S(stream)>>>(x, nullptr, nullptr, n, c, nullptr, nullptr, 0,
                                                   next_pow2(c), bn_rows(n, c), part, ty, nullptr, nullptr);
  }
  IRX_CHECK_LAUNCH("irx_bn_stats(partial)");
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
                      double all_count, const double* count_dev, const float* beta) {
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
  const bool any_bf = (x_bf | y_bf | dy_bf | dx_bf | dres_bf) != 0;
  const bool v8 = v4 && x_bf && (y_bf || !relu) && dy_bf && dx_bf && (!dresidual || dres_bf) && c % 8 == 0;
  const bool rm = mk_beta != nullptr;            // mask recomputed from x: separate instantiations (see k_bn_partial)
  // Measured per kernel (rocprofv3, same box, the 9 shortcut-free layers of both encoders): the apply pass gains from not
  // reading y (14.4 -> 11.7 us average), the statistics pass LOSES (15.7 -> 18.3 us: it is bound by the latency of a row
  // group's loads, not by their bytes, and the recomputation lengthens the dependent chain behind them) — so only the apply
  // pass recomputes the mask; the statistics pass keeps reading y.
  static const bool rm_stats = getenv("IRX_BN_REMASK_STATS") && atoi(getenv("IRX_BN_REMASK_STATS")) != 0;   // dev A/B knob
#define BN_PARTIAL1(V_, TY_, QP_)                                                                                          \
  do {                                                                                                                     \
    if (rm && rm_stats) k_bn_partial<1, V_, TY_, true, true><<<nblk, BN_PT, 0, S(stream)>>>(x, y, dy, n, c, mean, invstd, relu, QP_,     \
                                                                        bn_rows(n, c), part, ty, mk_gamma, mk_beta);       \
    else if (relu) k_bn_partial<1, V_, TY_, false, true><<<nblk, BN_PT, 0, S(stream)>>>(x, y, dy, n, c, mean, invstd, relu, QP_, \
                                                                      bn_rows(n, c), part, ty, nullptr, nullptr);          \
    else k_bn_partial<1, V_, TY_, false, false><<<nblk, BN_PT, 0, S(stream)>>>(x, y, dy, n, c, mean, invstd, relu, QP_,      \
                                                                      bn_rows(n, c), part, ty, nullptr, nullptr);          \
  } while (0)
  if (!(phases & 1) || (bn_abl() & 4)) {
  } else if (v8)
    BN_PARTIAL1(8, false, next_pow2(c / 8));
  else if (v4 && any_bf)
    BN_PARTIAL1(4, true, next_pow2(c / 4));
  else if (v4)
    BN_PARTIAL1(4, false, next_pow2(c / 4));
  else {
    IRX_REQUIRE(c <= 256, "irx_bn_backward: c=%d needs c %% 4 == 0 or c <= 256", c);
    BN_PARTIAL1(1, false, next_pow2(c));
  }
#undef BN_PARTIAL1
  if (phases & 1) {
    IRX_CHECK_LAUNCH("irx_bn_backward(partial)");
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
