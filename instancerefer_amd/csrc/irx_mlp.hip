// irx_mlp.hip — the two-layer head MLPs of the matching modules as ONE operator each way:
//     y = W2 . D( relu( N( W1 x + b1 ) ) ) + b2,     N = BatchNorm1d (train / eval) or LayerNorm, D = dropout (optional)
// (reference models/attribute_module.py:26-34 lang_emb_fc / vis_emb_fc, relation_module.py:18-27, scene_module.py:38-42
// vis_emb_fc1 / lang_emb_fc / cls: nn.Sequential(Linear, BatchNorm1d | LayerNorm, ReLU, [Dropout], Linear) on (B, 256) or
// (Nc, 128) rows).  Through ATen each of the seven is ~8 forward and ~25 backward operator dispatches on tensors of a few
// KB — with the bf16 step host-bound that is ~1 ms per step of pure dispatch.  Here: 2 launches forward, 2 (BatchNorm) or 3
// (LayerNorm) backward, issued by one C-ABI call each; deterministic (no atomics); any row count.
//
// The GEMMs are 16..512 rows x 128..256 (8 MFLOP): LDS-staged 64 x 32 tiles multiplied with v_mfma_f32_16x16x4_f32 (round 5; the
// fp32 FMA loop they replaced was most of a kernel's time once its dependent global round trips had been batched); a workgroup
// owns a block of 32 hidden (or output) columns for ALL rows, which makes every per-column reduction over the rows (BatchNorm
// statistics, d gamma, d beta, d bias) local to one workgroup.
#include "irx_common.h"

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

#define ML_TR 64   // rows per tile
#define ML_TC 32   // columns per tile
#define ML_TK 128  // reduction chunk

// C[i][j] (i < 64, j < 32) += sum_k A(i, k) * B(j, k) for k in [0, K): both operands are functors staged through LDS in
// chunks of 128; thread t owns column j = t & 31 and rows (t >> 5) + 8 m.  acc[m] accumulates; all 256 threads take part.
// These kernels are a handful of workgroups running a chain of dependent global round trips, so the chunk is as long as the
// static LDS budget allows (K <= 128: ONE round trip per tile, K = 256: two) and a chunk's 48 loads per thread are all in
// flight before the first one is stored (register staging), the next chunk's loads are issued before the current chunk is
// multiplied, and the multiply reads its operands as float4 (9 LDS reads per 32 FMAs instead of per 8): with 32-wide chunks
// filled by load-then-store loops every chunk paid its own round trip, 8 per tile — 34 us per launch for 8 MFLOP.
#define ML_LD (ML_TK + 4)   // LDS row stride in floats: rows stay 16-byte aligned for the float4 fragment reads
template <typename FA, typename FB>
__device__ __forceinline__ void ml_tile(FA A, FB B, int K, float (&acc)[8], float (*sA)[ML_LD], float (*sB)[ML_LD]) {
  const int t = threadIdx.x, j = t & 31, i0 = t >> 5;
  const int wv = t >> 6, mm = t & 15, g4 = (t & 63) >> 4;
  typedef float ml_f32x4 __attribute__((ext_vector_type(4)));
  ml_f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
  constexpr int NA = ML_TR * ML_TK / 256, NB = ML_TC * ML_TK / 256;
  float ra[NA], rb[NB];
  auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int e = t + 256 * q, i = e / ML_TK, k = e % ML_TK;
      ra[q] = (k0 + k < K) ? A(i, k0 + k) : 0.f;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int e = t + 256 * q, jj = e / ML_TK, k = e % ML_TK;
      rb[q] = (k0 + k < K) ? B(jj, k0 + k) : 0.f;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += ML_TK) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int e = t + 256 * q;
      sA[e / ML_TK][e % ML_TK] = ra[q];
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int e = t + 256 * q;
      sB[e / ML_TK][e % ML_TK] = rb[q];
    }
    __syncthreads();
    if (k0 + ML_TK < K) fetch(k0 + ML_TK);            // the next chunk travels while this one is multiplied
    const int kn = (K - k0 < ML_TK) ? K - k0 : ML_TK;  // (columns kn .. of the chunk hold zeros: whole 4-steps are safe)
    // round 5: the products on the matrix core — v_mfma_f32_16x16x4_f32, wave w owns rows 16 w .. 16 w + 15 x both 16-column
    // halves: 3 conflict-free ds_read_b32 + 2 MFMAs per 4 reduction steps instead of 9 ds_read_b128 + 32 v_fma per thread
    // (the FMA loop was ~2.7 us per 128-wide chunk with one wave per SIMD and nothing to overlap its LDS waits with)
#pragma unroll 8
    for (int k = 0; k < kn; k += 4) {
      const float av = sA[16 * wv + mm][k + g4];
      const float b0 = sB[mm][k + g4], b1 = sB[16 + mm][k + g4];
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, d1, 0, 0, 0);
    }
  }
  // accumulator layout of the MFMA (rows 16 w + 4 g4 + r, column mm | 16 + mm) -> the callers' (row i0 + 8 m, column j)
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    sA[16 * wv + 4 * g4 + r][mm] = d0[r];
    sA[16 * wv + 4 * g4 + r][16 + mm] = d1[r];
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] += sA[i0 + 8 * m][j];
}

// block-wide column sums: v[m] of thread (j, i0) for rows i0 + 8 m -> total over the 64 rows of the tile, valid in threads
// with i0 == 0 (returned to every thread through LDS)
__device__ __forceinline__ float ml_colsum(float part, float (*red)[ML_TC]) {
  const int t = threadIdx.x, j = t & 31, i0 = t >> 5;
  __syncthreads();
  red[i0][j] = part;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) s += red[q][j];
  return s;
}

// sum of p[0], p[stride], ... (n terms) with the loads of 8 terms in flight at a time: these kernels are a handful of workgroups, so a
// loop of dependent "load, add" steps costs a full memory round trip per term (64 terms = 25-50 us)
template <typename F>
__device__ __forceinline__ float ml_sum8(int n, F term) {
  float s = 0.f;
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = term(i + u);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < n; ++i) s += term(i);
  return s;
}

enum { ML_BN_TRAIN = 1, ML_BN_EVAL = 2, ML_LN = 3, ML_NONE = 4, ML_OUT_RELU = 8 };   // (ML_OUT_RELU: flag, y = relu(...))

// Dropout without a mask tensor: element o of call `seed` is kept iff a counter-based hash of (seed, o) falls above p — the
// same decision wherever it is evaluated (forward 1 / forward 2 recompute it), scaled by 1 / (1 - p) like F.dropout. The
// backward never needs it again: a kept, positive activation is a > 0.  (Another random stream than torch's Philox: dropout
// has no parity requirement beyond its distribution; tests run with p = 0.)
__device__ __forceinline__ float ml_drop(float v, float p, float scale, unsigned long long seed, size_t o) {
  if (p <= 0.f) return v;
  const unsigned u = (unsigned)(irx_mix64(seed ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(o + 1))) >> 40);   // 24 bits
  return ((float)u * (1.0f / 16777216.0f) >= p) ? v * scale : 0.f;
}

// ---- forward 1: h = x W1^T + b1 for this block's 32 hidden columns, all rows; BatchNorm: statistics + a = D(relu(N(h))) ----
__global__ __launch_bounds__(256) void k_mlp_fwd1(const float* __restrict__ x, int rows, int din, int dh,
                                                  const float* __restrict__ w1, const float* __restrict__ b1, int norm,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                  float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                                  float drop_p, unsigned long long seed, float* __restrict__ h,
                                                  float* __restrict__ stat, float* __restrict__ a) {
  __shared__ __attribute__((aligned(16))) float sA[ML_TR][ML_LD], sB[ML_TC][ML_LD], red[8][ML_TC];
  const int t = threadIdx.x, j = t & 31, i0 = t >> 5;
  const int c = blockIdx.x * ML_TC + j;
  const bool cv = c < dh;
  double s1 = 0.0, s2 = 0.0;
  float hv[8];
  // the column's parameters travel with the first tile's operands (loaded behind the tile they were one more dependent round
  // trip each: bias, then gamma / beta, then the running statistics)
  const bool has_nrm = norm != ML_NONE && norm != ML_LN;
  const float bias = cv ? b1[c] : 0.f;
  const float gam_c = (cv && has_nrm) ? gamma[c] : 1.f, bet_c = (cv && has_nrm) ? beta[c] : 0.f;
  const float rm_c = (cv && has_nrm && rmean) ? rmean[c] : 0.f, rv_c = (cv && has_nrm && rvar) ? rvar[c] : 1.f;
  for (int r0 = 0; r0 < rows; r0 += ML_TR) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    ml_tile([&](int i, int k) { return (r0 + i < rows) ? x[(size_t)(r0 + i) * din + k] : 0.f; },
            [&](int jj, int k) { const int cc = blockIdx.x * ML_TC + jj; return cc < dh ? w1[(size_t)cc * din + k] : 0.f; },
            din, acc, sA, sB);
    float p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int r = r0 + i0 + 8 * m;
      hv[m] = 0.f;
      if (r < rows && cv) {
        const float v = acc[m] + bias;
        hv[m] = v;
        h[(size_t)r * dh + c] = v;
        p1 += v;
        p2 = fmaf(v, v, p2);
      }
    }
    if (norm == ML_BN_TRAIN) {
      s1 += (double)ml_colsum(p1, red);
      s2 += (double)ml_colsum(p2, red);
    }
  }
  if (norm == ML_LN) return;                          // row statistics need the whole row: forward 2
  // (one row tile — every head of the model: 16 or 64 rows — keeps this thread's h values in registers, so the second pass
  //  of the variance and the activation below do not wait for the stores above to come back from memory)
  const bool one_tile = rows <= ML_TR;
  float mean, invstd;
  if (norm == ML_BN_TRAIN) {
    // two-pass variance over this block's own h (E[x^2] - mean^2 loses everything when two rows are nearly equal)
    const double m = s1 / rows;
    float pv = 0.f;
    if (cv) {
      if (one_tile) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (i0 + 8 * q < rows) { const float d = hv[q] - (float)m; pv = fmaf(d, d, pv); }
      } else {
        for (int r = i0; r < rows; r += 8) {           // (the entries of h this thread stored above)
          const float d = h[(size_t)r * dh + c] - (float)m;
          pv = fmaf(d, d, pv);
        }
      }
    }
    const double var = (double)ml_colsum(pv, red) / rows;
    (void)s2;
    mean = (float)m;
    invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (cv && i0 == 0 && rmean) {                     // nn.BatchNorm1d: running statistics with the unbiased variance
      const double unb = rows > 1 ? var * rows / (rows - 1) : var;
      rmean[c] = (1.f - momentum) * rm_c + momentum * mean;
      rvar[c] = (1.f - momentum) * rv_c + momentum * (float)unb;
    }
  } else if (norm == ML_NONE) {                         // Linear -> ReLU -> Dropout -> Linear: the identity "normalisation"
    mean = 0.f;
    invstd = 1.f;
  } else {
    mean = cv ? rm_c : 0.f;
    invstd = cv ? rsqrtf(rv_c + eps) : 0.f;
  }
  if (cv && i0 == 0) {
    stat[c] = mean;
    stat[dh + c] = invstd;
  }
  if (!cv) return;
  const float g = norm == ML_NONE ? 1.f : gam_c * invstd, bsh = norm == ML_NONE ? 0.f : bet_c - mean * g;
  const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  if (one_tile) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = i0 + 8 * q;
      if (r < rows) {
        const size_t o = (size_t)r * dh + c;
        a[o] = ml_drop(fmaxf(fmaf(hv[q], g, bsh), 0.f), drop_p, dscale, seed, o);
      }
    }
    return;
  }
  for (int r = i0; r < rows; r += 8) {                // (rows i0 + 8 m: exactly the entries of h this thread stored above)
    const size_t o = (size_t)r * dh + c;
    a[o] = ml_drop(fmaxf(fmaf(h[o], g, bsh), 0.f), drop_p, dscale, seed, o);
  }
}

// ---- forward 2: y = a W2^T + b2 for this block's 32 output columns; LayerNorm: row statistics and a are made here ----
__global__ __launch_bounds__(256) void k_mlp_fwd2(const float* __restrict__ h, int rows, int dh, int dout, int norm,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                  float drop_p, unsigned long long seed, const float* __restrict__ w2,
                                                  const float* __restrict__ b2, float* __restrict__ stat,
                                                  float* __restrict__ a, float* __restrict__ y, int out_relu) {
  __shared__ __attribute__((aligned(16))) float sA[ML_TR][ML_LD], sB[ML_TC][ML_LD];
  __shared__ float sMean[ML_TR], sInv[ML_TR];
  const int t = threadIdx.x, j = t & 31, i0 = t >> 5;
  const int c = blockIdx.x * ML_TC + j;
  const float dscale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const float bias2 = c < dout ? b2[c] : 0.f;           // (travels with the first loads instead of behind the tile)
  for (int r0 = 0; r0 < rows; r0 += ML_TR) {
    if (norm == ML_LN) {                              // every block recomputes the tile's row statistics (dh <= 512 floats a row)
      __syncthreads();
      {
        // four threads per row, all 64 rows of the tile at once (one wave per row, 16 rows in sequence, was 16 dependent
        // round trips); thread q of a row sums elements q, q + 4, ... — the four partial sums are combined in a fixed order
        // two passes (mean, then the centred squares — the row is in L1 by then): E[x^2] - mean^2 in fp32 loses ~1e-6 x
        // mean^2 / var of the variance, enough to show in the gradient's direction (training-trajectory test: cosine 2e-6)
        const int i = t >> 2, q = t & 3, r = r0 + i;
        float p1 = 0.f, p2 = 0.f;
        const float* hr = h + (size_t)(r < rows ? r : 0) * dh;
        float m;
        if (dh <= 256 && (dh & 15) == 0) {
          // the thread's quarter of the row in ONE round trip (<= 16 float4 loads, all in flight) and kept for the second pass:
          // this prologue was 16 dependent round trips (two passes of 8 batches) = 10-13 us of a 29 us kernel (round 5)
          float4 v[16];
          const int nv = (r < rows) ? dh / 16 : 0;
#pragma unroll
          for (int u = 0; u < 16; ++u)
            v[u] = (u < nv) ? *reinterpret_cast<const float4*>(hr + 4 * (q + 4 * u)) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int u = 0; u < 16; ++u) p1 += (v[u].x + v[u].y) + (v[u].z + v[u].w);
          p1 += __shfl_xor(p1, 1);
          p1 += __shfl_xor(p1, 2);
          m = p1 / dh;
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (u < nv) {
              const float d0 = v[u].x - m, d1 = v[u].y - m, d2 = v[u].z - m, d3 = v[u].w - m;
              p2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        } else {
          const int nq = (r < rows) ? (dh - q + 3) / 4 : 0;            // this thread's terms: k = q, q + 4, ...
          p1 = ml_sum8(nq, [&](int u) { return hr[q + 4 * u]; });
          p1 += __shfl_xor(p1, 1);
          p1 += __shfl_xor(p1, 2);
          m = p1 / dh;
          p2 = ml_sum8(nq, [&](int u) { const float d = hr[q + 4 * u] - m; return d * d; });
        }
        p2 += __shfl_xor(p2, 1);
        p2 += __shfl_xor(p2, 2);
        if (q == 0) {
          const float var = p2 / dh;
          sMean[i] = m;
          sInv[i] = rsqrtf(var + eps);
          if (blockIdx.x == 0 && r < rows) { stat[2 * r] = m; stat[2 * r + 1] = sInv[i]; }
        }
      }
      __syncthreads();
      // a (needed by the backward) is written once: every block's tile fetch below evaluates all of the tile's activations
      // anyway, so block b stores the rows r = b (mod blocks) as it goes — the separate loop in block 0 was 8 more dependent
      // round trips on the slowest workgroup
    }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    ml_tile([&](int i, int k) {
              const int r = r0 + i;
              if (r >= rows) return 0.f;
              const size_t o = (size_t)r * dh + k;
              if (norm != ML_LN) return a[o];
              const float v = ml_drop(fmaxf(fmaf((h[o] - sMean[i]) * sInv[i], gamma[k], beta[k]), 0.f), drop_p, dscale, seed, o);
              if (r % (int)gridDim.x == (int)blockIdx.x) a[o] = v;
              return v;
            },
            [&](int jj, int k) { const int cc = blockIdx.x * ML_TC + jj; return cc < dout ? w2[(size_t)cc * dh + k] : 0.f; },
            dh, acc, sA, sB);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int r = r0 + i0 + 8 * m;
      if (r < rows && c < dout) {
        const float v = acc[m] + bias2;
        y[(size_t)r * dout + c] = out_relu ? fmaxf(v, 0.f) : v;
      }
    }
  }
}

// ---- backward 1 (block = 32 hidden columns, all rows): da = dy W2, dz = relu'/dropout, d gamma, d beta, dW2 rows, db2;
//      BatchNorm: dh complete; LayerNorm: g = dz * gamma is left in dh for backward 2 ----
__global__ __launch_bounds__(256) void k_mlp_bwd1(const float* __restrict__ dy, int rows, int dh, int dout, int norm,
                                                  const float* __restrict__ gamma, const float* __restrict__ w2,
                                                  const float* __restrict__ h, const float* __restrict__ stat,
                                                  const float* __restrict__ a, float drop_scale, float* __restrict__ dhid,
                                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                  float* __restrict__ dw2, float* __restrict__ db2) {
  __shared__ __attribute__((aligned(16))) float sA[ML_TR][ML_LD], sB[ML_TC][ML_LD], red[8][ML_TC];
  const int t = threadIdx.x, j = t & 31, i0 = t >> 5;
  const int c = blockIdx.x * ML_TC + j;
  const bool cv = c < dh;
  if (blockIdx.y > 0) {
    // dW2[o][c] = sum_r dy[r][o] a[r][c] for this block's columns c and the 64 outputs of tile blockIdx.y - 1 (reduction over
    // the rows); it only reads forward-saved tensors, so it runs beside the data-gradient blocks instead of behind them
    const int o0 = ((int)blockIdx.y - 1) * ML_TR;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    ml_tile([&](int i, int k) { return (o0 + i < dout) ? dy[(size_t)k * dout + o0 + i] : 0.f; },
            [&](int jj, int k) { const int cc = blockIdx.x * ML_TC + jj; return cc < dh ? a[(size_t)k * dh + cc] : 0.f; },
            rows, acc, sA, sB);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int o = o0 + i0 + 8 * m;
      if (o < dout && cv) dw2[(size_t)o * dh + c] = acc[m];
    }
    if (blockIdx.x == 0)                                // db2 of this tile's outputs: 4 threads per output, fixed order
      for (int e = t; e < 4 * ML_TR; e += 256) {
        const int o = o0 + (e >> 2), q = e & 3;
        float sdb = (o < dout) ? ml_sum8((rows - q + 3) / 4, [&](int u) { return dy[(size_t)(q + 4 * u) * dout + o]; }) : 0.f;
        sdb += __shfl_xor(sdb, 1);
        sdb += __shfl_xor(sdb, 2);
        if (q == 0 && o < dout) db2[o] = sdb;
      }
    return;
  }
  const float mean_c = (norm != ML_LN && cv) ? stat[c] : 0.f, inv_c = (norm != ML_LN && cv) ? stat[dh + c] : 0.f;
  float sg = 0.f, sb = 0.f;
  // one row tile (every head of the model: 16 or 64 rows) with a column normalisation: dz and xhat stay in registers and dhid is
  // written ONCE below — the write-then-read-modify-write over dhid was 8 dependent round trips per thread
  const bool one_tile = rows <= ML_TR && norm != ML_LN && norm != ML_NONE;
  float dzv[8], xhv[8];
  for (int r0 = 0; r0 < rows; r0 += ML_TR) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    ml_tile([&](int i, int k) { return (r0 + i < rows) ? dy[(size_t)(r0 + i) * dout + k] : 0.f; },
            [&](int jj, int k) { const int cc = blockIdx.x * ML_TC + jj; return cc < dh ? w2[(size_t)k * dh + cc] : 0.f; },
            dout, acc, sA, sB);
    float pg = 0.f, pb = 0.f;
    float av[8], hv[8], m0[8], m1[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {                      // all loads of the tile's epilogue first
      const int r = r0 + i0 + 8 * m;
      const bool ok = r < rows && cv;
      const size_t o = ok ? (size_t)r * dh + c : 0;
      av[m] = ok ? a[o] : 0.f;
      hv[m] = ok ? h[o] : 0.f;
      m0[m] = (ok && norm == ML_LN) ? stat[2 * r] : mean_c;
      m1[m] = (ok && norm == ML_LN) ? stat[2 * r + 1] : inv_c;
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int r = r0 + i0 + 8 * m;
      dzv[m] = xhv[m] = 0.f;
      if (r < rows && cv) {
        const size_t o = (size_t)r * dh + c;
        const float dz = av[m] > 0.f ? acc[m] * drop_scale : 0.f;
        const float xh = (hv[m] - m0[m]) * m1[m];
        pg = fmaf(dz, xh, pg);
        pb += dz;
        dzv[m] = dz;
        xhv[m] = xh;
        if (!one_tile) dhid[o] = (norm == ML_LN) ? dz * gamma[c] : dz;      // (BatchNorm: finished below)
      }
    }
    sg += ml_colsum(pg, red);
    sb += ml_colsum(pb, red);
  }
  if (cv && i0 == 0 && norm != ML_NONE) {
    dgamma[c] = sg;
    dbeta[c] = sb;
  }
  if (norm != ML_LN && norm != ML_NONE && cv) {       // (rows i0 + 8 m: this thread's own dhid entries; no norm: dhid = dz)
    const float gi = gamma[c] * inv_c;
    const float kb = (norm == ML_BN_TRAIN) ? sb / rows : 0.f, kg = (norm == ML_BN_TRAIN) ? sg / rows : 0.f;
    if (one_tile) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int r = i0 + 8 * m;
        if (r < rows) dhid[(size_t)r * dh + c] = gi * (dzv[m] - kb - xhv[m] * kg);
      }
      return;
    }
    for (int r = i0; r < rows; r += 8) {
      const size_t o = (size_t)r * dh + c;
      const float xh = (h[o] - mean_c) * inv_c;
      dhid[o] = gi * (dhid[o] - kb - xh * kg);
    }
  }
}

// ---- backward 2 (LayerNorm): one wave per row: dh = invstd (g - mean(g) - xhat mean(g xhat)) ----
__global__ __launch_bounds__(256) void k_mlp_bwd2_ln(int rows, int dh, const float* __restrict__ h, const float* __restrict__ stat,
                                                     float* __restrict__ dhid) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float m = stat[2 * r], inv = stat[2 * r + 1];
  float p1 = 0.f, p2 = 0.f;
  for (int k = lane; k < dh; k += 64) {
    const size_t o = (size_t)r * dh + k;
    const float g = dhid[o], xh = (h[o] - m) * inv;
    p1 += g;
    p2 = fmaf(g, xh, p2);
  }
  for (int off = 32; off > 0; off >>= 1) { p1 += __shfl_xor(p1, off); p2 += __shfl_xor(p2, off); }
  p1 /= dh;
  p2 /= dh;
  for (int k = lane; k < dh; k += 64) {
    const size_t o = (size_t)r * dh + k;
    const float xh = (h[o] - m) * inv;
    dhid[o] = inv * (dhid[o] - p1 - xh * p2);
  }
}

// ---- backward 3: blocks [0, nW): dW1 tile (64 hidden rows x 32 input columns, reduction over rows) + db1;
//      blocks [nW, nW + nX): dx tile (64 rows x 32 input columns, reduction over hidden) ----
__global__ __launch_bounds__(256) void k_mlp_bwd3(const float* __restrict__ x, int rows, int din, int dh,
                                                  const float* __restrict__ w1, const float* __restrict__ dhid,
                                                  float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dx,
                                                  int nW) {
  __shared__ __attribute__((aligned(16))) float sA[ML_TR][ML_LD], sB[ML_TC][ML_LD];
  const int t = threadIdx.x, j = t & 31, i0 = t >> 5;
  const int kblocks = (din + ML_TC - 1) / ML_TC;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if ((int)blockIdx.x < nW) {
    const int c0 = ((int)blockIdx.x / kblocks) * ML_TR, k0 = ((int)blockIdx.x % kblocks) * ML_TC;
    ml_tile([&](int i, int r) { return (c0 + i < dh) ? dhid[(size_t)r * dh + c0 + i] : 0.f; },
            [&](int jj, int r) { return (k0 + jj < din) ? x[(size_t)r * din + k0 + jj] : 0.f; },
            rows, acc, sA, sB);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int c = c0 + i0 + 8 * m;
      if (c < dh && k0 + j < din) dw1[(size_t)c * din + k0 + j] = acc[m];
    }
    if (k0 == 0)
      for (int c = c0 + t; c < c0 + ML_TR && c < dh; c += 256) {
        db1[c] = ml_sum8(rows, [&](int r) { return dhid[(size_t)r * dh + c]; });
      }
  } else if (dx) {
    const int b = (int)blockIdx.x - nW;
    const int r0 = (b / kblocks) * ML_TR, k0 = (b % kblocks) * ML_TC;
    ml_tile([&](int i, int c) { return (r0 + i < rows) ? dhid[(size_t)(r0 + i) * dh + c] : 0.f; },
            [&](int jj, int c) { return (k0 + jj < din) ? w1[(size_t)c * din + k0 + jj] : 0.f; },
            dh, acc, sA, sB);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int r = r0 + i0 + 8 * m;
      if (r < rows && k0 + j < din) dx[(size_t)r * din + k0 + j] = acc[m];
    }
  }
}

// ---- stand-alone dropout over a flat tensor, the ml_drop decision (no mask tensor) ------------------------------------------
// y[o] = keep(seed, o) ? x[o] / (1 - p) : 0. The backward is the same call on the incoming gradient with the same seed.
__global__ __launch_bounds__(256) void k_dropout_flat(const float* __restrict__ x, size_t n, float p, float scale,
                                                      unsigned long long seed, float* __restrict__ y) {
  const size_t q = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (q + 4 <= n && ((((size_t)x | (size_t)y) & 15) == 0)) {
    const float4 v = *reinterpret_cast<const float4*>(x + q);
    float4 r;
    r.x = ml_drop(v.x, p, scale, seed, q);
    r.y = ml_drop(v.y, p, scale, seed, q + 1);
    r.z = ml_drop(v.z, p, scale, seed, q + 2);
    r.w = ml_drop(v.w, p, scale, seed, q + 3);
    *reinterpret_cast<float4*>(y + q) = r;
  } else {
    for (size_t o = q; o < n && o < q + 4; ++o) y[o] = ml_drop(x[o], p, scale, seed, o);
  }
}

extern "C" int irx_dropout_flat(const float* x, size_t n, float p, unsigned long long seed, float* y, void* stream) {
  IRX_REQUIRE(p >= 0.f && p < 1.f, "irx_dropout_flat: dropout probability %f outside [0, 1)", (double)p);
  if (n == 0) return IRX_OK;
  IRX_REQUIRE(x && y, "irx_dropout_flat: null pointer");
  k_dropout_flat<<<irx_cdiv((long long)((n + 3) / 4), 256), 256, 0, S(stream)>>>(x, n, p, p > 0.f ? 1.f / (1.f - p) : 1.f, seed, y);
  IRX_CHECK_LAUNCH("irx_dropout_flat");
  return IRX_OK;
}

// --------------------------------------------------------------------------------------------------------- host ---
extern "C" size_t irx_mlp2_saved_floats(int rows, int dh) {
  // h [rows][dh], a [rows][dh], stat (2 dh for BatchNorm, 2 rows for LayerNorm: the larger is reserved)
  return (size_t)2 * rows * dh + (size_t)2 * (rows > dh ? rows : dh);
}

extern "C" int irx_mlp2_fwd(const float* x, int rows, int din, int dh, int dout, const float* w1, const float* b1, int norm,
                            const float* gamma, const float* beta, float eps, float* running_mean, float* running_var,
                            float momentum, float drop_p, unsigned long long seed, const float* w2, const float* b2,
                            float* saved, float* y, void* stream) {
  IRX_REQUIRE(rows >= 0 && din >= 1 && dh >= 1 && dout >= 1, "irx_mlp2_fwd: bad sizes");
  const int out_relu = (norm & ML_OUT_RELU) ? 1 : 0;
  norm &= ~ML_OUT_RELU;
  IRX_REQUIRE(norm >= ML_BN_TRAIN && norm <= ML_NONE,
              "irx_mlp2_fwd: norm %d is not 1 (BatchNorm train), 2 (eval), 3 (LayerNorm) or 4 (none) [+ 8: ReLU on the output]", norm);
  if (rows == 0) return IRX_OK;
  IRX_REQUIRE(x && w1 && b1 && w2 && b2 && saved && y && (norm == ML_NONE || (gamma && beta)), "irx_mlp2_fwd: null pointer");
  IRX_REQUIRE(norm != ML_BN_EVAL || (running_mean && running_var), "irx_mlp2_fwd: BatchNorm eval needs the running statistics");
  IRX_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "irx_mlp2_fwd: dropout probability %f outside [0, 1)", (double)drop_p);
  IRX_REQUIRE(norm != ML_BN_TRAIN || rows > 1, "irx_mlp2_fwd: train-mode BatchNorm1d needs more than one row");
  float* h = saved;
  float* a = saved + (size_t)rows * dh;
  float* stat = a + (size_t)rows * dh;
  k_mlp_fwd1<<<irx_cdiv(dh, ML_TC), 256, 0, S(stream)>>>(x, rows, din, dh, w1, b1, norm, gamma, beta, eps, running_mean,
                                                          running_var, momentum, drop_p, seed, h, stat, a);
  IRX_CHECK_LAUNCH("irx_mlp2_fwd(1)");
  k_mlp_fwd2<<<irx_cdiv(dout, ML_TC), 256, 0, S(stream)>>>(h, rows, dh, dout, norm, gamma, beta, eps, drop_p, seed, w2, b2, stat, a, y, out_relu);
  IRX_CHECK_LAUNCH("irx_mlp2_fwd(2)");
  return IRX_OK;
}

// dhid: scratch [rows][dh]. dx may be NULL (no input gradient wanted).
extern "C" int irx_mlp2_bwd(const float* x, const float* dy, int rows, int din, int dh, int dout, const float* w1, int norm,
                            const float* gamma, const float* w2, const float* saved, float drop_scale, float* dhid, float* dx,
                            float* dw1, float* db1, float* dgamma, float* dbeta, float* dw2, float* db2, void* stream) {
  IRX_REQUIRE(rows >= 1 && din >= 1 && dh >= 1 && dout >= 1, "irx_mlp2_bwd: bad sizes");
  norm &= ~ML_OUT_RELU;                 // (an output ReLU is the caller's: dy arrives masked by y > 0)
  IRX_REQUIRE(norm >= ML_BN_TRAIN && norm <= ML_NONE, "irx_mlp2_bwd: bad norm %d", norm);
  IRX_REQUIRE(x && dy && w1 && w2 && saved && dhid && dw1 && db1 && dw2 && db2 && (norm == ML_NONE || (gamma && dgamma && dbeta)),
              "irx_mlp2_bwd: null pointer");
  const float* h = saved;
  const float* a = saved + (size_t)rows * dh;
  const float* stat = a + (size_t)rows * dh;
  k_mlp_bwd1<<<dim3(irx_cdiv(dh, ML_TC), 1 + irx_cdiv(dout, ML_TR)), 256, 0, S(stream)>>>(dy, rows, dh, dout, norm, gamma, w2, h, stat, a, drop_scale, dhid, dgamma,
                                                          dbeta, dw2, db2);
  IRX_CHECK_LAUNCH("irx_mlp2_bwd(1)");
  if (norm == ML_LN) {
    k_mlp_bwd2_ln<<<irx_cdiv(rows, 4), 256, 0, S(stream)>>>(rows, dh, h, stat, dhid);
    IRX_CHECK_LAUNCH("irx_mlp2_bwd(2)");
  }
  const int kblocks = irx_cdiv(din, ML_TC);
  const int nW = irx_cdiv(dh, ML_TR) * kblocks, nX = dx ? irx_cdiv(rows, ML_TR) * kblocks : 0;
  k_mlp_bwd3<<<nW + nX, 256, 0, S(stream)>>>(x, rows, din, dh, w1, dhid, dw1, db1, dx, nW);
  IRX_CHECK_LAUNCH("irx_mlp2_bwd(3)");
  return IRX_OK;
}


// ---------------------------------------------------------------------------------------- GRU weight gradients ---
// The four weight gradients of one GRU layer (reference models/lang_module.py:24-32: nn.GRU) from the BPTT outputs of
// irx_gru_backward, in ONE launch: per direction d
//   dW_ih[d][g][i] = sum_r dgi[r][d][g] x[r][i]          dW_hh[d][g][h] = sum_r dgh[r][d][g] hprev_d[r][h]
//   db_ih[d][g]    = sum_r dgi[r][d][g]                   db_hh[d][g]    = sum_r dgh[r][d][g]
// r = (b, t) over all B*T rows (padded steps carry zero gradients); hprev is read from the layer's OUTPUT with the
// direction's own shift (forward: out[b][t-1], reverse: out[b][t+1], zero outside [0, T)) — no shifted copy is materialised.
// Through ATen this was 8 GEMM / reduction launches + 3 to build hprev per layer; same 64 x 32 fp32 FMA tiles as the head MLPs.
// blocks: [0, nI) input-weight tiles, [nI, nI + nH) hidden-weight tiles; tile = 64 gate rows x 32 columns of one direction.
__global__ __launch_bounds__(256) void k_gru_wgrad(const float* __restrict__ dgi, const float* __restrict__ dgh,
                                                   const float* __restrict__ x, const float* __restrict__ out, int B, int T,
                                                   int I, int ndir, int H, float* __restrict__ dwi0, float* __restrict__ dwi1,
                                                   float* __restrict__ dwh0, float* __restrict__ dwh1,
                                                   float* __restrict__ dbi0, float* __restrict__ dbi1,
                                                   float* __restrict__ dbh0, float* __restrict__ dbh1, int nI) {
  __shared__ __attribute__((aligned(16))) float sA[ML_TR][ML_LD], sB[ML_TC][ML_LD];
  const int t = threadIdx.x, j = t & 31, i0 = t >> 5;
  const int G = 3 * H, BT = B * T, ldg = ndir * G;
  const bool hid = (int)blockIdx.x >= nI;
  const int b = hid ? (int)blockIdx.x - nI : (int)blockIdx.x;
  const int ncol = hid ? H : I;                             // columns of this weight
  const int cblocks = (ncol + ML_TC - 1) / ML_TC, gblocks = (G + ML_TR - 1) / ML_TR;
  const int d = b / (gblocks * cblocks), rem = b % (gblocks * cblocks);
  const int g0 = (rem / cblocks) * ML_TR, c0 = (rem % cblocks) * ML_TC;
  const float* gsrc = hid ? dgh : dgi;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (!hid) {
    ml_tile([&](int i, int r) { return (g0 + i < G) ? gsrc[(size_t)r * ldg + d * G + g0 + i] : 0.f; },
            [&](int jj, int r) { return (c0 + jj < I) ? x[(size_t)r * I + c0 + jj] : 0.f; }, BT, acc, sA, sB);
  } else {
    const int sh = d == 0 ? -1 : 1;                          // forward: h_{t-1} = out[t-1]; reverse: out[t+1]
    ml_tile([&](int i, int r) { return (g0 + i < G) ? gsrc[(size_t)r * ldg + d * G + g0 + i] : 0.f; },
            [&](int jj, int r) {
              const int tt = r % T + sh;
              return (c0 + jj < H && tt >= 0 && tt < T) ? out[((size_t)(r + sh) * ndir + d) * H + c0 + jj] : 0.f;
            }, BT, acc, sA, sB);
  }
  float* dw = hid ? (d == 0 ? dwh0 : dwh1) : (d == 0 ? dwi0 : dwi1);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    const int g = g0 + i0 + 8 * m;
    if (g < G && c0 + j < ncol) dw[(size_t)g * ncol + c0 + j] = acc[m];
  }
  if (c0 == 0) {                                            // the bias gradient of this tile's 64 gate rows: 4 threads per row
    float* db = hid ? (d == 0 ? dbh0 : dbh1) : (d == 0 ? dbi0 : dbi1);
    const int g = g0 + (t >> 2), q = t & 3;
    float sdb = (g < G) ? ml_sum8((BT - q + 3) / 4, [&](int u) { return gsrc[(size_t)(q + 4 * u) * ldg + d * G + g]; }) : 0.f;
    sdb += __shfl_xor(sdb, 1);
    sdb += __shfl_xor(sdb, 2);
    if (q == 0 && g < G) db[g] = sdb;
  }
}

extern "C" int irx_gru_wgrad(const float* dgi, const float* dgh, const float* x, const float* out, int B, int T, int I, int ndir,
                             int H, float* dw_ih0, float* dw_ih1, float* dw_hh0, float* dw_hh1, float* db_ih0, float* db_ih1,
                             float* db_hh0, float* db_hh1, void* stream) {
  IRX_REQUIRE(B >= 0 && T >= 0 && I >= 1 && H >= 1 && (ndir == 1 || ndir == 2), "irx_gru_wgrad: bad sizes");
  IRX_REQUIRE(dgi && dgh && x && out && dw_ih0 && dw_hh0 && db_ih0 && db_hh0, "irx_gru_wgrad: null pointer");
  IRX_REQUIRE(ndir == 1 || (dw_ih1 && dw_hh1 && db_ih1 && db_hh1), "irx_gru_wgrad: the second direction's outputs are missing");
  const int G = 3 * H, gblocks = irx_cdiv(G, ML_TR);
  const int nI = ndir * gblocks * irx_cdiv(I, ML_TC), nH = ndir * gblocks * irx_cdiv(H, ML_TC);
  k_gru_wgrad<<<nI + nH, 256, 0, S(stream)>>>(dgi, dgh, x, out, B, T, I, ndir, H, dw_ih0, dw_ih1, dw_hh0, dw_hh1, db_ih0, db_ih1,
                                              db_hh0, db_hh1, nI);
  IRX_CHECK_LAUNCH("irx_gru_wgrad");
  return IRX_OK;
}
