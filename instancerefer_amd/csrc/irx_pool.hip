// irx_pool.hip — segmented max / mean reductions and the batched instance-graph kNN.
// Small, latency-bound kernels (<= a few thousand rows per step); one launch covers every
// instance / query of the batch so the tail levels do not pay per-instance launch overhead.
#include "irx_common.h"

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

// spnn.GlobalMaxPooling / torch_scatter max: thread per channel, loop over the segment's rows.
__global__ void k_segment_max(const float* __restrict__ x, const int32_t* __restrict__ offsets, int c,
                              float* __restrict__ y, int32_t* __restrict__ argmax) {
  const int seg = blockIdx.x;
  const int ch = blockIdx.y * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const int beg = offsets[seg], end = offsets[seg + 1];
  float best = 0.f;
  int arg = -1;
  int r = beg;
  for (; r + 8 <= end; r += 8) {     // eight rows' loads in flight, compared in row order (a load per compare was ~35 dependent round trips)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[(size_t)(r + u) * c + ch];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (arg < 0 || v[u] > best) {
        best = v[u];
        arg = r + u;
      }
  }
  for (; r < end; ++r) {
    const float v = x[(size_t)r * c + ch];
    if (arg < 0 || v > best) {  // first maximum wins (torch.max / scatter_max tie rule)
      best = v;
      arg = r;
    }
  }
  y[(size_t)seg * c + ch] = best;
  argmax[(size_t)seg * c + ch] = arg;
}

__global__ void k_segment_max_bwd(const float* __restrict__ dy, const int32_t* __restrict__ argmax,
                                  size_t total, int c, float* __restrict__ dx) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int arg = argmax[i];
  if (arg >= 0) dx[(size_t)arg * c + (i % c)] = dy[i];
}

// mean over equal-length segments; fp64 accumulate.
__global__ __launch_bounds__(256) void k_segment_mean(const float* __restrict__ x, int len, int c,
                                                      int cpad, float* __restrict__ y) {
  __shared__ double sacc[256];
  const int seg = blockIdx.x;
  const int ch = threadIdx.x % cpad;
  const int rg = threadIdx.x / cpad;
  const int nrg = 256 / cpad;
  const float* base = x + (size_t)seg * len * c;
  for (int cb = 0; cb < c; cb += cpad) {
    // 8 independent partial sums per thread: with wide rows (c > 128: one row group) the loop is 1024 dependent
    // load + fp64-add steps otherwise (450 us for the 135-channel instances of the multiview input)
    double a = 0.0;
    if (cb + ch < c) {
      double p[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      const float* col = base + cb + ch;
      int r = rg;
      for (; r + 7 * nrg < len; r += 8 * nrg) {
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] += (double)col[(size_t)(r + u * nrg) * c];
      }
      for (; r < len; r += nrg) p[0] += (double)col[(size_t)r * c];
      a = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    }
    sacc[threadIdx.x] = a;
    __syncthreads();
    if (rg == 0 && cb + ch < c) {
      double t = 0.0;
      for (int g2 = 0; g2 < nrg; ++g2) t += sacc[g2 * cpad + ch];
      y[(size_t)seg * c + cb + ch] = (float)(t / (double)len);
    }
    __syncthreads();
  }
}

// One wave per query. Selection = k rounds of "smallest (dist, index) strictly greater than the
// previous pick", each round a wave-wide lexicographic min by shuffles.
__device__ static inline float sqdist3(const float* a, float qx, float qy, float qz) {
  const float dx = __fsub_rn(a[0], qx), dy = __fsub_rn(a[1], qy), dz = __fsub_rn(a[2], qz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

__global__ __launch_bounds__(256) void k_knn(const float* __restrict__ sup, const int32_t* __restrict__ sup_off,
                                             const float* __restrict__ qry, const int32_t* __restrict__ qry_batch,
                                             int nq, int k, int32_t* __restrict__ out) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= nq) return;
  const int b = qry_batch[q];
  const int beg = sup_off[b], end = sup_off[b + 1];
  const float qx = qry[3 * (size_t)q + 0], qy = qry[3 * (size_t)q + 1], qz = qry[3 * (size_t)q + 2];
  float pd = -1.f;  // previous pick (distances are >= 0)
  int pi = -1;
  for (int round = 0; round < k; ++round) {
    float bd = 3.4e38f;
    int bi = 0x7FFFFFFF;
    for (int j = beg + lane; j < end; j += 64) {
      const float d = sqdist3(sup + 3 * (size_t)j, qx, qy, qz);
      const bool after_prev = (d > pd) || (d == pd && j > pi);
      const bool better = (d < bd) || (d == bd && j < bi);
      if (after_prev && better) {
        bd = d;
        bi = j;
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      const float od = __shfl_xor(bd, off);
      const int oi = __shfl_xor(bi, off);
      if (od < bd || (od == bd && oi < bi)) {
        bd = od;
        bi = oi;
      }
    }
    const bool found = bi != 0x7FFFFFFF;
    if (lane == 0) out[(size_t)q * k + round] = found ? bi : -1;
    if (!found) {
      if (lane == 0)
        for (int r2 = round + 1; r2 < k; ++r2) out[(size_t)q * k + r2] = -1;
      break;
    }
    pd = bd;
    pi = bi;
  }
}

// ------------------------------------------------------------------------------ C entry ------
extern "C" int irx_segment_max(const float* x, const int32_t* offsets, int nseg, int c, float* y,
                               int32_t* argmax, void* stream) {
  IRX_REQUIRE(nseg >= 0 && c >= 1, "irx_segment_max: bad sizes");
  if (nseg == 0) return IRX_OK;
  IRX_REQUIRE(offsets && y && argmax, "irx_segment_max: null pointer");
  const int bs = c >= 128 ? 128 : 64;
  dim3 grid(nseg, irx_cdiv(c, bs));
  k_segment_max<<<grid, bs, 0, S(stream)>>>(x, offsets, c, y, argmax);
  IRX_CHECK_LAUNCH("irx_segment_max");
  return IRX_OK;
}

extern "C" int irx_segment_max_backward(const float* dy, const int32_t* argmax, int nseg, int c,
                                        float* dx, void* stream) {
  IRX_REQUIRE(nseg >= 0 && c >= 1, "irx_segment_max_backward: bad sizes");
  if (nseg == 0) return IRX_OK;
  IRX_REQUIRE(dy && argmax && dx, "irx_segment_max_backward: null pointer");
  const size_t total = (size_t)nseg * c;
  k_segment_max_bwd<<<irx_cdiv((long long)total, 256), 256, 0, S(stream)>>>(dy, argmax, total, c, dx);
  IRX_CHECK_LAUNCH("irx_segment_max_backward");
  return IRX_OK;
}

extern "C" int irx_segment_mean(const float* x, int nseg, int len, int c, float* y, void* stream) {
  IRX_REQUIRE(nseg >= 0 && len >= 1 && c >= 1, "irx_segment_mean: bad sizes");
  if (nseg == 0) return IRX_OK;
  IRX_REQUIRE(x && y, "irx_segment_mean: null pointer");
  int cpad = 1;
  while (cpad < c && cpad < 256) cpad <<= 1;
  k_segment_mean<<<nseg, 256, 0, S(stream)>>>(x, len, c, cpad, y);
  IRX_CHECK_LAUNCH("irx_segment_mean");
  return IRX_OK;
}

extern "C" int irx_knn_batched(const float* sup_xyz, const int32_t* sup_offsets, const float* qry_xyz,
                               const int32_t* qry_batch, int nq, int k, int32_t* nbr_idx,
                               void* stream) {
  IRX_REQUIRE(nq >= 0 && k >= 1, "irx_knn_batched: bad sizes");
  if (nq == 0) return IRX_OK;
  IRX_REQUIRE(sup_xyz && sup_offsets && qry_xyz && qry_batch && nbr_idx, "irx_knn_batched: null pointer");
  k_knn<<<irx_cdiv(nq, 4), 256, 0, S(stream)>>>(sup_xyz, sup_offsets, qry_xyz, qry_batch, nq, k, nbr_idx);
  IRX_CHECK_LAUNCH("irx_knn_batched");
  return IRX_OK;
}
