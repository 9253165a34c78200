// irx_edgeconv.hip — DynamicEdgeConv of the relation module as ONE launch per direction (reference
// models/basic_blocks.py:98-133: torch_geometric MessagePassing(aggr='max') over the kNN instance graph):
//   e_in  = [pos_j - pos_i, cls_i, cls_j]                       (3 + 2 nc)     cls = last nc feature channels
//   h1    = relu(W1 e_in + b1)                                  (hid = 64)     `weight` MLP, layer 0
//   ew    = W2 h1 + b2                                          (fin)          `weight` MLP, layer 2
//   m_in  = [x_i, ew, x_j]                                      (3 fin)
//   h2    = relu(W3 m_in + b3)                                  (fout = 128)   `mlp`, layer 0
//   m     = W4 h2 + b4                                          (fout)         `mlp`, layer 2
//   out_i = max over the valid neighbours j of m                               aggr = 'max'
// In PyTorch this is ~20 forward and ~40 backward ATen ops on (n_query * k, <= 459) tensors — tens of microseconds of
// GPU work behind ~0.5 ms of host dispatch per training step. Here: one workgroup per query (k <= 16 edges), every
// activation of its edges lives in LDS as a 16-row tile (rows >= k are zero padding), and the four tiny GEMMs — and their
// data / weight gradients — are v_mfma_f32_16x16x4_f32 chains: the 16 edges are exactly one MFMA row tile, each wave owns
// every fourth 16-column tile, weights stream from L2 (<= 330 KB, shared by all workgroups). (The first version ran them
// as thread-per-channel FMA loops: 330 us forward / 780 us backward at 153 input features, 100 / 265 us at 25.)
// Backward recomputes the activations (cheaper than storing them), routes d(out) to the arg-max edge of every channel
// and accumulates the parameter gradients into one slab per workgroup (fixed query -> workgroup assignment, fixed order),
// summed afterwards by k_ec_reduce in slab order: deterministic, no atomics.
#include "irx_common.h"

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

#define EC_MAXK 16
#define EC_KP 16          // rows of every LDS activation tile (one MFMA row tile)
#define EC_THREADS 256
typedef float ec_f32x4 __attribute__((ext_vector_type(4)));

struct EcDims {
  int nq, k, fin, nc, hid, fout;
  int ein;   // 3 + 2 nc
  int min_;  // 3 fin
};

struct EcParams {
  const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4;
};

// y[e][o] = act(b[o] + sum_i W[o][i] * x[e][i]) for the 16 rows of the tile; x, y in LDS (row strides ldx / ldy).
// MFMA operands: A[m = edge][kk] = x[m][4 s + kk], B[kk][n] = W[16 nt + n][4 s + kk]; D[row = 4 (lane >> 4) + r][col = lane & 15].
template <bool RELU>
__device__ __forceinline__ void ec_linear(const float* __restrict__ W, const float* __restrict__ b, int n_out, int n_in,
                                          const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, kk = lane >> 4;
  const int nsteps = (n_in + 3) >> 2;
  for (int nt = wave; nt * 16 < n_out; nt += EC_THREADS / 64) {
    const int col = nt * 16 + m;
    const bool cok = col < n_out;
    const float bias = cok ? b[col] : 0.f;
    ec_f32x4 acc = (ec_f32x4){bias, bias, bias, bias};
    const float* wr = W + (size_t)(cok ? col : 0) * n_in;
    for (int st = 0; st < nsteps; ++st) {
      const int i = 4 * st + kk;
      const bool iok = i < n_in;
      const float av = iok ? x[m * ldx + i] : 0.f;
      const float bv = (iok && cok) ? wr[i] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
    if (cok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) y[(4 * kk + r) * ldy + col] = RELU ? fmaxf(acc[r], 0.f) : acc[r];
    }
  }
}

// Forward activations of one query into LDS. Returns nothing; invalid edges (nbr < 0) get zero inputs (masked later).
__device__ __forceinline__ void ec_forward_query(const EcDims& D, const EcParams& P, const float* __restrict__ feats,
                                                 const float* __restrict__ pos, int qrow, const int32_t* __restrict__ nb,
                                                 float* sE, float* sH1, float* sM, float* sH2, float* sOutM) {
  const int k = D.k, fin = D.fin, nc = D.nc;
  // e_in and the x_i / x_j thirds of m_in; rows of missing neighbours and the padding rows k .. 15 are zero
  for (int idx = threadIdx.x; idx < EC_KP * D.ein; idx += EC_THREADS) {
    const int e = idx / D.ein, i = idx % D.ein;
    const int j = (e < k) ? nb[e] : -1;
    float v = 0.f;
    if (j >= 0) {
      if (i < 3) v = pos[3 * (size_t)j + i] - pos[3 * (size_t)qrow + i];
      else if (i < 3 + nc) v = feats[(size_t)qrow * fin + (fin - nc) + (i - 3)];
      else v = feats[(size_t)j * fin + (fin - nc) + (i - 3 - nc)];
    }
    sE[e * D.ein + i] = v;
  }
  for (int idx = threadIdx.x; idx < EC_KP * fin; idx += EC_THREADS) {
    const int e = idx / fin, i = idx % fin;
    const int j = (e < k) ? nb[e] : -1;
    sM[e * D.min_ + i] = (j >= 0) ? feats[(size_t)qrow * fin + i] : 0.f;
    sM[e * D.min_ + 2 * fin + i] = (j >= 0) ? feats[(size_t)j * fin + i] : 0.f;
  }
  __syncthreads();
  ec_linear<true>(P.w1, P.b1, D.hid, D.ein, sE, D.ein, sH1, D.hid);
  __syncthreads();
  ec_linear<false>(P.w2, P.b2, fin, D.hid, sH1, D.hid, sM + fin, D.min_);          // ew lands in the middle third of m_in
  __syncthreads();
  ec_linear<true>(P.w3, P.b3, D.fout, D.min_, sM, D.min_, sH2, D.fout);
  __syncthreads();
  ec_linear<false>(P.w4, P.b4, D.fout, D.fout, sH2, D.fout, sOutM, D.fout);
  __syncthreads();
}

__device__ __forceinline__ void ec_carve(const EcDims& D, float* base, float*& sE, float*& sH1, float*& sM, float*& sH2,
                                         float*& sOutM) {
  sE = base;
  sH1 = sE + EC_KP * D.ein;
  sM = sH1 + EC_KP * D.hid;
  sH2 = sM + EC_KP * D.min_;
  sOutM = sH2 + EC_KP * D.fout;
}

static size_t ec_lds_floats(const EcDims& D, bool backward) {
  size_t f = (size_t)EC_KP * (D.ein + D.hid + D.min_ + 2 * D.fout);
  if (backward) f += (size_t)EC_KP * (D.fout + D.min_ + D.hid) + EC_KP;   // dh2, dm_in, dh1 (+ padding)
  return f;
}

__global__ __launch_bounds__(EC_THREADS) void k_edgeconv_fwd(EcDims D, EcParams P, const float* __restrict__ feats,
                                                             const float* __restrict__ pos,
                                                             const int64_t* __restrict__ qidx,
                                                             const int32_t* __restrict__ nbr, float* __restrict__ out,
                                                             int32_t* __restrict__ arg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int32_t nb[EC_MAXK];
  float *sE, *sH1, *sM, *sH2, *sOutM;
  ec_carve(D, lds, sE, sH1, sM, sH2, sOutM);
  const int q = blockIdx.x;
  if ((int)threadIdx.x < D.k) nb[threadIdx.x] = nbr[(size_t)q * D.k + threadIdx.x];
  __syncthreads();
  ec_forward_query(D, P, feats, pos, (int)qidx[q], nb, sE, sH1, sM, sH2, sOutM);
  for (int o = threadIdx.x; o < D.fout; o += EC_THREADS) {
    float best = -INFINITY;
    int a = -1;
    for (int e = 0; e < D.k; ++e)
      if (nb[e] >= 0) {
        const float v = sOutM[e * D.fout + o];
        if (a < 0 || v > best) { best = v; a = e; }      // torch.max(dim): first maximum
      }
    out[(size_t)q * D.fout + o] = best;
    arg[(size_t)q * D.fout + o] = a;
  }
}

// part[w] += sum_e dy[e][o] * x[e][i] for (o, i) of a layer; this workgroup's slab, same thread every time.
// part_w[o][i] (+)= sum_e dy[e][o] * x[e][i] for (o, i) of a layer: 16 x 16 tiles, the 16 edge rows are the MFMA reduction
// (A[m = o][kk = e] = dy[e][16 mt + m], B[kk = e][n = i] = x[e][16 nt + n]; 4 steps).
// first: this is the workgroup's first query — the slab element is written, not read (a read-modify-write of the slab in
// global memory is a dependent L2 round trip per element; with one query per workgroup, the usual case, the slab is
// write-only). Same thread for the same element every time: deterministic.
__device__ __forceinline__ void ec_wgrad(float* __restrict__ part_w, float* __restrict__ part_b, int n_out, int n_in,
                                         const float* __restrict__ dy, int ldy, const float* __restrict__ x, int ldx,
                                         bool first) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, kk = lane >> 4;
  const int mtn = (n_out + 15) >> 4, ntn = (n_in + 15) >> 4;
  for (int t = wave; t < mtn * ntn; t += EC_THREADS / 64) {
    const int mt = t / ntn, nt = t % ntn;
    const int o = mt * 16 + m, i = nt * 16 + m;
    const bool ook = o < n_out, iok = i < n_in;
    ec_f32x4 acc = (ec_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < EC_KP / 4; ++st) {
      const int e = 4 * st + kk;
      const float av = ook ? dy[e * ldy + o] : 0.f;
      const float bv = iok ? x[e * ldx + i] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
    if (iok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int orow = mt * 16 + 4 * kk + r;
        if (orow < n_out) {
          float* dst = part_w + (size_t)orow * n_in + i;
          *dst = first ? acc[r] : *dst + acc[r];
        }
      }
    }
  }
  for (int o = threadIdx.x; o < n_out; o += EC_THREADS) {
    float acc = first ? 0.f : part_b[o];
    for (int e = 0; e < EC_KP; ++e) acc += dy[e * ldy + o];
    part_b[o] = acc;
  }
}

// dx[e][i] = sum_o dy[e][o] * W[o][i]   (optionally masked by act[e][i] > 0):
// A[m = e][kk] = dy[m][4 s + kk], B[kk][n = i] = W[4 s + kk][16 nt + n].
__device__ __forceinline__ void ec_dgrad(const float* __restrict__ W, int n_out, int n_in, const float* __restrict__ dy,
                                         int ldy, float* __restrict__ dx, int ldx, const float* __restrict__ act) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, kk = lane >> 4;
  const int nsteps = (n_out + 3) >> 2;
  for (int nt = wave; nt * 16 < n_in; nt += EC_THREADS / 64) {
    const int i = nt * 16 + m;
    const bool iok = i < n_in;
    ec_f32x4 acc = (ec_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int st = 0; st < nsteps; ++st) {
      const int o = 4 * st + kk;
      const bool ook = o < n_out;
      const float av = ook ? dy[m * ldy + o] : 0.f;
      const float bv = (ook && iok) ? W[(size_t)o * n_in + i] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
    if (iok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = 4 * kk + r;
        float v = acc[r];
        if (act && !(act[e * ldx + i] > 0.f)) v = 0.f;
        dx[e * ldx + i] = v;
      }
    }
  }
}

__global__ __launch_bounds__(EC_THREADS) void k_edgeconv_bwd(EcDims D, EcParams P, const float* __restrict__ feats,
                                                             const float* __restrict__ pos,
                                                             const int64_t* __restrict__ qidx,
                                                             const int32_t* __restrict__ nbr,
                                                             const float* __restrict__ dout,
                                                             const int32_t* __restrict__ arg, float* __restrict__ part,
                                                             size_t slab, float* __restrict__ dmin_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int32_t nb[EC_MAXK];
  float *sE, *sH1, *sM, *sH2, *sOutM;
  ec_carve(D, lds, sE, sH1, sM, sH2, sOutM);
  float* sDm = sOutM;                                   // dm overwrites m
  float* sDh2 = sOutM + EC_KP * D.fout;
  float* sDmin = sDh2 + EC_KP * D.fout;
  float* sDh1 = sDmin + EC_KP * D.min_;
  float* mine = part + (size_t)blockIdx.x * slab;
  // slab layout = parameter order: w1 b1 w2 b2 w3 b3 w4 b4
  float* g_w1 = mine;
  float* g_b1 = g_w1 + (size_t)D.hid * D.ein;
  float* g_w2 = g_b1 + D.hid;
  float* g_b2 = g_w2 + (size_t)D.fin * D.hid;
  float* g_w3 = g_b2 + D.fin;
  float* g_b3 = g_w3 + (size_t)D.fout * D.min_;
  float* g_w4 = g_b3 + D.fout;
  float* g_b4 = g_w4 + (size_t)D.fout * D.fout;
  (void)slab;
  for (int q = blockIdx.x; q < D.nq; q += gridDim.x) {
    const bool first = q == (int)blockIdx.x;          // every slab element is written by the first query's ec_wgrad calls
    __syncthreads();
    if ((int)threadIdx.x < D.k) nb[threadIdx.x] = nbr[(size_t)q * D.k + threadIdx.x];
    __syncthreads();
    ec_forward_query(D, P, feats, pos, (int)qidx[q], nb, sE, sH1, sM, sH2, sOutM);
    // d m: the arg-max edge of every channel receives d out
    for (int idx = threadIdx.x; idx < EC_KP * D.fout; idx += EC_THREADS) {
      const int e = idx / D.fout, o = idx % D.fout;
      sDm[idx] = (arg[(size_t)q * D.fout + o] == e) ? dout[(size_t)q * D.fout + o] : 0.f;   // padding rows: never the arg-max
    }
    __syncthreads();
    ec_wgrad(g_w4, g_b4, D.fout, D.fout, sDm, D.fout, sH2, D.fout, first);
    ec_dgrad(P.w4, D.fout, D.fout, sDm, D.fout, sDh2, D.fout, sH2);          // through relu(h2)
    __syncthreads();
    ec_wgrad(g_w3, g_b3, D.fout, D.min_, sDh2, D.fout, sM, D.min_, first);
    ec_dgrad(P.w3, D.fout, D.min_, sDh2, D.fout, sDmin, D.min_, nullptr);
    __syncthreads();
    if (dmin_out)
      for (int idx = threadIdx.x; idx < D.k * D.min_; idx += EC_THREADS)
        dmin_out[(size_t)q * D.k * D.min_ + idx] = (nb[idx / D.min_] >= 0) ? sDmin[idx] : 0.f;
    ec_wgrad(g_w2, g_b2, D.fin, D.hid, sDmin + D.fin, D.min_, sH1, D.hid, first);   // d ew = middle third of d m_in
    ec_dgrad(P.w2, D.fin, D.hid, sDmin + D.fin, D.min_, sDh1, D.hid, sH1);    // through relu(h1)
    __syncthreads();
    ec_wgrad(g_w1, g_b1, D.hid, D.ein, sDh1, D.hid, sE, D.ein, first);
    if (dmin_out) {                                      // d e_in behind the d m_in block of this query (feature gradients)
      float* sDein = sDm;                                // d m is dead since the layer-4 step
      ec_dgrad(P.w1, D.hid, D.ein, sDh1, D.hid, sDein, D.ein, nullptr);
      __syncthreads();
      float* dst = dmin_out + (size_t)D.nq * D.k * D.min_ + (size_t)q * D.k * D.ein;
      for (int idx = threadIdx.x; idx < D.k * D.ein; idx += EC_THREADS) dst[idx] = (nb[idx / D.ein] >= 0) ? sDein[idx] : 0.f;
    }
  }
}

struct EcGrads {
  float* p[8];
  size_t end[8];     // running end offset of each parameter inside a slab
};

// grads[param][i] = sum over the workgroup slabs, in slab order
__global__ void k_ec_reduce(const float* __restrict__ part, int nslabs, size_t slab, EcGrads G) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= slab) return;
  float s = 0.f;
  for (int b = 0; b < nslabs; ++b) s += part[(size_t)b * slab + i];
  int j = 0;
  while (i >= G.end[j]) ++j;
  G.p[j][i - (j ? G.end[j - 1] : 0)] = s;
}

static int ec_check(const char* who, int nq, int k, int fin, int nc, int hid, int fout) {
  IRX_REQUIRE(nq >= 0 && k >= 1 && k <= EC_MAXK && fin >= nc && nc >= 0 && hid >= 1 && hid <= 128 && fout >= 1,
              "%s: bad sizes (k <= %d, hid <= 128)", who, EC_MAXK);
  // the backward reuses the dead d_m tile (16 x fout) as the d_e_in tile (16 x (3 + 2 nc)): reference shapes have
  // 3 + 2*18 = 39 <= 128 (models/relation_module.py:27-29)
  IRX_REQUIRE(3 + 2 * nc <= fout, "%s: 3 + 2*nc = %d exceeds fout = %d (edge-input gradient tile)", who, 3 + 2 * nc, fout);
  return IRX_OK;
}

static EcDims ec_dims(int nq, int k, int fin, int nc, int hid, int fout) {
  EcDims D;
  D.nq = nq; D.k = k; D.fin = fin; D.nc = nc; D.hid = hid; D.fout = fout;
  D.ein = 3 + 2 * nc;
  D.min_ = 3 * fin;
  return D;
}

static size_t ec_param_floats(const EcDims& D) {
  return (size_t)D.hid * D.ein + D.hid + (size_t)D.fin * D.hid + D.fin + (size_t)D.fout * D.min_ + D.fout +
         (size_t)D.fout * D.fout + D.fout;
}

// one query per workgroup up to 512 queries (two workgroups per CU): the slabs stay write-only (ec_wgrad)
static int ec_blocks(int nq) { return nq < 512 ? nq : 512; }

extern "C" size_t irx_edgeconv_workspace_bytes(int nq, int k, int fin, int nc, int hid, int fout) {
  if (nq <= 0) return 0;
  const EcDims D = ec_dims(nq, k, fin, nc, hid, fout);
  return (size_t)ec_blocks(nq) * ec_param_floats(D) * sizeof(float);
}

extern "C" int irx_edgeconv_max_fwd(const float* feats, const float* pos, const int64_t* qidx, const int32_t* nbr,
                                    int nq, int k, int fin, int nc, int hid, int fout, const float* const* params,
                                    float* out, int32_t* arg, void* stream) {
  int rc = ec_check("irx_edgeconv_max_fwd", nq, k, fin, nc, hid, fout);
  if (rc) return rc;
  if (nq == 0) return IRX_OK;
  IRX_REQUIRE(feats && pos && qidx && nbr && params && out && arg, "irx_edgeconv_max_fwd: null pointer");
  const EcDims D = ec_dims(nq, k, fin, nc, hid, fout);
  const size_t lds = ec_lds_floats(D, false) * sizeof(float);
  IRX_REQUIRE(lds <= 150 * 1024, "irx_edgeconv_max_fwd: %zu bytes of LDS per query exceed the CU (fin = %d, k = %d)", lds, fin, k);
  EcParams P = {params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7]};
  IRX_CHECK_HIP(hipFuncSetAttribute((const void*)k_edgeconv_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                "irx_edgeconv_max_fwd(lds)");
  k_edgeconv_fwd<<<nq, EC_THREADS, lds, S(stream)>>>(D, P, feats, pos, qidx, nbr, out, arg);
  IRX_CHECK_LAUNCH("irx_edgeconv_max_fwd");
  return IRX_OK;
}

// grads[0..7]: w1 b1 w2 b2 w3 b3 w4 b4 (shapes of the parameters). dmin_out (optional): [nq][k][3 fin] d m_in per edge
// (thirds: d x_i, d ew, d x_j) followed by [nq][k][3 + 2 nc] d e_in per edge, for callers that need the gradient of the
// node features.
extern "C" int irx_edgeconv_max_bwd(const float* feats, const float* pos, const int64_t* qidx, const int32_t* nbr,
                                    int nq, int k, int fin, int nc, int hid, int fout, const float* const* params,
                                    const float* dout, const int32_t* arg, float* const* grads, float* dmin_out,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  int rc = ec_check("irx_edgeconv_max_bwd", nq, k, fin, nc, hid, fout);
  if (rc) return rc;
  IRX_REQUIRE(params && grads, "irx_edgeconv_max_bwd: null pointer");
  const EcDims D = ec_dims(nq, k, fin, nc, hid, fout);
  const size_t sizes[8] = {(size_t)D.hid * D.ein, (size_t)D.hid, (size_t)D.fin * D.hid, (size_t)D.fin,
                           (size_t)D.fout * D.min_, (size_t)D.fout, (size_t)D.fout * D.fout, (size_t)D.fout};
  if (nq == 0) {
    for (int i = 0; i < 8; ++i)
      IRX_CHECK_HIP(hipMemsetAsync(grads[i], 0, sizes[i] * sizeof(float), S(stream)), "irx_edgeconv_max_bwd(memset)");
    return IRX_OK;
  }
  IRX_REQUIRE(feats && pos && qidx && nbr && dout && arg, "irx_edgeconv_max_bwd: null pointer");
  const size_t need = irx_edgeconv_workspace_bytes(nq, k, fin, nc, hid, fout);
  if (!workspace || workspace_bytes < need) {
    irx_set_error("irx_edgeconv_max_bwd: workspace %zu < %zu", workspace_bytes, need);
    return IRX_ERR_WORKSPACE;
  }
  const size_t lds = ec_lds_floats(D, true) * sizeof(float);
  IRX_REQUIRE(lds <= 150 * 1024, "irx_edgeconv_max_bwd: %zu bytes of LDS per query exceed the CU (fin = %d, k = %d)", lds, fin, k);
  EcParams P = {params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7]};
  const int blocks = ec_blocks(nq);
  const size_t slab = ec_param_floats(D);
  IRX_CHECK_HIP(hipFuncSetAttribute((const void*)k_edgeconv_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                "irx_edgeconv_max_bwd(lds)");
  k_edgeconv_bwd<<<blocks, EC_THREADS, lds, S(stream)>>>(D, P, feats, pos, qidx, nbr, dout, arg, (float*)workspace, slab,
                                                         dmin_out);
  IRX_CHECK_LAUNCH("irx_edgeconv_max_bwd");
  EcGrads G;
  size_t off = 0;
  for (int i = 0; i < 8; ++i) {
    IRX_REQUIRE(grads[i], "irx_edgeconv_max_bwd: null gradient pointer %d", i);
    off += sizes[i];
    G.p[i] = grads[i];
    G.end[i] = off;
  }
  k_ec_reduce<<<irx_cdiv((long long)slab, 256), 256, 0, S(stream)>>>((const float*)workspace, blocks, slab, G);
  IRX_CHECK_LAUNCH("irx_edgeconv_max_bwd(reduce)");
  return IRX_OK;
}
