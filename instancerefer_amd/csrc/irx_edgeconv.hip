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
// activation of its edges lives in LDS, the four tiny GEMMs are plain fp32 FMA loops (one output channel per thread,
// all edges of the query as independent accumulators), weights stream from L2 (<= 330 KB, shared by all workgroups).
// Backward recomputes the activations (cheaper than storing them), routes d(out) to the arg-max edge of every channel
// and accumulates the parameter gradients into one slab per workgroup (fixed query -> workgroup assignment, fixed order),
// summed afterwards by k_ec_reduce in slab order: deterministic, no atomics.
#include "irx_common.h"

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

#define EC_MAXK 16
#define EC_THREADS 256

struct EcDims {
  int nq, k, fin, nc, hid, fout;
  int ein;   // 3 + 2 nc
  int min_;  // 3 fin
};

struct EcParams {
  const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4;
};

// y[e][o] = act(b[o] + sum_i W[o][i] * x[e][i]) for the k edges of this query; x, y in LDS (row strides ldx / ldy).
template <bool RELU>
__device__ __forceinline__ void ec_linear(const float* __restrict__ W, const float* __restrict__ b, int n_out, int n_in,
                                          const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int k) {
  // thread -> (output channel o, edge half): 128 channels x 2 halves of the edge list
  const int o = threadIdx.x & 127, half = threadIdx.x >> 7;
  const int e0 = half * ((k + 1) / 2), e1 = (half == 0) ? (k + 1) / 2 : k;
  for (int oo = o; oo < n_out; oo += 128) {
    float acc[EC_MAXK / 2];
#pragma unroll
    for (int e = 0; e < EC_MAXK / 2; ++e) acc[e] = b[oo];
    const float* wr = W + (size_t)oo * n_in;
    for (int i = 0; i < n_in; ++i) {
      const float w = wr[i];
#pragma unroll
      for (int e = 0; e < EC_MAXK / 2; ++e)
        if (e0 + e < e1) acc[e] = fmaf(w, x[(e0 + e) * ldx + i], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < EC_MAXK / 2; ++e)
      if (e0 + e < e1) y[(e0 + e) * ldy + oo] = RELU ? fmaxf(acc[e], 0.f) : acc[e];
  }
}

// Forward activations of one query into LDS. Returns nothing; invalid edges (nbr < 0) get zero inputs (masked later).
__device__ __forceinline__ void ec_forward_query(const EcDims& D, const EcParams& P, const float* __restrict__ feats,
                                                 const float* __restrict__ pos, int qrow, const int32_t* __restrict__ nb,
                                                 float* sE, float* sH1, float* sM, float* sH2, float* sOutM) {
  const int k = D.k, fin = D.fin, nc = D.nc;
  // e_in and the x_i / x_j thirds of m_in
  for (int idx = threadIdx.x; idx < k * D.ein; idx += EC_THREADS) {
    const int e = idx / D.ein, i = idx % D.ein;
    const int j = nb[e];
    float v = 0.f;
    if (j >= 0) {
      if (i < 3) v = pos[3 * (size_t)j + i] - pos[3 * (size_t)qrow + i];
      else if (i < 3 + nc) v = feats[(size_t)qrow * fin + (fin - nc) + (i - 3)];
      else v = feats[(size_t)j * fin + (fin - nc) + (i - 3 - nc)];
    }
    sE[e * D.ein + i] = v;
  }
  for (int idx = threadIdx.x; idx < k * fin; idx += EC_THREADS) {
    const int e = idx / fin, i = idx % fin;
    const int j = nb[e];
    sM[e * D.min_ + i] = (j >= 0) ? feats[(size_t)qrow * fin + i] : 0.f;
    sM[e * D.min_ + 2 * fin + i] = (j >= 0) ? feats[(size_t)j * fin + i] : 0.f;
  }
  __syncthreads();
  ec_linear<true>(P.w1, P.b1, D.hid, D.ein, sE, D.ein, sH1, D.hid, k);
  __syncthreads();
  ec_linear<false>(P.w2, P.b2, fin, D.hid, sH1, D.hid, sM + fin, D.min_, k);      // ew lands in the middle third of m_in
  __syncthreads();
  ec_linear<true>(P.w3, P.b3, D.fout, D.min_, sM, D.min_, sH2, D.fout, k);
  __syncthreads();
  ec_linear<false>(P.w4, P.b4, D.fout, D.fout, sH2, D.fout, sOutM, D.fout, k);
  __syncthreads();
}

__device__ __forceinline__ void ec_carve(const EcDims& D, float* base, float*& sE, float*& sH1, float*& sM, float*& sH2,
                                         float*& sOutM) {
  sE = base;
  sH1 = sE + D.k * D.ein;
  sM = sH1 + D.k * D.hid;
  sH2 = sM + D.k * D.min_;
  sOutM = sH2 + D.k * D.fout;
}

static size_t ec_lds_floats(const EcDims& D, bool backward) {
  size_t f = (size_t)D.k * (D.ein + D.hid + D.min_ + 2 * D.fout);
  if (backward) f += (size_t)D.k * (D.fout + D.min_ + D.hid) + D.k;   // dh2, dm_in, dh1 (+ padding)
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
// first: this is the workgroup's first query — the slab element is written, not read (a read-modify-write of the slab in
// global memory is a dependent L2 round trip per element: 320 of them per thread and query were most of this kernel's time;
// with one query per workgroup, the usual case, the slab is write-only)
__device__ __forceinline__ void ec_wgrad(float* __restrict__ part_w, float* __restrict__ part_b, int n_out, int n_in,
                                         const float* __restrict__ dy, int ldy, const float* __restrict__ x, int ldx, int k,
                                         bool first) {
  for (int idx = threadIdx.x; idx < n_out * n_in; idx += EC_THREADS) {
    const int o = idx / n_in, i = idx % n_in;
    float acc = first ? 0.f : part_w[idx];
    for (int e = 0; e < k; ++e) acc = fmaf(dy[e * ldy + o], x[e * ldx + i], acc);
    part_w[idx] = acc;
  }
  for (int o = threadIdx.x; o < n_out; o += EC_THREADS) {
    float acc = first ? 0.f : part_b[o];
    for (int e = 0; e < k; ++e) acc += dy[e * ldy + o];
    part_b[o] = acc;
  }
}

// dx[e][i] = sum_o dy[e][o] * W[o][i]   (optionally masked by act[e][i] > 0)
__device__ __forceinline__ void ec_dgrad(const float* __restrict__ W, int n_out, int n_in, const float* __restrict__ dy,
                                         int ldy, float* __restrict__ dx, int ldx, const float* __restrict__ act, int k) {
  for (int idx = threadIdx.x; idx < k * n_in; idx += EC_THREADS) {
    const int e = idx / n_in, i = idx % n_in;
    float acc = 0.f;
    for (int o = 0; o < n_out; ++o) acc = fmaf(dy[e * ldy + o], W[(size_t)o * n_in + i], acc);
    if (act && !(act[e * ldx + i] > 0.f)) acc = 0.f;
    dx[e * ldx + i] = acc;
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
  float* sDh2 = sOutM + D.k * D.fout;
  float* sDmin = sDh2 + D.k * D.fout;
  float* sDh1 = sDmin + D.k * D.min_;
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
    for (int idx = threadIdx.x; idx < D.k * D.fout; idx += EC_THREADS) {
      const int e = idx / D.fout, o = idx % D.fout;
      sDm[idx] = (arg[(size_t)q * D.fout + o] == e) ? dout[(size_t)q * D.fout + o] : 0.f;
    }
    __syncthreads();
    ec_wgrad(g_w4, g_b4, D.fout, D.fout, sDm, D.fout, sH2, D.fout, D.k, first);
    ec_dgrad(P.w4, D.fout, D.fout, sDm, D.fout, sDh2, D.fout, sH2, D.k);          // through relu(h2)
    __syncthreads();
    ec_wgrad(g_w3, g_b3, D.fout, D.min_, sDh2, D.fout, sM, D.min_, D.k, first);
    ec_dgrad(P.w3, D.fout, D.min_, sDh2, D.fout, sDmin, D.min_, nullptr, D.k);
    __syncthreads();
    if (dmin_out)
      for (int idx = threadIdx.x; idx < D.k * D.min_; idx += EC_THREADS)
        dmin_out[(size_t)q * D.k * D.min_ + idx] = (nb[idx / D.min_] >= 0) ? sDmin[idx] : 0.f;
    ec_wgrad(g_w2, g_b2, D.fin, D.hid, sDmin + D.fin, D.min_, sH1, D.hid, D.k, first);   // d ew = middle third of d m_in
    ec_dgrad(P.w2, D.fin, D.hid, sDmin + D.fin, D.min_, sDh1, D.hid, sH1, D.k);    // through relu(h1)
    __syncthreads();
    ec_wgrad(g_w1, g_b1, D.hid, D.ein, sDh1, D.hid, sE, D.ein, D.k, first);
    if (dmin_out) {                                      // d e_in behind the d m_in block of this query (feature gradients)
      float* sDein = sDm;                                // d m is dead since the layer-4 step
      ec_dgrad(P.w1, D.hid, D.ein, sDh1, D.hid, sDein, D.ein, nullptr, D.k);
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
