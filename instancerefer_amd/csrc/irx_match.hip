// irx_match.hip — the language-instance matching scores (reference models/attribute_module.py:122-126,
// relation_module.py:104-105, scene_module.py:104-106): every candidate's visual vector against the language / scene
// vector of ITS scene, as a cosine.  In PyTorch each head is an index_select + two normalisations + a row dot
// (~9 forward and ~18 backward ATen ops on (Nc, 128..256) tensors); with the training step host-bound that is ~0.5 ms
// per step of pure dispatch.  Here: one launch forward, one backward, deterministic (no atomics).
//   score[i] = <a_i, b_j> / (max(|a_i|, eps) * max(|b_j|, eps)),  j = idx[i]   (idx non-decreasing or arbitrary)
#include "irx_common.h"

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

__device__ __forceinline__ float wave_sum(float v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// one wave per row i
__global__ __launch_bounds__(256) void k_cosine_rows_fwd(const float* __restrict__ a, const float* __restrict__ b,
                                                         const int64_t* __restrict__ idx, int n, int d, float eps,
                                                         float* __restrict__ score, float* __restrict__ norms) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  const float* ar = a + (size_t)i * d;
  const float* br = b + (size_t)(idx ? idx[i] : i) * d;
  float saa = 0.f, sbb = 0.f, sab = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float x = ar[c], y = br[c];
    saa = fmaf(x, x, saa);
    sbb = fmaf(y, y, sbb);
    sab = fmaf(x, y, sab);
  }
  saa = wave_sum(saa);
  sbb = wave_sum(sbb);
  sab = wave_sum(sab);
  const float na = sqrtf(saa), nb = sqrtf(sbb);
  if (lane == 0) {
    score[i] = sab / (fmaxf(na, eps) * fmaxf(nb, eps));
    norms[2 * i] = na;
    norms[2 * i + 1] = nb;
  }
}

// da_i = ds_i * (bhat - s * ahat) / na          (|a| > eps; below the clamp: ds_i * bhat / eps)
__global__ __launch_bounds__(256) void k_cosine_rows_bwd_a(const float* __restrict__ a, const float* __restrict__ b,
                                                           const int64_t* __restrict__ idx, const float* __restrict__ score,
                                                           const float* __restrict__ norms,
                                                           const float* __restrict__ dscore, int n, int d, float eps,
                                                           float* __restrict__ da) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  const float* ar = a + (size_t)i * d;
  const float* br = b + (size_t)(idx ? idx[i] : i) * d;
  const float na = norms[2 * i], nb = norms[2 * i + 1];
  const float ca = fmaxf(na, eps), cb = fmaxf(nb, eps);
  const float g = dscore[i], s = score[i];
  const float kb = g / (ca * cb);                       // coefficient of b
  const float ka = (na > eps) ? g * s / (na * na) : 0.f; // coefficient of a (projection term; absent below the clamp)
  for (int c = lane; c < d; c += 64) da[(size_t)i * d + c] = kb * br[c] - ka * ar[c];
}

// db_j = sum over i with idx[i] == j, in ascending i (deterministic):  ds_i * (ahat - s * bhat) / nb
__global__ __launch_bounds__(256) void k_cosine_rows_bwd_b(const float* __restrict__ a, const float* __restrict__ b,
                                                           const int64_t* __restrict__ idx, const float* __restrict__ score,
                                                           const float* __restrict__ norms,
                                                           const float* __restrict__ dscore, int n, int m, int d, float eps,
                                                           float* __restrict__ db) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= m) return;
  const float* br = b + (size_t)j * d;
  for (int c0 = 0; c0 < d; c0 += 64) {
    const int c = c0 + lane;
    float acc = 0.f;
    const float y = (c < d) ? br[c] : 0.f;
    for (int i = 0; i < n; ++i) {
      if ((idx ? idx[i] : (int64_t)i) != j) continue;    // wave-uniform
      const float na = norms[2 * i], nb = norms[2 * i + 1];
      const float ca = fmaxf(na, eps), cb = fmaxf(nb, eps);
      const float g = dscore[i], s = score[i];
      const float ka = g / (ca * cb);
      const float kb = (nb > eps) ? g * s / (nb * nb) : 0.f;
      if (c < d) acc += ka * a[(size_t)i * d + c] - kb * y;
    }
    if (c < d) db[(size_t)j * d + c] = acc;
  }
}

extern "C" int irx_cosine_rows_fwd(const float* a, const float* b, const int64_t* idx, int n, int d, float eps,
                                   float* score, float* norms, void* stream) {
  IRX_REQUIRE(n >= 0 && d >= 1, "irx_cosine_rows_fwd: bad sizes");
  if (n == 0) return IRX_OK;
  IRX_REQUIRE(a && b && score && norms, "irx_cosine_rows_fwd: null pointer");
  k_cosine_rows_fwd<<<irx_cdiv(n, 4), 256, 0, S(stream)>>>(a, b, idx, n, d, eps, score, norms);
  IRX_CHECK_LAUNCH("irx_cosine_rows_fwd");
  return IRX_OK;
}

extern "C" int irx_cosine_rows_bwd(const float* a, const float* b, const int64_t* idx, const float* score,
                                   const float* norms, const float* dscore, int n, int m, int d, float eps, float* da,
                                   float* db, void* stream) {
  IRX_REQUIRE(n >= 0 && m >= 0 && d >= 1, "irx_cosine_rows_bwd: bad sizes");
  if (da && n > 0) {
    IRX_REQUIRE(a && b && score && norms && dscore, "irx_cosine_rows_bwd: null pointer");
    k_cosine_rows_bwd_a<<<irx_cdiv(n, 4), 256, 0, S(stream)>>>(a, b, idx, score, norms, dscore, n, d, eps, da);
    IRX_CHECK_LAUNCH("irx_cosine_rows_bwd(a)");
  }
  if (db && m > 0) {
    IRX_REQUIRE(n == 0 || (a && b && score && norms && dscore), "irx_cosine_rows_bwd: null pointer");
    k_cosine_rows_bwd_b<<<irx_cdiv(m, 4), 256, 0, S(stream)>>>(a, b, idx, score, norms, dscore, n, m, d, eps, db);
    IRX_CHECK_LAUNCH("irx_cosine_rows_bwd(b)");
  }
  return IRX_OK;
}

// ---- batched ContrastiveLoss (reference lib/loss_helper.py:93-107, called per sample at :248-258) ------------------
// For every scored scene s (rows [off[s], off[s+1]) of the candidate list):  x = gamma * (s1 + s2 + s3);
//   loss_s = keep_s * max( logsumexp_i( x_i * (1 - lab_i) ) - sum_i x_i * lab_i + margin, 0 );   out = sum_s loss_s.
// (The positive slot enters the log-sum-exp as exp(0): reference quirk kept.)  One wave per scene, deterministic;
// act[s] = keep_s if the hinge is active else 0 is kept for the backward, which recomputes the softmax weights.
__global__ __launch_bounds__(64) void k_contrastive_fwd(const float* __restrict__ s1, const float* __restrict__ s2,
                                                        const float* __restrict__ s3, const float* __restrict__ lab,
                                                        const int64_t* __restrict__ off, const float* __restrict__ keep,
                                                        int nseg, float gamma, float margin, float* __restrict__ per,
                                                        float* __restrict__ act, float* __restrict__ lse) {
  const int s = blockIdx.x, lane = threadIdx.x;
  const int lo = (int)off[s], hi = (int)off[s + 1];
  float mx = -INFINITY, sim = 0.f;
  for (int i = lo + lane; i < hi; i += 64) {
    const float x = gamma * (s1[i] + s2[i] + s3[i]);
    const float l = lab[i];
    sim += x * l;
    mx = fmaxf(mx, x * (1.f - l));
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  sim = wave_sum(sim);
  float se = 0.f;
  for (int i = lo + lane; i < hi; i += 64) {
    const float x = gamma * (s1[i] + s2[i] + s3[i]);
    se += expf(x * (1.f - lab[i]) - mx);
  }
  se = wave_sum(se);
  if (lane == 0) {
    const float l = (hi > lo) ? mx + logf(se) : -INFINITY;
    const float v = l - sim + margin;
    per[s] = (v > 0.f) ? keep[s] * v : 0.f;
    act[s] = (v > 0.f) ? keep[s] : 0.f;
    lse[s] = l;
  }
}

__global__ void k_contrastive_sum(const float* __restrict__ per, int nseg, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float t = 0.f;
    for (int s = 0; s < nseg; ++s) t += per[s];      // fixed order
    out[0] = t;
  }
}

// d out / d s{1,2,3}[i] = dout * act_s * gamma * ( softmax_i * (1 - lab_i) - lab_i )
__global__ __launch_bounds__(64) void k_contrastive_bwd(const float* __restrict__ s1, const float* __restrict__ s2,
                                                        const float* __restrict__ s3, const float* __restrict__ lab,
                                                        const int64_t* __restrict__ off, const float* __restrict__ act,
                                                        const float* __restrict__ lse, const float* __restrict__ dout,
                                                        float gamma, float* __restrict__ ds) {
  const int s = blockIdx.x, lane = threadIdx.x;
  const int lo = (int)off[s], hi = (int)off[s + 1];
  const float k = dout[0] * act[s] * gamma, l = lse[s];
  for (int i = lo + lane; i < hi; i += 64) {
    const float x = gamma * (s1[i] + s2[i] + s3[i]);
    const float lb = lab[i];
    const float p = expf(x * (1.f - lb) - l);
    ds[i] = k * (p * (1.f - lb) - lb);
  }
}

extern "C" int irx_contrastive_fwd(const float* s1, const float* s2, const float* s3, const float* lab,
                                   const int64_t* seg_off, const float* keep, int nseg, float gamma, float margin,
                                   float* out, float* per, float* act, float* lse, void* stream) {
  IRX_REQUIRE(nseg >= 0 && out, "irx_contrastive_fwd: bad arguments");
  if (nseg > 0) {
    IRX_REQUIRE(s1 && s2 && s3 && lab && seg_off && keep && per && act && lse, "irx_contrastive_fwd: null pointer");
    k_contrastive_fwd<<<nseg, 64, 0, S(stream)>>>(s1, s2, s3, lab, seg_off, keep, nseg, gamma, margin, per, act, lse);
    IRX_CHECK_LAUNCH("irx_contrastive_fwd");
  }
  k_contrastive_sum<<<1, 64, 0, S(stream)>>>(per, nseg, out);
  IRX_CHECK_LAUNCH("irx_contrastive_fwd(sum)");
  return IRX_OK;
}

extern "C" int irx_contrastive_bwd(const float* s1, const float* s2, const float* s3, const float* lab,
                                   const int64_t* seg_off, const float* act, const float* lse, const float* dout,
                                   int nseg, float gamma, float* ds, void* stream) {
  IRX_REQUIRE(nseg >= 0, "irx_contrastive_bwd: bad arguments");
  if (nseg == 0) return IRX_OK;
  IRX_REQUIRE(s1 && s2 && s3 && lab && seg_off && act && lse && dout && ds, "irx_contrastive_bwd: null pointer");
  k_contrastive_bwd<<<nseg, 64, 0, S(stream)>>>(s1, s2, s3, lab, seg_off, act, lse, dout, gamma, ds);
  IRX_CHECK_LAUNCH("irx_contrastive_bwd");
  return IRX_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// The whole training loss of get_loss (reference lib/loss_helper.py:196-269) as ONE launch, gradients included (round 5):
//   lang_loss = CE(lang_scores [B][n_lang], lang_label)  (:110-118, mean over B)
//   seg_loss  = CE(seg_scores [B][n_seg], seg_label), seg_acc = mean(argmax == label)  (:121-150)
//   ref_loss  = sum_s keep_s * max(logsumexp(x (1 - lab)) - sum(x lab) + margin, 0) / B, x = gamma (s1 + s2 + s3)  (:93-107,248-260)
//   loss      = ref_weight * ref_loss + lang_loss + seg_loss  (:263)
// The loss sits between the last head's forward and the first head's backward, on the step's critical path, and was ~18 forward
// + ~12 backward launches of 2-5 us each. One workgroup: a thread per utterance for the two cross-entropies, a thread per
// scored scene for the contrastive term, fixed-order sums. d_lang / d_seg / d_s hold d loss / d input for an upstream gradient
// of 1 (the backward scales them). out[5] = loss, ref_loss, lang_loss, seg_loss, seg_acc.
#define TL_PT 256
__device__ __forceinline__ float tl_ce_row(const float* __restrict__ x, int n, int label, float inv_b, float* __restrict__ d, int* hit) {
  float mx = -INFINITY;
  int am = 0;
  for (int j = 0; j < n; ++j)
    if (x[j] > mx) { mx = x[j]; am = j; }                // first maximum, like torch.argmax
  float se = 0.f;
  for (int j = 0; j < n; ++j) se += expf(x[j] - mx);
  const float lse = mx + logf(se);
  for (int j = 0; j < n; ++j) d[j] = (expf(x[j] - lse) - (j == label ? 1.f : 0.f)) * inv_b;
  if (hit) *hit = (am == label) ? 1 : 0;
  return lse - x[label];
}

__global__ __launch_bounds__(TL_PT) void k_total_loss(const float* __restrict__ lang_scores, const int64_t* __restrict__ lang_label,
                                                      int B, int n_lang, const float* __restrict__ seg_scores,
                                                      const int64_t* __restrict__ seg_label, int n_seg,
                                                      const float* __restrict__ s1, const float* __restrict__ s2,
                                                      const float* __restrict__ s3, const float* __restrict__ lab,
                                                      const int64_t* __restrict__ seg_off, const float* __restrict__ keep,
                                                      int nscored, float gamma, float margin, float ref_weight, float inv_batch,
                                                      float* __restrict__ out, float* __restrict__ d_lang,
                                                      float* __restrict__ d_seg, float* __restrict__ d_s) {
  __shared__ float sh[3][TL_PT];
  __shared__ int shi[TL_PT];
  const int tid = threadIdx.x;
  const float inv_b = 1.f / (float)B;
  float l_lang = 0.f, l_seg = 0.f;
  int hits = 0;
  for (int b = tid; b < B; b += TL_PT) {
    l_lang += tl_ce_row(lang_scores + (size_t)b * n_lang, n_lang, (int)lang_label[b], inv_b, d_lang + (size_t)b * n_lang, nullptr);
    int h = 0;
    l_seg += tl_ce_row(seg_scores + (size_t)b * n_seg, n_seg, (int)seg_label[b], inv_b, d_seg + (size_t)b * n_seg, &h);
    hits += h;
  }
  float l_ref = 0.f;
  for (int s = tid; s < nscored; s += TL_PT) {
    const int lo = (int)seg_off[s], hi = (int)seg_off[s + 1];
    float mx = -INFINITY, sim = 0.f;
    for (int i = lo; i < hi; ++i) {
      const float x = gamma * (s1[i] + s2[i] + s3[i]);
      const float l = lab[i];
      sim += x * l;
      mx = fmaxf(mx, x * (1.f - l));
    }
    float se = 0.f;
    for (int i = lo; i < hi; ++i) se += expf(gamma * (s1[i] + s2[i] + s3[i]) * (1.f - lab[i]) - mx);
    const float lse = (hi > lo) ? mx + logf(se) : -INFINITY;
    const float v = lse - sim + margin;
    const float act = (v > 0.f) ? keep[s] : 0.f;
    l_ref += (v > 0.f) ? keep[s] * v : 0.f;
    const float k = ref_weight * inv_batch * act * gamma;
    for (int i = lo; i < hi; ++i) {
      const float lb = lab[i];
      const float p = expf(gamma * (s1[i] + s2[i] + s3[i]) * (1.f - lb) - lse);
      d_s[i] = k * (p * (1.f - lb) - lb);
    }
  }
  sh[0][tid] = l_lang; sh[1][tid] = l_seg; sh[2][tid] = l_ref; shi[tid] = hits;
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, b = 0.f, c = 0.f;
    int h = 0;
    for (int t = 0; t < TL_PT; ++t) { a += sh[0][t]; b += sh[1][t]; c += sh[2][t]; h += shi[t]; }     // fixed order
    const float lang = a * inv_b, seg = b * inv_b, ref = c * inv_batch;
    out[0] = ref_weight * ref + lang + seg;
    out[1] = ref;
    out[2] = lang;
    out[3] = seg;
    out[4] = (float)h * inv_b;
  }
}

extern "C" int irx_total_loss(const float* lang_scores, const int64_t* lang_label, int B, int n_lang, const float* seg_scores,
                              const int64_t* seg_label, int n_seg, const float* s1, const float* s2, const float* s3,
                              const float* lab, const int64_t* seg_off, const float* keep, int nscored, float gamma,
                              float margin, float ref_weight, int batch_size, float* out, float* d_lang, float* d_seg, float* d_s,
                              void* stream) {
  IRX_REQUIRE(B >= 1 && n_lang >= 1 && n_seg >= 1 && nscored >= 0 && batch_size >= 1, "irx_total_loss: bad sizes");
  IRX_REQUIRE(lang_scores && lang_label && seg_scores && seg_label && out && d_lang && d_seg, "irx_total_loss: null pointer");
  IRX_REQUIRE(nscored == 0 || (s1 && s2 && s3 && lab && seg_off && keep && d_s), "irx_total_loss: null pointer (scores)");
  k_total_loss<<<1, TL_PT, 0, S(stream)>>>(lang_scores, lang_label, B, n_lang, seg_scores, seg_label, n_seg, s1, s2, s3, lab,
                                          seg_off, keep, nscored, gamma, margin, ref_weight, 1.f / (float)batch_size, out, d_lang,
                                          d_seg, d_s);
  IRX_CHECK_LAUNCH("irx_total_loss");
  return IRX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Language-guided attention pooling of the scene head (reference models/scene_module.py:84-93):
//   logit[b][i] = <feats[b][i][:], lang[b][:]> / sqrt(D);  atten[b][:] = softmax_i(logit[b][:]);  out[b][:] = sum_i atten[b][i] feats[b][i][:]
// Through ATen: bmm, div, softmax, mul, sum forward and ~10 launches backward on a (B, 231, 128) tensor — all on the step's
// critical path (between the scene encoder's forward and its backward). Here one workgroup per scene each way, deterministic.
#define AP_PT 1024
#define AP_NW (AP_PT / 64)
// row-parallel everywhere: wave w owns rows w, w + 16, ...; lane l owns channels l, l + 64, ... (d <= 256: 4 per lane). A thread
// per channel walking all n rows (the first version) was n dependent round trips: slower than the five ATen launches it replaced.
__global__ __launch_bounds__(AP_PT) void k_attn_pool_fwd(const float* __restrict__ feats, const float* __restrict__ lang, int n, int d,
                                                         float scale, float* __restrict__ atten, float* __restrict__ out) {
  extern __shared__ float ap_sm[];                 // [n] logits / weights | [AP_NW][d] per-wave partial pooled rows | [AP_PT] scratch
  float* w = ap_sm;
  float* part = ap_sm + n;
  float* red = part + AP_NW * d;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* f = feats + (size_t)b * n * d;
  const float* l = lang + (size_t)b * d;
  float lv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) lv[q] = (lane + 64 * q < d) ? l[lane + 64 * q] : 0.f;
  for (int i = wave; i < n; i += AP_NW) {
    float p = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (lane + 64 * q < d) p = fmaf(f[(size_t)i * d + lane + 64 * q], lv[q], p);
    p = wave_sum(p);
    if (lane == 0) w[i] = p * scale;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int i = tid; i < n; i += AP_PT) mx = fmaxf(mx, w[i]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
  for (int t = 1; t < AP_NW; ++t) mx = fmaxf(mx, red[t]);
  __syncthreads();
  float se = 0.f;
  for (int i = tid; i < n; i += AP_PT) {
    const float e = expf(w[i] - mx);
    w[i] = e;
    se += e;
  }
  se = wave_sum(se);
  if (lane == 0) red[wave] = se;
  __syncthreads();
  float tot = 0.f;
  for (int t = 0; t < AP_NW; ++t) tot += red[t];     // fixed order, the same in every thread
  const float inv = 1.f / tot;
  __syncthreads();
  for (int i = tid; i < n; i += AP_PT) {
    const float a = w[i] * inv;
    w[i] = a;
    atten[(size_t)b * n + i] = a;
  }
  __syncthreads();
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = wave; i < n; i += AP_NW) {
    const float a = w[i];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (lane + 64 * q < d) acc[q] = fmaf(a, f[(size_t)i * d + lane + 64 * q], acc[q]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (lane + 64 * q < d) part[wave * d + lane + 64 * q] = acc[q];
  __syncthreads();
  for (int c = tid; c < d; c += AP_PT) {
    float t2 = 0.f;
    for (int t = 0; t < AP_NW; ++t) t2 += part[t * d + c];
    out[(size_t)b * d + c] = t2;
  }
}

// d feats[i][:] = atten_i d out + dlogit_i scale lang;  dlogit_i = atten_i (g_i - sum_j atten_j g_j), g_i = <d out, feats_i> + d atten_i;
// d lang = scale sum_i dlogit_i feats_i
__global__ __launch_bounds__(AP_PT) void k_attn_pool_bwd(const float* __restrict__ feats, const float* __restrict__ lang,
                                                         const float* __restrict__ atten, const float* __restrict__ dout,
                                                         const float* __restrict__ datten, int n, int d, float scale,
                                                         float* __restrict__ dfeats, float* __restrict__ dlang) {
  extern __shared__ float ap_sm[];                 // [n] g / dlogit | [AP_NW][d] per-wave partial d lang | [AP_PT] scratch
  float* g = ap_sm;
  float* part = ap_sm + n;
  float* red = part + AP_NW * d;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* f = feats + (size_t)b * n * d;
  const float* a = atten + (size_t)b * n;
  float lv[4], gv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool ok = lane + 64 * q < d;
    lv[q] = ok ? lang[(size_t)b * d + lane + 64 * q] * scale : 0.f;
    gv[q] = ok ? dout[(size_t)b * d + lane + 64 * q] : 0.f;
  }
  for (int i = wave; i < n; i += AP_NW) {
    float p = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (lane + 64 * q < d) p = fmaf(gv[q], f[(size_t)i * d + lane + 64 * q], p);
    p = wave_sum(p);
    if (lane == 0) g[i] = p + (datten ? datten[(size_t)b * n + i] : 0.f);
  }
  __syncthreads();
  float sdot = 0.f;
  for (int i = tid; i < n; i += AP_PT) sdot += a[i] * g[i];
  sdot = wave_sum(sdot);
  if (lane == 0) red[wave] = sdot;
  __syncthreads();
  float dot = 0.f;
  for (int t = 0; t < AP_NW; ++t) dot += red[t];
  __syncthreads();
  for (int i = tid; i < n; i += AP_PT) g[i] = a[i] * (g[i] - dot);       // dlogit
  __syncthreads();
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = wave; i < n; i += AP_NW) {
    const float ai = a[i], gi = g[i];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (lane + 64 * q < d) {
        const size_t o = (size_t)i * d + lane + 64 * q;
        dfeats[(size_t)b * n * d + o] = fmaf(ai, gv[q], gi * lv[q]);
        acc[q] = fmaf(gi, f[o], acc[q]);
      }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (lane + 64 * q < d) part[wave * d + lane + 64 * q] = acc[q];
  __syncthreads();
  for (int c = tid; c < d; c += AP_PT) {
    float t2 = 0.f;
    for (int t = 0; t < AP_NW; ++t) t2 += part[t * d + c];
    dlang[(size_t)b * d + c] = t2 * scale;
  }
}

extern "C" int irx_attn_pool_fwd(const float* feats, const float* lang, int B, int n, int d, float scale, float* atten, float* out,
                                 void* stream) {
  IRX_REQUIRE(B >= 0 && n >= 1 && d >= 1 && d <= 256 && n <= 8192, "irx_attn_pool_fwd: bad sizes (d <= 256, n <= 8192)");
  if (B == 0) return IRX_OK;
  IRX_REQUIRE(feats && lang && atten && out, "irx_attn_pool_fwd: null pointer");
  k_attn_pool_fwd<<<B, AP_PT, (size_t)(n + AP_NW * d + AP_PT) * sizeof(float), S(stream)>>>(feats, lang, n, d, scale, atten, out);
  IRX_CHECK_LAUNCH("irx_attn_pool_fwd");
  return IRX_OK;
}

extern "C" int irx_attn_pool_bwd(const float* feats, const float* lang, const float* atten, const float* dout, const float* datten,
                                 int B, int n, int d, float scale, float* dfeats, float* dlang, void* stream) {
  IRX_REQUIRE(B >= 0 && n >= 1 && d >= 1 && d <= 256 && n <= 8192, "irx_attn_pool_bwd: bad sizes (d <= 256, n <= 8192)");
  if (B == 0) return IRX_OK;
  IRX_REQUIRE(feats && lang && atten && dout && dfeats && dlang, "irx_attn_pool_bwd: null pointer");
  k_attn_pool_bwd<<<B, AP_PT, (size_t)(n + AP_NW * d + AP_PT) * sizeof(float), S(stream)>>>(feats, lang, atten, dout, datten, n, d, scale, dfeats,
                                                                              dlang);
  IRX_CHECK_LAUNCH("irx_attn_pool_bwd");
  return IRX_OK;
}
