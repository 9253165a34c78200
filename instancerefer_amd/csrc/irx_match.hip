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
    // the rows of this target 64 at a time: one idx read per lane, then only the matching rows, in ascending i (the order of the
    // one-row-at-a-time walk: bit-identical) — that walk was n dependent global reads per wave (14-26 us for ~100 rows)
    for (int i0 = 0; i0 < n; i0 += 64) {
      const int ii = i0 + lane;
      unsigned long long hit = __ballot(ii < n && (idx ? idx[ii] : (int64_t)ii) == (int64_t)j);
      while (hit) {
        const int i = i0 + __builtin_ctzll(hit);
        hit &= hit - 1;
        const float na = norms[2 * i], nb = norms[2 * i + 1];
        const float ca = fmaxf(na, eps), cb = fmaxf(nb, eps);
        const float g = dscore[i], s = score[i];
        const float ka = g / (ca * cb);
        const float kb = (nb > eps) ? g * s / (nb * nb) : 0.f;
        if (c < d) acc += ka * a[(size_t)i * d + c] - kb * y;
      }
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

// ---------------------------------------------------------------------------------------------------------------------
// The four attention heads of the language module (reference models/lang_module.py:61-83) as one launch each way:
//   logit[b][t][h] = <feats[b][t][:], w_h> + b_h;  p = softmax over ALL T positions (padded ones included: reference quirk);
//   q = p * [t < len_b];  att = q / sum_t q;  pooled[b][h][:] = sum_t att[b][t][h] * embed[b][t][:]
// feats [B][T][O] = GRU output, embed [B][T][E] = projected word embeddings. Through ATen: 2 cat + matmul + add + softmax + mul + sum +
// div + transpose + bmm forward and ~20 nodes backward, issued by the language helper thread — which the training thread waits
// for before it can issue the attribute head. One workgroup per utterance, deterministic.
#define LP_PT 256
struct LpHeads { const float* w[4]; const float* b[4]; };

__global__ __launch_bounds__(LP_PT) void k_lang_pool_fwd(const float* __restrict__ feats, const float* __restrict__ embed,
                                                         const int64_t* __restrict__ len, int T, int O, int E, LpHeads hd,
                                                         float* __restrict__ att, float* __restrict__ prob, float* __restrict__ qsum,
                                                         float* __restrict__ pooled) {
  extern __shared__ float lp_sm[];                  // [T][4] logits -> p -> att
  float* a = lp_sm;
  __shared__ float s_q[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* f = feats + (size_t)b * T * O;
  const float* e = embed + (size_t)b * T * E;
  const int n = (int)len[b];
  for (int t = wave; t < T; t += LP_PT / 64) {
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    for (int o = lane; o < O; o += 64) {
      const float x = f[(size_t)t * O + o];
      p0 = fmaf(x, hd.w[0][o], p0); p1 = fmaf(x, hd.w[1][o], p1); p2 = fmaf(x, hd.w[2][o], p2); p3 = fmaf(x, hd.w[3][o], p3);
    }
    p0 = wave_sum(p0); p1 = wave_sum(p1); p2 = wave_sum(p2); p3 = wave_sum(p3);
    if (lane == 0) {
      a[t * 4 + 0] = p0 + hd.b[0][0]; a[t * 4 + 1] = p1 + hd.b[1][0]; a[t * 4 + 2] = p2 + hd.b[2][0]; a[t * 4 + 3] = p3 + hd.b[3][0];
    }
  }
  __syncthreads();
  if (tid < 4) {                                    // T <= 126: one thread per head walks the positions (fixed order)
    const int h = tid;
    float mx = -INFINITY;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, a[t * 4 + h]);
    float se = 0.f;
    for (int t = 0; t < T; ++t) se += expf(a[t * 4 + h] - mx);
    float sq = 0.f;
    for (int t = 0; t < T; ++t) {
      const float p = expf(a[t * 4 + h] - mx) / se;
      prob[((size_t)b * T + t) * 4 + h] = p;
      const float q = (t < n) ? p : 0.f;
      a[t * 4 + h] = q;
      sq += q;
    }
    s_q[h] = sq;
    qsum[b * 4 + h] = sq;
    for (int t = 0; t < T; ++t) {
      const float v = a[t * 4 + h] / sq;
      a[t * 4 + h] = v;
      att[((size_t)b * T + t) * 4 + h] = v;
    }
  }
  __syncthreads();
  for (int c = tid; c < E; c += LP_PT) {
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
    int t = 0;
    for (; t + 4 <= T; t += 4) {                    // 4 rows' loads in flight
      float x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = e[(size_t)(t + u) * E + c];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        o0 = fmaf(a[(t + u) * 4 + 0], x[u], o0); o1 = fmaf(a[(t + u) * 4 + 1], x[u], o1);
        o2 = fmaf(a[(t + u) * 4 + 2], x[u], o2); o3 = fmaf(a[(t + u) * 4 + 3], x[u], o3);
      }
    }
    for (; t < T; ++t) {
      const float x = e[(size_t)t * E + c];
      o0 = fmaf(a[t * 4 + 0], x, o0); o1 = fmaf(a[t * 4 + 1], x, o1); o2 = fmaf(a[t * 4 + 2], x, o2); o3 = fmaf(a[t * 4 + 3], x, o3);
    }
    float* po = pooled + (size_t)b * 4 * E;
    po[c] = o0; po[E + c] = o1; po[2 * E + c] = o2; po[3 * E + c] = o3;
  }
}

// part_w [B][4][O], part_b [B][4]: this utterance's share of the head weights' / biases' gradients (summed by k_lang_pool_wsum)
__global__ __launch_bounds__(LP_PT) void k_lang_pool_bwd(const float* __restrict__ feats, const float* __restrict__ embed,
                                                         const int64_t* __restrict__ len, int T, int O, int E, LpHeads hd,
                                                         const float* __restrict__ att, const float* __restrict__ prob,
                                                         const float* __restrict__ qsum, const float* __restrict__ dpooled,
                                                         const float* __restrict__ datt, float* __restrict__ dfeats,
                                                         float* __restrict__ dembed, float* __restrict__ part_w,
                                                         float* __restrict__ part_b) {
  extern __shared__ float lp_sm[];                  // [T][4] g_att -> d logit | [T][4] att
  float* g = lp_sm;
  float* a = lp_sm + 4 * T;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* f = feats + (size_t)b * T * O;
  const float* e = embed + (size_t)b * T * E;
  const float* dp = dpooled + (size_t)b * 4 * E;
  const int n = (int)len[b];
  for (int i = tid; i < 4 * T; i += LP_PT) a[i] = att[(size_t)b * T * 4 + i];
  for (int t = wave; t < T; t += LP_PT / 64) {
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    for (int c = lane; c < E; c += 64) {
      const float x = e[(size_t)t * E + c];
      p0 = fmaf(dp[c], x, p0); p1 = fmaf(dp[E + c], x, p1); p2 = fmaf(dp[2 * E + c], x, p2); p3 = fmaf(dp[3 * E + c], x, p3);
    }
    p0 = wave_sum(p0); p1 = wave_sum(p1); p2 = wave_sum(p2); p3 = wave_sum(p3);
    if (lane == 0) {
      const float* da = datt ? datt + ((size_t)b * T + t) * 4 : nullptr;
      g[t * 4 + 0] = p0 + (da ? da[0] : 0.f); g[t * 4 + 1] = p1 + (da ? da[1] : 0.f);
      g[t * 4 + 2] = p2 + (da ? da[2] : 0.f); g[t * 4 + 3] = p3 + (da ? da[3] : 0.f);
    }
  }
  __syncthreads();
  if (tid < 4) {
    const int h = tid;
    const float s = qsum[b * 4 + h];
    float d1 = 0.f;
    for (int t = 0; t < T; ++t) d1 += a[t * 4 + h] * g[t * 4 + h];
    float d2 = 0.f;
    for (int t = 0; t < T; ++t) {
      const float dpv = (t < n) ? (g[t * 4 + h] - d1) / s : 0.f;       // d p_t = d q_t * mask
      g[t * 4 + h] = dpv;
      d2 += prob[((size_t)b * T + t) * 4 + h] * dpv;
    }
    float sb = 0.f;
    for (int t = 0; t < T; ++t) {
      const float dl = prob[((size_t)b * T + t) * 4 + h] * (g[t * 4 + h] - d2);
      g[t * 4 + h] = dl;
      sb += dl;
    }
    part_b[b * 4 + h] = sb;
  }
  __syncthreads();
  for (int c = tid; c < E; c += LP_PT) {            // d embed[t][c] = sum_h att[t][h] d pooled[h][c]
    const float q0 = dp[c], q1 = dp[E + c], q2 = dp[2 * E + c], q3 = dp[3 * E + c];
    for (int t = 0; t < T; ++t)
      dembed[((size_t)b * T + t) * E + c] = fmaf(a[t * 4 + 0], q0, fmaf(a[t * 4 + 1], q1, fmaf(a[t * 4 + 2], q2, a[t * 4 + 3] * q3)));
  }
  for (int o = tid; o < O; o += LP_PT) {            // d feats[t][o] = sum_h d logit[t][h] w_h[o];  d w_h[o] += sum_t d logit[t][h] feats[t][o]
    const float w0 = hd.w[0][o], w1 = hd.w[1][o], w2 = hd.w[2][o], w3 = hd.w[3][o];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int t = 0;
    for (; t + 4 <= T; t += 4) {
      float x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = f[(size_t)(t + u) * O + o];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* gl = g + (t + u) * 4;
        dfeats[((size_t)b * T + t + u) * O + o] = fmaf(gl[0], w0, fmaf(gl[1], w1, fmaf(gl[2], w2, gl[3] * w3)));
        s0 = fmaf(gl[0], x[u], s0); s1 = fmaf(gl[1], x[u], s1); s2 = fmaf(gl[2], x[u], s2); s3 = fmaf(gl[3], x[u], s3);
      }
    }
    for (; t < T; ++t) {
      const float x = f[(size_t)t * O + o];
      const float* gl = g + t * 4;
      dfeats[((size_t)b * T + t) * O + o] = fmaf(gl[0], w0, fmaf(gl[1], w1, fmaf(gl[2], w2, gl[3] * w3)));
      s0 = fmaf(gl[0], x, s0); s1 = fmaf(gl[1], x, s1); s2 = fmaf(gl[2], x, s2); s3 = fmaf(gl[3], x, s3);
    }
    float* pw = part_w + (size_t)b * 4 * O;
    pw[o] = s0; pw[O + o] = s1; pw[2 * O + o] = s2; pw[3 * O + o] = s3;
  }
}

__global__ void k_lang_pool_wsum(const float* __restrict__ part_w, const float* __restrict__ part_b, int B, int O,
                                 float* __restrict__ dw, float* __restrict__ db) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over 4 * O weights, then 4 biases
  if (i < 4 * O) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += part_w[(size_t)b * 4 * O + i];
    dw[i] = s;
  } else if (i < 4 * O + 4) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += part_b[b * 4 + (i - 4 * O)];
    db[i - 4 * O] = s;
  }
}

extern "C" int irx_lang_pool_fwd(const float* feats, const float* embed, const int64_t* len, int B, int T, int O, int E,
                                 const float* const* w, const float* const* bias, float* att, float* prob, float* qsum, float* pooled,
                                 void* stream) {
  IRX_REQUIRE(B >= 0 && T >= 1 && T <= 1024 && O >= 1 && E >= 1, "irx_lang_pool_fwd: bad sizes");
  if (B == 0) return IRX_OK;
  IRX_REQUIRE(feats && embed && len && w && bias && att && prob && qsum && pooled, "irx_lang_pool_fwd: null pointer");
  LpHeads hd;
  for (int h = 0; h < 4; ++h) {
    IRX_REQUIRE(w[h] && bias[h], "irx_lang_pool_fwd: null head parameter");
    hd.w[h] = w[h]; hd.b[h] = bias[h];
  }
  k_lang_pool_fwd<<<B, LP_PT, (size_t)4 * T * sizeof(float), S(stream)>>>(feats, embed, len, T, O, E, hd, att, prob, qsum, pooled);
  IRX_CHECK_LAUNCH("irx_lang_pool_fwd");
  return IRX_OK;
}

// dw [4][O], db [4]; part: scratch of B * 4 * (O + 1) floats
extern "C" int irx_lang_pool_bwd(const float* feats, const float* embed, const int64_t* len, int B, int T, int O, int E,
                                 const float* const* w, const float* att, const float* prob, const float* qsum, const float* dpooled,
                                 const float* datt, float* dfeats, float* dembed, float* dw, float* db, float* part, void* stream) {
  IRX_REQUIRE(B >= 1 && T >= 1 && T <= 1024 && O >= 1 && E >= 1, "irx_lang_pool_bwd: bad sizes");
  IRX_REQUIRE(feats && embed && len && w && att && prob && qsum && dpooled && dfeats && dembed && dw && db && part,
              "irx_lang_pool_bwd: null pointer");
  LpHeads hd;
  for (int h = 0; h < 4; ++h) { hd.w[h] = w[h]; hd.b[h] = nullptr; }
  float* part_w = part;
  float* part_b = part + (size_t)B * 4 * O;
  k_lang_pool_bwd<<<B, LP_PT, (size_t)8 * T * sizeof(float), S(stream)>>>(feats, embed, len, T, O, E, hd, att, prob, qsum, dpooled,
                                                                          datt, dfeats, dembed, part_w, part_b);
  IRX_CHECK_LAUNCH("irx_lang_pool_bwd");
  k_lang_pool_wsum<<<irx_cdiv(4 * O + 4, 256), 256, 0, S(stream)>>>(part_w, part_b, B, O, dw, db);
  IRX_CHECK_LAUNCH("irx_lang_pool_bwd(sum)");
  return IRX_OK;
}
