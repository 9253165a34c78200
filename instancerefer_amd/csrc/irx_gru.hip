// irx_gru.hip — persistent GRU recurrence for the language encoder (reference models/lang_module.py:22-28,53-57:
// nn.GRU(256, 128, num_layers=2, bidirectional) on a packed sequence, i.e. cuDNN/MIOpen RNN).
//
// MIOpen's fp32 GRU launches ~2400 tiny element-wise kernels per training step here (rocprof: 8.5 ms GPU and
// ~12 ms of host launch time). The recurrence is inherently sequential but tiny (H = 128), so it is run by ONE
// workgroup per (sequence, direction) that keeps W_hh in VGPRs (one gate row per thread, 128 floats), h in LDS
// (broadcast reads) and walks the T steps with two barriers per step; the time-parallel parts (input
// projections X*W_ih^T, dW_ih, dW_hh, dX) stay dense GEMMs (rocBLAS on MFMA) in the host layer.
// Why the recurrence itself is NOT on the matrix core (round 3, costed): h(t) W_hh^T for all 16 sequences of a direction
// is a 16 x 128 x 384 product = 24 column tiles x 32 v_mfma_f32_16x16x4_f32 per step; one workgroup (it cannot be split
// over CUs without a cross-CU hand-off of h every step) has 4 SIMDs, i.e. 24 * 32 * 32 / 4 = 6144 cycles = 2.6 us per step
// on 2 CUs, against ~0.3 us of FMA work per step per workgroup here, spread over 32 CUs. The steps are latency chains, not
// throughput: what cost 135 us per launch was a global load at the top of every step (now prefetched one step ahead).
// Packed-sequence semantics: forward direction runs t = 0..len-1, reverse runs t = len-1..0; outputs (and
// gradients) at t >= len are zero, exactly what pack_padded_sequence / pad_packed_sequence produce.
#include "irx_common.h"

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

__device__ static inline float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// gi    [B][T][ndir][3H]  input projections incl. b_ih, gate order (r, z, n) as in torch.nn.GRU
// gates [B][T][ndir][4H]  saved (r, z, n, hn_pre) with hn_pre = W_hn h + b_hn
template <int H>
__global__ __launch_bounds__(3 * H) void k_gru_fwd(const float* __restrict__ gi, const int32_t* __restrict__ lengths,
                                                   const float* __restrict__ w_hh, const float* __restrict__ b_hh,
                                                   int T, int ndir, float* __restrict__ out,
                                                   float* __restrict__ gates) {
  __shared__ float sH[H];
  __shared__ float sG[3 * H];
  const int b = blockIdx.x, dir = blockIdx.y;
  const int g = threadIdx.x;          // gate row 0..3H-1
  const int type = g / H, j = g % H;
  int len = lengths[b];
  if (len > T) len = T;
  float w[H];
  const float* wrow = w_hh + ((size_t)dir * 3 * H + g) * H;
#pragma unroll
  for (int c = 0; c < H; c += 4) {
    const float4 v = *reinterpret_cast<const float4*>(wrow + c);
    w[c] = v.x; w[c + 1] = v.y; w[c + 2] = v.z; w[c + 3] = v.w;
  }
  const float bh = b_hh[dir * 3 * H + g];
  if (g < H) sH[g] = 0.f;
  __syncthreads();
  // the input projection of step s + 1 is requested before the arithmetic of step s: a load at the top of every step put a
  // full L2 / HBM round trip on the serial chain (135 us per launch for 30 steps of ~0.3 us of arithmetic each)
  auto gi_at = [&](int step) {
    const int t = dir == 0 ? step : len - 1 - step;
    return gi[(((size_t)b * T + t) * ndir + dir) * 3 * H + g];
  };
  float gi_next = (len > 0) ? gi_at(0) : 0.f;
  for (int step = 0; step < len; ++step) {
    const int t = dir == 0 ? step : len - 1 - step;
    const size_t row = ((size_t)b * T + t) * ndir + dir;
    const float gi_val = gi_next;
    if (step + 1 < len) gi_next = gi_at(step + 1);
    float a0 = bh, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int c = 0; c < H; c += 4) {
      const float4 hv = *reinterpret_cast<const float4*>(&sH[c]);   // broadcast read
      a0 = fmaf(w[c], hv.x, a0);
      a1 = fmaf(w[c + 1], hv.y, a1);
      a2 = fmaf(w[c + 2], hv.z, a2);
      a3 = fmaf(w[c + 3], hv.w, a3);
    }
    const float acc = (a0 + a1) + (a2 + a3);
    if (type < 2) sG[g] = sigmoidf_(gi_val + acc);
    __syncthreads();
    if (type == 2) {
      const float r = sG[j], z = sG[H + j];
      const float n = tanhf(gi_val + r * acc);
      const float hprev = sH[j];
      const float hnew = (1.f - z) * n + z * hprev;
      float* gp = gates + row * 4 * H;
      gp[j] = r; gp[H + j] = z; gp[2 * H + j] = n; gp[3 * H + j] = acc;
      out[((size_t)b * T + t) * ndir * H + dir * H + j] = hnew;
      sH[j] = hnew;   // only thread j reads sH[j] after the barrier above, so the in-place update is safe
    }
    __syncthreads();
  }
  // padded positions
  for (int t = len; t < T; ++t)
    if (g < H) out[((size_t)b * T + t) * ndir * H + dir * H + g] = 0.f;
}

// dout [B][T][ndir*H]; out = forward outputs (h_t); writes dgi, dgh [B][T][ndir][3H]
template <int H>
__global__ __launch_bounds__(3 * H) void k_gru_bwd(const float* __restrict__ dout, const float* __restrict__ out,
                                                   const float* __restrict__ gates, const int32_t* __restrict__ lengths,
                                                   const float* __restrict__ w_hh, int T, int ndir,
                                                   float* __restrict__ dgi, float* __restrict__ dgh) {
  __shared__ float sD[3 * H];      // dgh of the current step
  __shared__ float sP[3 * H];      // partial W_hh^T * dgh per gate type
  const int b = blockIdx.x, dir = blockIdx.y;
  const int g = threadIdx.x;
  const int type = g / H, j = g % H;
  int len = lengths[b];
  if (len > T) len = T;
  // column j of the `type` block of W_hh:  wt[i] = w_hh[dir][type*H + i][j]
  float wt[H];
  const float* wbase = w_hh + ((size_t)dir * 3 * H + type * H) * H + j;
#pragma unroll
  for (int i = 0; i < H; ++i) wt[i] = wbase[(size_t)i * H];
  float dh_carry = 0.f;
  // the six per-step inputs of step s - 1 are requested before the arithmetic of step s (see k_gru_fwd)
  struct In { float r, z, n, hn, hprev, dout; };
  auto load_in = [&](int step) {
    const int t = dir == 0 ? step : len - 1 - step;
    const size_t bt = (size_t)b * T + t;
    const float* gp = gates + (bt * ndir + dir) * 4 * H;
    In v;
    v.r = gp[j]; v.z = gp[H + j]; v.n = gp[2 * H + j]; v.hn = gp[3 * H + j];
    v.hprev = 0.f;
    if (step > 0) {
      const int tp = dir == 0 ? t - 1 : t + 1;
      v.hprev = out[((size_t)b * T + tp) * ndir * H + dir * H + j];
    }
    v.dout = dout[bt * ndir * H + dir * H + j];
    return v;
  };
  In nxt = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (len > 0) nxt = load_in(len - 1);
  for (int step = len - 1; step >= 0; --step) {
    const int t = dir == 0 ? step : len - 1 - step;
    const size_t bt = (size_t)b * T + t;
    const size_t row = bt * ndir + dir;
    const In cur = nxt;
    if (step > 0) nxt = load_in(step - 1);
    const float r = cur.r, z = cur.z, n = cur.n, hn = cur.hn, hprev = cur.hprev;
    const float dh = cur.dout + dh_carry;
    const float dn_pre = dh * (1.f - z) * (1.f - n * n);
    const float dz_pre = dh * (hprev - n) * z * (1.f - z);
    const float dr_pre = dn_pre * hn * r * (1.f - r);
    float my_gi, my_gh;
    if (type == 0) { my_gi = dr_pre; my_gh = dr_pre; }
    else if (type == 1) { my_gi = dz_pre; my_gh = dz_pre; }
    else { my_gi = dn_pre; my_gh = dn_pre * r; }
    dgi[row * 3 * H + g] = my_gi;
    dgh[row * 3 * H + g] = my_gh;
    sD[g] = my_gh;
    __syncthreads();
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int i = 0; i < H; i += 4) {
      const float4 dv = *reinterpret_cast<const float4*>(&sD[type * H + i]);
      a0 = fmaf(wt[i], dv.x, a0);
      a1 = fmaf(wt[i + 1], dv.y, a1);
      a2 = fmaf(wt[i + 2], dv.z, a2);
      a3 = fmaf(wt[i + 3], dv.w, a3);
    }
    sP[g] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    dh_carry = dh * z + sP[j] + sP[H + j] + sP[2 * H + j];
    // next iteration's sD writes are ordered behind this barrier pair (see irx_gru.hip header note)
  }
  for (int t = len; t < T; ++t) {
    const size_t row = ((size_t)b * T + t) * ndir + dir;
    dgi[row * 3 * H + g] = 0.f;
    dgh[row * 3 * H + g] = 0.f;
  }
}

extern "C" int irx_gru_forward(const float* gi, const int32_t* lengths, const float* w_hh, const float* b_hh,
                               int B, int T, int ndir, int H, float* out, float* gates, void* stream) {
  IRX_REQUIRE(B >= 0 && T >= 1 && (ndir == 1 || ndir == 2), "irx_gru_forward: bad sizes");
  IRX_REQUIRE(H == 128 || H == 64, "irx_gru_forward: hidden size %d unsupported (64 or 128)", H);
  if (B == 0) return IRX_OK;
  IRX_REQUIRE(gi && lengths && w_hh && b_hh && out && gates, "irx_gru_forward: null pointer");
  dim3 grid(B, ndir);
  if (H == 128) k_gru_fwd<128><<<grid, 384, 0, S(stream)>>>(gi, lengths, w_hh, b_hh, T, ndir, out, gates);
  else k_gru_fwd<64><<<grid, 192, 0, S(stream)>>>(gi, lengths, w_hh, b_hh, T, ndir, out, gates);
  IRX_CHECK_LAUNCH("irx_gru_forward");
  return IRX_OK;
}

extern "C" int irx_gru_backward(const float* dout, const float* out, const float* gates, const int32_t* lengths,
                                const float* w_hh, int B, int T, int ndir, int H, float* dgi, float* dgh,
                                void* stream) {
  IRX_REQUIRE(B >= 0 && T >= 1 && (ndir == 1 || ndir == 2), "irx_gru_backward: bad sizes");
  IRX_REQUIRE(H == 128 || H == 64, "irx_gru_backward: hidden size %d unsupported (64 or 128)", H);
  if (B == 0) return IRX_OK;
  IRX_REQUIRE(dout && out && gates && lengths && w_hh && dgi && dgh, "irx_gru_backward: null pointer");
  dim3 grid(B, ndir);
  if (H == 128) k_gru_bwd<128><<<grid, 384, 0, S(stream)>>>(dout, out, gates, lengths, w_hh, T, ndir, dgi, dgh);
  else k_gru_bwd<64><<<grid, 192, 0, S(stream)>>>(dout, out, gates, lengths, w_hh, T, ndir, dgi, dgh);
  IRX_CHECK_LAUNCH("irx_gru_backward");
  return IRX_OK;
}
