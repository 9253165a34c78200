// irx_spconv.hip — sparse 3D convolution for gfx950: forward / data-gradient (one kernel,
// output-stationary gather -> LDS -> v_mfma_f32_16x16x4_f32) and weight-gradient
// (row-chunk gather -> LDS -> MFMA with the voxel rows as the reduction dimension).
//
// Replaces torchsparse's per-offset gather -> cuBLAS GEMM -> scatter-add conv (27 or 8
// launches x 3 per layer, float atomics) reached from spnn.Conv3d at
// models/basic_blocks.py:14-19,32-43 (reference tree).  Design:
//   * voxels are in Morton order, so a 64-row output tile is spatially compact: for most of the
//     27 offsets either every row or no row of a tile has a neighbour. A wave-wide ballot over the
//     tile's table column decides, block-uniformly, whether offset k is skipped entirely.
//   * each output row is produced by exactly one wave -> no atomics, bit-reproducible.
//   * fp32 MFMA (16x16x4) is an exact fp32 FMA chain, so results stay within fp32 round-off of
//     the reference's cuBLAS SGEMM + atomic adds.
#include "irx_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_wgrad_reduce(const float* __restrict__ part, int S, size_t elems, float* __restrict__ dw,
                               int accumulate = 0, int out_bf = 0);

#define SC_TM 64  // output rows per workgroup (4 waves x 16 rows)
#define SC_KC 32  // reduction (input-channel) chunk staged per barrier pair
#define SC_LDA (SC_KC + 2)   // LDS row stride (dwords) of the gathered A tile: (2m+g)%32 distinct

// ---------------------------------------------------------------------------------------------
// y[q][n] = sum_k sum_c x[nbr[k'][q]][c] * Wk[c][n]
//   !TRANS_W : Wk[c][n] = w[(k*cin + c)*cout + n]
//    TRANS_W : Wk[c][n] = w[(k*cout + n)*cin + c]
// BN = output-channel tile per workgroup (32 / 64 / 128).  VEC: cin % 4 == 0 && cout % 4 == 0.
template <int BN, bool TRANS_W, bool VEC>
__global__ __launch_bounds__(256) void k_spconv_fwd(const float* __restrict__ x,
                                                    const float* __restrict__ w,
                                                    const int32_t* __restrict__ nbr, int ld,
                                                    int n_out, int K, int cin, int cout, int flip_k,
                                                    float* __restrict__ y) {
  constexpr int NT = BN / 16;
  constexpr int LDB = TRANS_W ? (SC_KC + 2) : (BN + 16);
  constexpr int B_ELEMS = TRANS_W ? BN * LDB : SC_KC * LDB;
  __shared__ __attribute__((aligned(16))) float sA[SC_TM * SC_LDA];
  __shared__ __attribute__((aligned(16))) float sB[B_ELEMS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int m = lane & 15;
  const int g = lane >> 4;
  const int q0 = blockIdx.x * SC_TM;
  const int n0 = blockIdx.y * BN;

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int k = 0; k < K; ++k) {
    const int kt = flip_k ? (K - 1 - k) : k;
    // every wave reads the same 64 table entries (lane == tile row) -> block-uniform ballot
    int my = -1;
    if (q0 + lane < n_out) my = nbr[(size_t)kt * ld + q0 + lane];
    const unsigned long long valid = __ballot(my >= 0);
    if (valid == 0ull) continue;  // no row of this tile has a neighbour at offset k
    const bool wave_any = ((valid >> (16 * wave)) & 0xFFFFull) != 0ull;

    for (int c0 = 0; c0 < cin; c0 += SC_KC) {
      __syncthreads();  // previous chunk's fragment reads are done
      // ---- gather A: rows r and r+32, 8 threads x 16 B per row ----
      {
        const int r = tid >> 3;
        const int c4 = (tid & 7) * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = r + 32 * h;
          const int idx = __shfl(my, row);
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (idx >= 0) {
            const float* src = x + (size_t)idx * cin + c0 + c4;
            if (VEC) {
              if (c0 + c4 < cin) v = *reinterpret_cast<const float4*>(src);
            } else {
              if (c0 + c4 + 0 < cin) v.x = src[0];
              if (c0 + c4 + 1 < cin) v.y = src[1];
              if (c0 + c4 + 2 < cin) v.z = src[2];
              if (c0 + c4 + 3 < cin) v.w = src[3];
            }
          }
          float* dst = &sA[row * SC_LDA + c4];
          *reinterpret_cast<float2*>(dst) = make_float2(v.x, v.y);
          *reinterpret_cast<float2*>(dst + 2) = make_float2(v.z, v.w);
        }
      }
      // ---- stage the weight chunk ----
      if (!TRANS_W) {
        // rows c0..c0+KC-1 of W[k], columns n0..n0+BN-1; BN/4 float4 per row
        constexpr int F4_PER_ROW = BN / 4;
        constexpr int TOTAL = SC_KC * F4_PER_ROW;
#pragma unroll
        for (int f = tid; f < TOTAL; f += 256) {
          const int kc = f / F4_PER_ROW;
          const int n4 = (f % F4_PER_ROW) * 4;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (c0 + kc < cin) {
            const float* src = w + ((size_t)k * cin + c0 + kc) * cout + n0 + n4;
            if (VEC) {
              if (n0 + n4 < cout) v = *reinterpret_cast<const float4*>(src);
            } else {
              if (n0 + n4 + 0 < cout) v.x = src[0];
              if (n0 + n4 + 1 < cout) v.y = src[1];
              if (n0 + n4 + 2 < cout) v.z = src[2];
              if (n0 + n4 + 3 < cout) v.w = src[3];
            }
          }
          *reinterpret_cast<float4*>(&sB[kc * LDB + n4]) = v;
        }
      } else {
        // rows n0..n0+BN-1 of w[k] ([cout][cin]), columns c0..c0+KC-1; KC/4 float4 per row
        constexpr int F4_PER_ROW = SC_KC / 4;
        constexpr int TOTAL = BN * F4_PER_ROW;
#pragma unroll
        for (int f = tid; f < TOTAL; f += 256) {
          const int n = f / F4_PER_ROW;
          const int kc4 = (f % F4_PER_ROW) * 4;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (n0 + n < cout) {
            const float* src = w + ((size_t)k * cout + n0 + n) * cin + c0 + kc4;
            if (VEC) {
              if (c0 + kc4 < cin) v = *reinterpret_cast<const float4*>(src);
            } else {
              if (c0 + kc4 + 0 < cin) v.x = src[0];
              if (c0 + kc4 + 1 < cin) v.y = src[1];
              if (c0 + kc4 + 2 < cin) v.z = src[2];
              if (c0 + kc4 + 3 < cin) v.w = src[3];
            }
          }
          float* dst = &sB[n * LDB + kc4];
          *reinterpret_cast<float2*>(dst) = make_float2(v.x, v.y);
          *reinterpret_cast<float2*>(dst + 2) = make_float2(v.z, v.w);
        }
      }
      __syncthreads();
      if (wave_any) {
        const float* pa = &sA[(16 * wave + m) * SC_LDA + g];
#pragma unroll
        for (int ks = 0; ks < SC_KC / 4; ++ks) {
          const float a = pa[ks * 4];
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            float b;
            if (!TRANS_W)
              b = sB[(ks * 4 + g) * LDB + t * 16 + m];
            else
              b = sB[(t * 16 + m) * LDB + ks * 4 + g];
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- epilogue: C/D layout col = lane&15, row = (lane>>4)*4 + reg ----
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = n0 + t * 16 + m;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = q0 + 16 * wave + g * 4 + r;
      if (row < n_out && col < cout) y[(size_t)row * cout + col] = acc[t][r];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: part[s][k][c][n] = sum_{q in split s} x[nbr[k][q]][c] * dy[q][n]
// Workgroup = (split s, offset k, 64x64 (c, n) tile); reduction over voxel rows in chunks of 64.
#define WG_TQ 64
#define WG_LD (64 + 16)  // LDS row stride (dwords): consecutive rows 16 banks apart

template <bool VEC>
__global__ __launch_bounds__(256) void k_spconv_wgrad(const float* __restrict__ x,
                                                      const float* __restrict__ dy,
                                                      const int32_t* __restrict__ nbr, int ld,
                                                      int n_out, int K, int cin, int cout,
                                                      int rows_per_split, int n_ctile_n,
                                                      float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float sX[WG_TQ * WG_LD];
  __shared__ __attribute__((aligned(16))) float sD[WG_TQ * WG_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int m = lane & 15;
  const int g = lane >> 4;
  const int s = blockIdx.x;
  const int k = blockIdx.y;
  const int c0 = (blockIdx.z / n_ctile_n) * 64;
  const int n0 = (blockIdx.z % n_ctile_n) * 64;
  const int qbeg = s * rows_per_split;
  int qend = qbeg + rows_per_split;
  if (qend > n_out) qend = n_out;

  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int q0 = qbeg; q0 < qend; q0 += WG_TQ) {
    int my = -1;
    if (q0 + lane < qend) my = nbr[(size_t)k * ld + q0 + lane];
    const unsigned long long valid = __ballot(my >= 0);
    if (valid == 0ull) continue;
    __syncthreads();
    {
      const int r = tid >> 4;          // 0..15
      const int c4 = (tid & 15) * 4;   // 0..60
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int row = r + 16 * h;
        const int idx = __shfl(my, row);
        float4 vx = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 vd = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx >= 0) {
          const float* sx = x + (size_t)idx * cin + c0 + c4;
          const float* sd = dy + (size_t)(q0 + row) * cout + n0 + c4;
          if (VEC) {
            if (c0 + c4 < cin) vx = *reinterpret_cast<const float4*>(sx);
            if (n0 + c4 < cout) vd = *reinterpret_cast<const float4*>(sd);
          } else {
            if (c0 + c4 + 0 < cin) vx.x = sx[0];
            if (c0 + c4 + 1 < cin) vx.y = sx[1];
            if (c0 + c4 + 2 < cin) vx.z = sx[2];
            if (c0 + c4 + 3 < cin) vx.w = sx[3];
            if (n0 + c4 + 0 < cout) vd.x = sd[0];
            if (n0 + c4 + 1 < cout) vd.y = sd[1];
            if (n0 + c4 + 2 < cout) vd.z = sd[2];
            if (n0 + c4 + 3 < cout) vd.w = sd[3];
          }
        }
        *reinterpret_cast<float4*>(&sX[row * WG_LD + c4]) = vx;
        *reinterpret_cast<float4*>(&sD[row * WG_LD + c4]) = vd;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < WG_TQ / 4; ++ks) {
      const int qq = ks * 4 + g;
      const float a = sX[qq * WG_LD + 16 * wave + m];  // A[m = c][kk = q]
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float b = sD[qq * WG_LD + t * 16 + m];   // B[kk = q][n]
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
      }
    }
  }
  float* out = part + ((size_t)s * K + k) * cin * cout;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = n0 + t * 16 + m;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = c0 + 16 * wave + g * 4 + r;
      if (c < cin && n < cout) out[(size_t)c * cout + n] = acc[t][r];
    }
  }
}

// Many partial blocks, few elements (the stem's weight gradient: 1024 partial blocks of 27 x 7 x 32 = 6048 floats): k_wgrad_reduce gives
// every element ONE thread that walks all S partials — 24 workgroups, 128 dependent load batches each, 47 us at the very end of the
// scene encoder's backward.  Here a workgroup owns 32 elements x 8 slices of the partial blocks (slice t: blocks t, t + 8, ...), the
// slices are folded through LDS in slice order: 189 workgroups, 16 batches each.  A different (fixed) summation order than
// k_wgrad_reduce's, used from 64 partial blocks on only — the offset-split slabs (S <= 9), whose order the BatchNorm slab passes
// reproduce, stay on k_wgrad_reduce.
__global__ __launch_bounds__(256) void k_wgrad_reduce_wide(const float* __restrict__ part, int S, size_t elems, float* __restrict__ dw) {
  __shared__ float sl[8][32];
  const int e = threadIdx.x & 31, t = threadIdx.x >> 5;
  const size_t i = (size_t)blockIdx.x * 32 + e;
  float s = 0.f;
  if (i < elems) {
    int j = t;
    for (; j + 56 < S; j += 64) {                 // eight of this slice's blocks per trip
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(j + 8 * u) * elems + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; j < S; j += 8) s += part[(size_t)j * elems + i];
  }
  sl[t][e] = s;
  __syncthreads();
  if (t == 0 && i < elems) {
    float r = sl[0][e];
#pragma unroll
    for (int u = 1; u < 8; ++u) r += sl[u][e];
    dw[i] = r;
  }
}

// out_bf: dw is a bf16 tensor (conv output of the executor's bf16 storage mode; the slabs are always fp32)
__global__ void k_wgrad_reduce(const float* __restrict__ part, int S, size_t elems,
                               float* __restrict__ dw, int accumulate, int out_bf) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elems) return;
  // loads in batches of 8 (memory-level parallelism: one load per add left the kernel latency-bound), adds in the
  // original order (bit-identical)
  float s = 0.f;
  int j = 0;
  for (; j + 8 <= S; j += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(j + u) * elems + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; j < S; ++j) s += part[(size_t)j * elems + i];
  if (out_bf) {
    unsigned short* o = reinterpret_cast<unsigned short*>(dw) + i;
    if (accumulate) s = __uint_as_float((unsigned)*o << 16) + s;
    *o = (unsigned short)(irx_pk_bf16(s, 0.f) & 0xffffu);
    return;
  }
  dw[i] = accumulate ? dw[i] + s : s;            // (old + sum), the order of the unfused `old.add_(sum)`
}

// ---------------------------------------------------------------------------- host side -----
#include <atomic>
#include <string.h>
namespace {
struct KnobDef { const char* name; const char* env; long def; };
const KnobDef kKnobs[IRX_KNOB_COUNT] = {
    {"spconv3", "IRX_SPCONV3", 1},               // bf16-input convs on the third-generation kernel (irx_spconv3.hip)
    {"updgrad", "IRX_UPDGRAD", 1},               // fp32 stride-2 data-gradient tiled by parent rows (k_updgrad)
    {"updgrad_min", "IRX_UPDGRAD_MIN", 40000},   // ... from this many parent rows on
    {"wgrad_v1", "IRX_WGRAD_V1", 0},             // fp32 pair-list weight-gradient on the first-generation kernel
    {"wgrad3", "IRX_WGRAD3", 1},                 // bf16-row pair-list weight-gradient on k_wgrad3 (transposing LDS reads)
    {"wgrad3_units", "IRX_WGRAD3_UNITS", 448},   // ... work units (workgroups) of its XCD-segment mapping
    {"wgrad3_xcd_min", "IRX_WGRAD3_XCD_MIN", 200000},   // ... used from this many table entries (n_out * K) on
    {"wgrad_xcd_f32", "IRX_WGRAD_XCD_F32", 0},   // fp32 pair-list weight-gradient on XCD-segment work units: their number, 0 = off
    {"enc_fold_slabs", "IRX_ENC_FOLD_SLABS", 1}, // encoder executor: offset-split slabs folded by the BatchNorm statistics pass
    {"enc_abl", "IRX_ENC_ABL", 0},               // dev, TIMING ONLY (results wrong): encoder backward without bit 0 = weight gradients, bit 1 = data gradients
    {"stem_mfma", "IRX_STEM_MFMA", 1},           // 7-channel stem forward as im2col + fp32 MFMA (k_stem_fwd_mfma); 0: vector-ALU kernel
    {"spconv3_xcd_min", "IRX_SPCONV3_XCD_MIN", 16},   // k_spconv3: XCD-contiguous tile ranges from this many row tiles on (0 = off)
    {"spconv4", "IRX_SPCONV4", 1},               // bf16-input convs with 64 / 128 input channels on k_spconv4 (LDS-DMA, row-shaped gathers)
};
std::atomic<long> g_knob_val[IRX_KNOB_COUNT];
std::atomic<int> g_knob_set[IRX_KNOB_COUNT];
}  // namespace
long irx_knob(int id) {
  if (id < 0 || id >= IRX_KNOB_COUNT) return 0;
  if (!g_knob_set[id].load(std::memory_order_acquire)) {
    const char* e = getenv(kKnobs[id].env);
    g_knob_val[id].store(e ? atol(e) : kKnobs[id].def, std::memory_order_relaxed);
    g_knob_set[id].store(1, std::memory_order_release);
  }
  return g_knob_val[id].load(std::memory_order_relaxed);
}
extern "C" int irx_debug_set_knob(const char* name, long value) {
  IRX_REQUIRE(name != nullptr, "irx_debug_set_knob: null name");
  for (int i = 0; i < IRX_KNOB_COUNT; ++i)
    if (!strcmp(name, kKnobs[i].name)) {
      g_knob_val[i].store(value, std::memory_order_relaxed);
      g_knob_set[i].store(1, std::memory_order_release);
      return IRX_OK;
    }
  IRX_REQUIRE(false, "irx_debug_set_knob: unknown knob '%s'", name);
}
extern "C" long irx_debug_get_knob(const char* name) {
  for (int i = 0; name && i < IRX_KNOB_COUNT; ++i)
    if (!strcmp(name, kKnobs[i].name)) return irx_knob(i);
  return -1;
}

static thread_local hipEvent_t g_br_start = nullptr, g_br_stop = nullptr;
extern "C" int irx_profile_next_kernel(void* ev_start, void* ev_stop) {
  g_br_start = (hipEvent_t)ev_start;
  g_br_stop = (hipEvent_t)ev_stop;
  return IRX_OK;
}
void irx_bracket_begin(hipStream_t st) {
  if (g_br_start) (void)hipEventRecord(g_br_start, st);
}
void irx_bracket_end(hipStream_t st) {
  if (g_br_stop) (void)hipEventRecord(g_br_stop, st);
  g_br_start = g_br_stop = nullptr;
}
// One bracket around a COMPOSITE operator (the wide stem = two dominant kernels): begin records the start event and
// parks the pair so that the inner launches do not consume it; end restores it and records the stop event.
struct IrxBracketSpan {
  hipEvent_t stop;
  hipStream_t st;
  explicit IrxBracketSpan(hipStream_t s) : stop(g_br_stop), st(s) {
    if (g_br_start) (void)hipEventRecord(g_br_start, st);
    g_br_start = g_br_stop = nullptr;
  }
  ~IrxBracketSpan() {
    if (stop) (void)hipEventRecord(stop, st);
  }
};

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

template <bool TRANS_W, bool VEC>
static void launch_fwd(int bn, dim3 grid, hipStream_t st, const float* x, const float* w,
                       const int32_t* nbr, int ld, int n_out, int K, int cin, int cout, int flip_k,
                       float* y) {
  irx_bracket_begin(st);
  if (bn == 128)
    k_spconv_fwd<128, TRANS_W, VEC><<<grid, 256, 0, st>>>(x, w, nbr, ld, n_out, K, cin, cout, flip_k, y);
  else if (bn == 64)
    k_spconv_fwd<64, TRANS_W, VEC><<<grid, 256, 0, st>>>(x, w, nbr, ld, n_out, K, cin, cout, flip_k, y);
  else
    k_spconv_fwd<32, TRANS_W, VEC><<<grid, 256, 0, st>>>(x, w, nbr, ld, n_out, K, cin, cout, flip_k, y);
  irx_bracket_end(st);
}

static size_t fwd_ws_weights(int K, int cin, int cout) { return ((size_t)K * cin * cout * sizeof(float) + 255) & ~(size_t)255; }

// ---- wide stem: the multiview input (reference scripts/train.py:74-75, lib/dataset.py:112-118: 128 ENet channels on
// top of xyz / rgb / height -> C0 = 135) through the 3^3 stem conv C0 -> 32. The row is split: channels [0, 128) run on
// the MFMA kernels of the 128-channel layers reading x with row stride C0 (rows are only 4-byte aligned: 16-byte global
// loads at dword alignment are legal on gfx950, tools/micro/unaligned_load.hip), channels [128, C0) on the small-Cin
// stem kernels; forward adds the two (accumulating epilogue), the weight gradient merges the two blocks.
#define WS_MAIN 128
bool irx_wide_stem(int K, int cin, int cout) { return K == 27 && cout == 32 && cin > WS_MAIN && cin <= WS_MAIN + 8; }

// wt[k][c][n] = w[k][c0 + c][n]
__global__ void k_slice_w(const float* __restrict__ w, int K, int cin, int c0, int ct, int cout, float* __restrict__ wt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * ct * cout) return;
  const int n = i % cout, c = (i / cout) % ct, k = i / (cout * ct);
  wt[i] = w[((size_t)k * cin + c0 + c) * cout + n];
}
// dw[k][c][n] = c < c0 ? a[k][c][n] : b[k][c - c0][n]
__global__ void k_merge_w(const float* __restrict__ a, const float* __restrict__ b, int K, int cin, int c0, int cout,
                          float* __restrict__ dw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * cin * cout) return;
  const int n = i % cout, c = (i / cout) % cin, k = i / (cout * cin);
  dw[i] = (c < c0) ? a[((size_t)k * c0 + c) * cout + n] : b[((size_t)k * (cin - c0) + (c - c0)) * cout + n];
}
static size_t wide_tail_bytes(int K, int cin, int cout) { return ((size_t)K * (cin - WS_MAIN) * cout * sizeof(float) + 255) & ~(size_t)255; }

extern "C" size_t irx_spconv_fwd_workspace_bytes(int n_out, int K, int cin, int cout, int trans_w) {
  // fast path: a fragment-major weight image (one coalesced 1 KiB load per wave fragment) + the partial-sum slabs
  // of the offset splits used for small layers
  if (n_out <= 0 || K <= 0 || cin <= 0 || cout <= 0) return 0;
  if (!trans_w && irx_wide_stem(K, cin, cout)) {
    const int splits = irx_spconv2_splits(n_out, K);
    return fwd_ws_weights(K, WS_MAIN, cout) + wide_tail_bytes(K, cin, cout) +
           (splits > 1 ? (size_t)splits * n_out * cout * sizeof(float) : 0);
  }
  if (!irx_spconv2_supported(cin, cout)) return 0;
  int splits = irx_spconv2_splits(n_out, K);
  if (irx_spconv3_supported(cin, cout) && irx_spconv3_enabled()) {   // a bf16 input takes the third-generation kernel
    const int s3 = irx_spconv3_splits(n_out, K);
    if (s3 > splits) splits = s3;
  }
  return fwd_ws_weights(K, cin, cout) + (splits > 1 ? (size_t)splits * n_out * cout * sizeof(float) : 0);
}

extern "C" int irx_spconv_fwd(const float* x, const float* w, const int32_t* nbr, int ld, int n_out,
                              int K, int cin, int cout, int flip_k, int trans_w, float* y,
                              void* workspace, size_t workspace_bytes, void* stream) {
  return irx_spconv_fwd_impl(x, w, nbr, ld, n_out, K, cin, cout, flip_k, trans_w, y, 0, nullptr, workspace,
                             workspace_bytes, stream);
}

extern "C" int irx_spconv_fwd_t(const void* x, const float* w, const int32_t* nbr, int ld, int n_in, int n_out, int K,
                                int cin, int cout, int flip_k, int trans_w, void* y, int accumulate, int x_bf, int y_bf,
                                void* workspace, size_t workspace_bytes, void* stream) {
  IRX_REQUIRE(!(x_bf || y_bf) || irx_conv_bf16(), "irx_spconv_fwd_t: bf16 tensors need irx_set_compute_dtype(1 | 2)");
  IrxStore ty;
  ty.x = x_bf ? 1 : 0;
  ty.y = y_bf ? 1 : 0;
  ty.x_rows = n_in;
  return irx_spconv_fwd_impl((const float*)x, w, nbr, ld, n_out, K, cin, cout, flip_k, trans_w, (float*)y, accumulate,
                             nullptr, workspace, workspace_bytes, stream, ty);
}

bool irx_spconv_fast_path(const void* x, const void* w, const void* y, int cin, int cout, int trans_w) {
  const bool aligned = (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0;
  return aligned && irx_spconv2_supported(cin, cout) && irx_spconv2_enabled(trans_w ? 'd' : 'f');
}

int irx_spconv_fwd_impl(const float* x, const float* w, const int32_t* nbr, int ld, int n_out, int K, int cin, int cout,
                        int flip_k, int trans_w, float* y, int accumulate, const float* wimg, void* workspace,
                        size_t workspace_bytes, void* stream, IrxStore ty) {
  IRX_REQUIRE(n_out >= 0 && K >= 1 && cin >= 1 && cout >= 1, "irx_spconv_fwd: bad sizes");
  if (n_out == 0) return IRX_OK;
  IRX_REQUIRE(x && w && nbr && y, "irx_spconv_fwd: null pointer");
  IRX_REQUIRE(ld >= n_out, "irx_spconv_fwd: ld %d < n_out %d", ld, n_out);
  const bool aligned = (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0;
  if (!accumulate && !trans_w && !flip_k && !ty.x && irx_stem_supported(K, cin, cout) && (((uintptr_t)y & 15) == 0))
    return irx_stem_fwd_launch(x, w, nbr, ld, n_out, K, cin, y, S(stream), 0, ty.y);
  if (aligned && irx_spconv2_supported(cin, cout) && irx_spconv2_enabled(trans_w ? 'd' : 'f')) {
    const size_t need = irx_spconv_fwd_workspace_bytes(n_out, K, cin, cout, trans_w);
    if (workspace == nullptr || workspace_bytes < need || ((uintptr_t)workspace & 15)) {
      irx_set_error("irx_spconv_fwd: workspace %zu < %zu", workspace_bytes, need);
      return IRX_ERR_WORKSPACE;
    }
    // a bf16 input of known size: third-generation kernel (irx_spconv3.hip) with its own weight image and offset splits
    const bool v3 = irx_spconv3_use(cin, cout, ty.x, ty.x_rows, 0);
    const int splits = v3 ? irx_spconv3_splits(n_out, K) : irx_spconv2_splits(n_out, K);
    float* slabs = (float*)((char*)workspace + fwd_ws_weights(K, cin, cout));
    int rc = IRX_OK;
    if (!wimg) {
      rc = v3 ? irx_permute_w3_launch(w, K, cin, cout, trans_w, (float*)workspace, S(stream))
              : irx_permute_w_launch(w, K, cin, cout, trans_w, (float*)workspace, S(stream));
      if (rc) return rc;
      wimg = (const float*)workspace;
    }
    if (v3)
      rc = irx_spconv3_launch(x, wimg, nbr, ld, n_out, K, cin, cout, flip_k, splits > 1 ? slabs : y, splits,
                              splits > 1 ? 0 : accumulate, S(stream), 0, splits > 1 ? 0 : ty.y);
    else
      rc = irx_spconv2_launch(x, wimg, nbr, ld, n_out, K, cin, cout, flip_k,
                              splits > 1 ? slabs : y, splits, accumulate, S(stream), 0, ty);
    if (rc) return rc;
    if (splits > 1) {
      if (ty.slabs_out && ty.splits_out) {      // the caller folds the slabs (into its BatchNorm pass), `accumulate` included
        *ty.slabs_out = slabs;
        *ty.splits_out = splits;
        return IRX_OK;
      }
      const size_t elems = (size_t)n_out * cout;
      k_wgrad_reduce<<<irx_cdiv((long long)elems, 256), 256, 0, S(stream)>>>(slabs, splits, elems, y, accumulate, ty.y);
      IRX_CHECK_LAUNCH("irx_spconv_fwd(split reduce)");
    }
    return IRX_OK;
  }
  IRX_REQUIRE(!accumulate, "irx_spconv_fwd: accumulation needs the fast path (aligned, channels in {32,64,128})");
  IRX_REQUIRE(!ty.x, "irx_spconv_fwd: a bf16 input needs the fast path (aligned, channels in {32,64,128})");
  if (!trans_w && !flip_k && irx_wide_stem(K, cin, cout) && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0) &&
      irx_spconv2_enabled('f')) {
    const size_t need = irx_spconv_fwd_workspace_bytes(n_out, K, cin, cout, 0);
    if (workspace == nullptr || workspace_bytes < need || ((uintptr_t)workspace & 15)) {
      irx_set_error("irx_spconv_fwd(wide stem): workspace %zu < %zu", workspace_bytes, need);
      return IRX_ERR_WORKSPACE;
    }
    const int ct = cin - WS_MAIN;
    IrxBracketSpan span(S(stream));
    float* wimg_main = (float*)workspace;
    float* wtail = (float*)((char*)workspace + fwd_ws_weights(K, WS_MAIN, cout));
    float* slabs = (float*)((char*)wtail + wide_tail_bytes(K, cin, cout));
    k_slice_w<<<irx_cdiv(K * ct * cout, 256), 256, 0, S(stream)>>>(w, K, cin, WS_MAIN, ct, cout, wtail);
    IRX_CHECK_LAUNCH("irx_spconv_fwd(wide stem slice)");
    int rc = irx_permute_w_launch(w, K, WS_MAIN, cout, 0, wimg_main, S(stream), cin);
    if (rc) return rc;
    rc = irx_stem_fwd_launch(x + WS_MAIN, wtail, nbr, ld, n_out, K, ct, y, S(stream), cin, ty.y);   // y  = tail channels
    if (rc) return rc;
    const int splits = irx_spconv2_splits(n_out, K);
    IrxStore tm;
    tm.y = ty.y;
    rc = irx_spconv2_launch(x, wimg_main, nbr, ld, n_out, K, WS_MAIN, cout, 0, splits > 1 ? slabs : y, splits, 1,
                            S(stream), cin, tm);                                                  // y += main channels
    if (rc) return rc;
    if (splits > 1) {
      const size_t elems = (size_t)n_out * cout;
      k_wgrad_reduce<<<irx_cdiv((long long)elems, 256), 256, 0, S(stream)>>>(slabs, splits, elems, y, 1, ty.y);
      IRX_CHECK_LAUNCH("irx_spconv_fwd(wide stem reduce)");
    }
    return IRX_OK;
  }
  IRX_REQUIRE(!ty.y, "irx_spconv_fwd: a bf16 output needs the fast paths (stem, wide stem, channels in {32,64,128})");
  const int bn = cout > 64 ? 128 : (cout > 32 ? 64 : 32);
  dim3 grid(irx_cdiv(n_out, SC_TM), irx_cdiv(cout, bn));
  const bool vec = (cin % 4 == 0) && (cout % 4 == 0) && (((uintptr_t)x & 15) == 0) &&
                   (((uintptr_t)w & 15) == 0);
  if (trans_w) {
    if (vec) launch_fwd<true, true>(bn, grid, S(stream), x, w, nbr, ld, n_out, K, cin, cout, flip_k, y);
    else launch_fwd<true, false>(bn, grid, S(stream), x, w, nbr, ld, n_out, K, cin, cout, flip_k, y);
  } else {
    if (vec) launch_fwd<false, true>(bn, grid, S(stream), x, w, nbr, ld, n_out, K, cin, cout, flip_k, y);
    else launch_fwd<false, false>(bn, grid, S(stream), x, w, nbr, ld, n_out, K, cin, cout, flip_k, y);
  }
  IRX_CHECK_LAUNCH("irx_spconv_fwd");
  return IRX_OK;
}

// number of row splits: enough workgroups to fill 256 CUs a few times over, >= 256 rows each
static int wgrad_splits(int n_out, int K, int cin, int cout) {
  const int tiles = irx_spconv2_supported(cin, cout) ? 1 : irx_cdiv(cin, 64) * irx_cdiv(cout, 64);
  int s = irx_cdiv(1536, (long long)K * tiles);
  const int max_s = irx_cdiv(n_out, 256);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return s;
}

extern "C" size_t irx_spconv_wgrad_workspace_bytes(int n_out, int K, int cin, int cout) {
  if (n_out <= 0 || K <= 0 || cin <= 0 || cout <= 0) return 0;
  if (irx_stem_supported(K, cin, cout)) return (size_t)irx_stem_wgrad_blocks(n_out) * K * cin * cout * sizeof(float);
  if (irx_wide_stem(K, cin, cout)) {
    // [main partial slabs | main sum | tail partial slabs | tail sum]; the main partials are the offset-major kernel's
    // row-split slabs or, with pair lists, k_wgrad_pairs' share slabs — sized for the larger of the two
    const int s = wgrad_splits(n_out, K, WS_MAIN, cout);
    size_t main_part = (size_t)s * K * WS_MAIN * cout * sizeof(float);
    const size_t pairs_part = irx_wgrad_pairs_wide_workspace_bytes(n_out, K, cout);
    if (pairs_part > main_part) main_part = pairs_part;
    return main_part + ((size_t)K * WS_MAIN * cout + (size_t)(irx_stem_wgrad_blocks(n_out) + 1) * K * (cin - WS_MAIN) * cout) *
           sizeof(float);
  }
  const int s = wgrad_splits(n_out, K, cin, cout);
  return s <= 1 ? 0 : (size_t)s * K * cin * cout * sizeof(float);
}

extern "C" int irx_spconv_wgrad(const float* x, const float* dy, const int32_t* nbr, int ld,
                                int n_out, int K, int cin, int cout, float* dw, void* workspace,
                                size_t workspace_bytes, void* stream) {
  return irx_spconv_wgrad_impl(x, dy, nbr, ld, n_out, K, cin, cout, dw, workspace, workspace_bytes, stream, 0);
}

// dy_bf != 0 (executor, bf16 storage mode): dy is a bf16 tensor; x stays fp32 (this entry only serves the stems there)
int irx_spconv_wgrad_impl(const float* x, const float* dy, const int32_t* nbr, int ld, int n_out, int K, int cin, int cout,
                          float* dw, void* workspace, size_t workspace_bytes, void* stream, int dy_bf, IrxPairLists pairs) {
  IRX_REQUIRE(n_out >= 0 && K >= 1 && cin >= 1 && cout >= 1 && dw, "irx_spconv_wgrad: bad arguments");
  const size_t elems = (size_t)K * cin * cout;
  if (n_out == 0) {
    IRX_CHECK_HIP(hipMemsetAsync(dw, 0, elems * sizeof(float), S(stream)), "irx_spconv_wgrad(memset)");
    return IRX_OK;
  }
  IRX_REQUIRE(x && dy && nbr, "irx_spconv_wgrad: null pointer");
  IRX_REQUIRE(ld >= n_out, "irx_spconv_wgrad: ld %d < n_out %d", ld, n_out);
  const int s = wgrad_splits(n_out, K, cin, cout);
  const size_t need = irx_spconv_wgrad_workspace_bytes(n_out, K, cin, cout);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need)) {
    irx_set_error("irx_spconv_wgrad: workspace %zu < %zu", workspace_bytes, need);
    return IRX_ERR_WORKSPACE;
  }
  if (irx_stem_supported(K, cin, cout)) {
    const int blocks = irx_stem_wgrad_blocks(n_out);
    int rc = irx_stem_wgrad_launch(x, dy, nbr, ld, n_out, cin, blocks, (float*)workspace, S(stream), 0, dy_bf);
    if (rc) return rc;
    if (blocks >= 64)
      k_wgrad_reduce_wide<<<irx_cdiv((long long)elems, 32), 256, 0, S(stream)>>>((const float*)workspace, blocks, elems, dw);
    else
      k_wgrad_reduce<<<irx_cdiv((long long)elems, 256), 256, 0, S(stream)>>>((const float*)workspace, blocks, elems, dw);
    IRX_CHECK_LAUNCH("irx_spconv_wgrad(stem reduce)");
    return IRX_OK;
  }
  if (irx_wide_stem(K, cin, cout) && ((((uintptr_t)x | (uintptr_t)dy) & 15) == 0) && irx_spconv2_enabled('w')) {
    const int ct = cin - WS_MAIN;
    IrxBracketSpan span(S(stream));
    const int sm = wgrad_splits(n_out, K, WS_MAIN, cout);
    const size_t em = (size_t)K * WS_MAIN * cout, et = (size_t)K * ct * cout;
    float* part_m = (float*)workspace;
    size_t main_part = (size_t)sm * em * sizeof(float);
    if (irx_wgrad_pairs_wide_workspace_bytes(n_out, K, cout) > main_part) main_part = irx_wgrad_pairs_wide_workspace_bytes(n_out, K, cout);
    float* sum_m = (float*)((char*)workspace + main_part);
    const int blocks = irx_stem_wgrad_blocks(n_out);
    float* part_t = sum_m + em;
    float* sum_t = part_t + (size_t)blocks * et;
    int rc;
    if (pairs.in_list) {
      // rulebook form: dense 64-pair stages, bf16 MFMA in the bf16 modes (the offset-major kernel below is fp32 MFMA only
      // and walks the table with ~36 % of its rows valid: 1313 -> ... us per launch on the 200 k-point stress scenes)
      rc = irx_wgrad_pairs_wide_launch(x, cin, dy, pairs.in_list, pairs.out_list, pairs.ldp, pairs.counts, n_out, K, cout,
                                       part_m, sum_m, S(stream), dy_bf);
      if (rc) return rc;
    } else {
      int rpm = irx_cdiv(n_out, sm);
      rpm = irx_cdiv(rpm, WG_TQ) * WG_TQ;
      rc = irx_spconv2_wgrad_launch(x, dy, nbr, ld, n_out, K, WS_MAIN, cout, sm, rpm, part_m, S(stream), cin, dy_bf);
      if (rc) return rc;
      k_wgrad_reduce<<<irx_cdiv((long long)em, 256), 256, 0, S(stream)>>>(part_m, sm, em, sum_m);
      IRX_CHECK_LAUNCH("irx_spconv_wgrad(wide stem reduce)");
    }
    rc = irx_stem_wgrad_launch(x + WS_MAIN, dy, nbr, ld, n_out, ct, blocks, part_t, S(stream), cin, dy_bf);
    if (rc) return rc;
    if (blocks >= 64) k_wgrad_reduce_wide<<<irx_cdiv((long long)et, 32), 256, 0, S(stream)>>>(part_t, blocks, et, sum_t);
    else k_wgrad_reduce<<<irx_cdiv((long long)et, 256), 256, 0, S(stream)>>>(part_t, blocks, et, sum_t);
    IRX_CHECK_LAUNCH("irx_spconv_wgrad(wide stem tail reduce)");
    k_merge_w<<<irx_cdiv((long long)elems, 256), 256, 0, S(stream)>>>(sum_m, sum_t, K, cin, WS_MAIN, cout, dw);
    IRX_CHECK_LAUNCH("irx_spconv_wgrad(wide stem merge)");
    return IRX_OK;
  }
  IRX_REQUIRE(!dy_bf, "irx_spconv_wgrad: a bf16 gradient needs the stem / wide-stem paths");
  int rps = irx_cdiv(n_out, s);
  rps = irx_cdiv(rps, WG_TQ) * WG_TQ;
  const int nct_n = irx_cdiv(cout, 64);
  dim3 grid(s, K, irx_cdiv(cin, 64) * nct_n);
  float* part = (s > 1) ? (float*)workspace : dw;
  const bool vec = (cin % 4 == 0) && (cout % 4 == 0) && (((uintptr_t)x & 15) == 0) &&
                   (((uintptr_t)dy & 15) == 0);
  if (vec && irx_spconv2_supported(cin, cout) && irx_spconv2_enabled('w')) {
    int rc = irx_spconv2_wgrad_launch(x, dy, nbr, ld, n_out, K, cin, cout, s, rps, part, S(stream));
    if (rc) return rc;
  } else {
    irx_bracket_begin(S(stream));
    if (vec)
      k_spconv_wgrad<true><<<grid, 256, 0, S(stream)>>>(x, dy, nbr, ld, n_out, K, cin, cout, rps, nct_n, part);
    else
      k_spconv_wgrad<false><<<grid, 256, 0, S(stream)>>>(x, dy, nbr, ld, n_out, K, cin, cout, rps, nct_n, part);
    irx_bracket_end(S(stream));
  }
  IRX_CHECK_LAUNCH("irx_spconv_wgrad");
  if (s > 1) {
    k_wgrad_reduce<<<irx_cdiv((long long)elems, 256), 256, 0, S(stream)>>>(part, s, elems, dw);
    IRX_CHECK_LAUNCH("irx_spconv_wgrad(reduce)");
  }
  return IRX_OK;
}
