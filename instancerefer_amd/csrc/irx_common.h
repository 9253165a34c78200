// irx_common.h — shared device/host helpers for libirx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/irx.h"

#define IRX_WAVE 64
#define IRX_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define IRX_COORD_BIAS 32768
#define IRX_POISON_KEY 0xFFFFFFFFFFFFFFFEull   // k_quantize: a point outside the 16-bit voxel range / 15-bit batch range

// ---- error plumbing (thread-local message; functions never throw) -----------------------
void irx_set_error(const char* fmt, ...);

#define IRX_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      irx_set_error(__VA_ARGS__);              \
      return IRX_ERR_INVALID_ARG;              \
    }                                          \
  } while (0)

#define IRX_CHECK_LAUNCH(name)                                                  \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      irx_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
      return IRX_ERR_LAUNCH;                                                    \
    }                                                                           \
  } while (0)

#define IRX_CHECK_HIP(expr, name)                                               \
  do {                                                                          \
    hipError_t e__ = (expr);                                                    \
    if (e__ != hipSuccess) {                                                    \
      irx_set_error("%s: %s", name, hipGetErrorString(e__));                    \
      return IRX_ERR_LAUNCH;                                                    \
    }                                                                           \
  } while (0)

static inline int irx_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- Morton (Z-order) keys ---------------------------------------------------------------
// 16-bit coordinate -> bits spread to every third position (48-bit interleave).
__host__ __device__ static inline uint64_t irx_spread3(uint32_t v) {
  uint64_t x = v & 0xFFFFu;
  x = (x | (x << 32)) & 0x00FF00000000FFFFull;  // not needed for 16 bits but keeps pattern
  x = (x | (x << 16)) & 0x00FF0000FF0000FFull;
  x = (x | (x << 8)) & 0xF00F00F00F00F00Full;
  x = (x | (x << 4)) & 0x30C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x9249249249249249ull;
  return x;
}

// key = batch<<48 | interleave(x,y,z) with x in bit 0 (x fastest), biased to unsigned.
__host__ __device__ static inline uint64_t irx_make_key(int x, int y, int z, int b) {
  uint32_t ux = (uint32_t)(x + IRX_COORD_BIAS), uy = (uint32_t)(y + IRX_COORD_BIAS),
           uz = (uint32_t)(z + IRX_COORD_BIAS);
  uint64_t m = irx_spread3(ux) | (irx_spread3(uy) << 1) | (irx_spread3(uz) << 2);
  return ((uint64_t)(uint32_t)b << 48) | (m & 0xFFFFFFFFFFFFull);
}

// inverse of irx_spread3: every third bit of a 48-bit interleave -> 16-bit value
__host__ __device__ static inline uint32_t irx_compact3(uint64_t x) {
  x &= 0x9249249249249249ull;
  x = (x | (x >> 2)) & 0x30C30C30C30C30C3ull;
  x = (x | (x >> 4)) & 0xF00F00F00F00F00Full;
  x = (x | (x >> 8)) & 0x00FF0000FF0000FFull;
  x = (x | (x >> 16)) & 0x00FF00000000FFFFull;
  x = (x | (x >> 32)) & 0xFFFFull;
  return (uint32_t)x;
}

// ---- open-addressing hash (linear probing, 64-bit keys) ---------------------------------
__host__ __device__ static inline uint64_t irx_mix64(uint64_t k) {
  // murmur3 finaliser
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}

__device__ static inline int irx_hash_lookup(const uint64_t* __restrict__ tk,
                                             const int32_t* __restrict__ tv, uint64_t mask,
                                             uint64_t key) {
  uint64_t slot = irx_mix64(key) & mask;
  while (true) {
    uint64_t cur = tk[slot];
    if (cur == key) return tv[slot];
    if (cur == IRX_EMPTY_KEY) return -1;
    slot = (slot + 1) & mask;
  }
}

// ---- bf16 operand helpers of the MFMA conv kernels (irx_set_compute_dtype(1)) ---------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
// two fp32 -> one dword of two bf16 (lo = a, hi = b), round-to-nearest-even
// (compiler-native convert, NOT inline asm: the hazard recognizer has to see the VALU write to pad the wait states a
// following MFMA operand read needs — an asm v_cvt_pk_bf16_f32 fed the matrix core stale registers)
__device__ __forceinline__ unsigned irx_pk_bf16(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(((f32x2){a, b}), bf16x2));
}
__device__ __forceinline__ s16x4 irx_frag_bf16(unsigned lo, unsigned hi) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(s16x4, ((u32x2){lo, hi}));
}


// ---- bf16 STORAGE (irx_set_compute_dtype(2)): activations / gradients inside the encoder executor are bf16 in HBM ------
// 4 consecutive elements at ELEMENT offset `off` (a multiple of 4) of a tensor that is float32 (bf == 0) or bf16 (bf != 0;
// the pointer is then really an unsigned short*). The flag is wave-uniform, so the branch is free in an HBM-bound kernel.
__device__ __forceinline__ float4 irx_bf4_to_f4(uint2 u) {
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ uint2 irx_f4_to_bf4(float4 v) { return make_uint2(irx_pk_bf16(v.x, v.y), irx_pk_bf16(v.z, v.w)); }
__device__ __forceinline__ float4 irx_ld4(const float* p, size_t off, int bf) {
  if (bf) return irx_bf4_to_f4(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p) + off));
  return *reinterpret_cast<const float4*>(p + off);
}
__device__ __forceinline__ void irx_st4(float* p, size_t off, int bf, float4 v) {
  if (bf) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p) + off) = irx_f4_to_bf4(v);
  else *reinterpret_cast<float4*>(p + off) = v;
}
static inline size_t irx_esz(int bf) { return bf ? 2 : 4; }
// element types of one conv call inside the executor (0 = float32, 1 = bf16); the C-ABI entry points use all zeros
struct IrxStore {
  int x = 0, y = 0;
  const int32_t* order = nullptr;   // launch order of the 64-row output tiles (irx_tile_order) or NULL = blockIdx order
  long long x_rows = 0;             // rows of x when the caller knows them (the third-generation bf16 kernel addresses rows with
                                    // 32-bit byte offsets and is only taken for a known size below 2 GiB); 0 = unknown
  // the caller folds the offset-split slabs itself (encoder executor: into the BatchNorm statistics pass, irx_bn_forward_slabs_t):
  // when both are set and the launch used S > 1 fp32 slabs [S][n_out][cout], *slabs_out / *splits_out receive them, y is NOT
  // written and the reduce launch is skipped (the slabs live in the conv workspace: consume them before the next conv call)
  float** slabs_out = nullptr;
  int* splits_out = nullptr;
};

// ---- dev / test knobs (irx_debug_set_knob, include/irx.h): each is read from its environment variable ONCE, on first use, and
// can afterwards only be changed through the setter (an atomic store) — a per-call getenv() from library lane threads raced
// with tests that mutate os.environ from the Python thread.
enum IrxKnob { IRX_KNOB_SPCONV3 = 0, IRX_KNOB_UPDGRAD, IRX_KNOB_UPDGRAD_MIN, IRX_KNOB_WGRAD_V1, IRX_KNOB_WGRAD3, IRX_KNOB_WGRAD3_UNITS, IRX_KNOB_WGRAD3_XCD_MIN, IRX_KNOB_WGRAD_XCD_F32, IRX_KNOB_FOLD_SLABS, IRX_KNOB_ABL, IRX_KNOB_STEM_MFMA, IRX_KNOB_SPCONV3_XCD_MIN, IRX_KNOB_SPCONV4, IRX_KNOB_COUNT };
long irx_knob(int id);

// ---- measurement aid (irx_profile_next_kernel, include/irx.h): brackets the next DOMINANT sparse-conv kernel of this
// host thread (k_spconv2 / k_wgrad_pairs / k_spconv2_wgrad / k_stem_*) with two caller-owned events, excluding the small helper
// launches (weight permute, split reduce) that share the C-ABI call.
void irx_bracket_begin(hipStream_t st);
void irx_bracket_end(hipStream_t st);

// ---- second-generation sparse-conv launchers (irx_spconv2.hip) ---------------------------------
// pins the compute mode of the calling thread's launches for its lifetime (irx_spconv2.hip); mode < 0 = no override
struct IrxModeScope {
  int prev;
  explicit IrxModeScope(int mode);
  ~IrxModeScope();
};
bool irx_conv_bf16();          // irx_set_compute_dtype(1 | 2): bf16 operands / fp32 accumulation in the MFMA conv kernels
bool irx_conv_bf16_storage();  // irx_set_compute_dtype(2): ... and bf16 activations / gradients inside the encoder executor
bool irx_spconv2_supported(int cin, int cout);
bool irx_spconv2_enabled(char pass);
int irx_spconv2_splits(int n_out, int K);
// ldx (last argument, 0 = cin): row stride of x in floats when x is the leading `cin` columns of wider rows
int irx_spconv2_launch(const float* x, const float* wn, const int32_t* nbr, int ld, int n_out, int K, int cin,
                       int cout, int flip_k, float* y, int splits, int accumulate, hipStream_t st, int ldx = 0,
                       IrxStore ty = IrxStore());
// irx_spconv_fwd with gradient accumulation (accumulate != 0: y += result; fast-path channel counts only) and an
// optional prebuilt fragment-major weight image (wimg != NULL: the per-call permute launch is skipped)
int irx_spconv_fwd_impl(const float* x, const float* w, const int32_t* nbr, int ld, int n_out, int K, int cin, int cout,
                        int flip_k, int trans_w, float* y, int accumulate, const float* wimg, void* workspace,
                        size_t workspace_bytes, void* stream, IrxStore ty = IrxStore());
// pair lists (optional; in_list != NULL): the wide stem's 128 leading channels then go through k_wgrad_pairs
struct IrxPairLists {
  const int32_t *in_list = nullptr, *out_list = nullptr, *counts = nullptr;
  int ldp = 0;
};
int irx_spconv_wgrad_impl(const float* x, const float* dy, const int32_t* nbr, int ld, int n_out, int K, int cin, int cout,
                          float* dw, void* workspace, size_t workspace_bytes, void* stream, int dy_bf,
                          IrxPairLists pairs = IrxPairLists());
size_t irx_wgrad_pairs_wide_workspace_bytes(int n_out, int K, int cout);
int irx_wgrad_pairs_wide_launch(const float* x, int ldx, const float* dy, const int32_t* in_list, const int32_t* out_list,
                                int ldp, const int32_t* counts, int n_out, int K, int cout, float* part, float* sum,
                                hipStream_t st, int dy_bf);
bool irx_wide_stem(int K, int cin, int cout);
int irx_permute_w_launch(const float* w, int K, int cin, int cout, int trans_w, float* wf, hipStream_t st, int src_cin = 0);
// fragment images of up to 16 layers in one launch; dims are kernel-relative (cin = reduction, cout = outputs),
// end4[j] = running total of float4 elements (K*cin*cout/4) up to and including job j
struct IrxPermuteJobs {
  const float* w[16];
  float* dst[16];
  int K[16], cin[16], cout[16];
  size_t end4[16];
  int n;
};
int irx_permute_w_multi_launch(const IrxPermuteJobs& jobs, int trans_w, hipStream_t st);
bool irx_spconv_fast_path(const void* x, const void* w, const void* y, int cin, int cout, int trans_w);
int irx_spconv2_wgrad_launch(const float* x, const float* dy, const int32_t* nbr, int ld, int n_out, int K,
                             int cin, int cout, int splits, int rps, float* part, hipStream_t st, int ldx = 0,
                             int dy_bf = 0);

// ---- third-generation forward / data-gradient for bf16 inputs (irx_spconv3.hip) ------------------------------------------
bool irx_spconv3_supported(int cin, int cout);
bool irx_spconv3_enabled();
bool irx_spconv3_use(int cin, int cout, int x_bf, long long x_rows, int ldx);
int irx_spconv3_splits(int n_out, int K);
int irx_permute_w3_launch(const float* w, int K, int cin, int cout, int trans_w, float* dst, hipStream_t st);
int irx_permute_w3_multi_launch(const IrxPermuteJobs& jobs, int trans_w, hipStream_t st);
int irx_spconv3_launch(const float* x, const float* wimg, const int32_t* nbr, int ld, int n_out, int K, int cin, int cout,
                       int flip_k, float* y, int splits, int accumulate, hipStream_t st, int ldx, int y_bf);

// data-gradient of a stride-2 convolution tiled by parent rows (irx_spconv2.hip, k_updgrad; fp32 only)
bool irx_updgrad_supported(int cr, int co);
int irx_updgrad_launch(const float* dy, const float* wn, const int32_t* child, int ldc, int n_parent, int cr, int co,
                       float* dx, hipStream_t st);

int irx_spconv_wgrad_pairs_impl(const float* x, const float* dy, const int32_t* in_list, const int32_t* out_list, int ldp,
                                const int32_t* counts, int n_out, int K, int cin, int cout, float* dw, void* workspace,
                                size_t workspace_bytes, void* stream, int bf_rows, int n_in = 0);
// (n_in: rows of x when the caller knows them — the fp32 second-generation kernel addresses x and dy with 32-bit byte
//  offsets and is only taken when both tensors are known to stay below 2 GiB; 0 = unknown -> first-generation kernel)

// ---- BatchNorm with per-tensor element types (irx_norm.hip; 0 = float32, 1 = bf16) — the executor's bf16 storage mode
int irx_bn_stats_t(const float* x, int n, int c, float eps, float momentum, float* mean, float* invstd,
                   float* running_mean, float* running_var, void* workspace, size_t workspace_bytes, void* stream,
                   int x_bf);
int irx_bn_apply_t(const float* x, int n, int c, const float* mean, const float* invstd, const float* gamma,
                   const float* beta, const float* residual, int relu, float* y, void* stream, int x_bf, int res_bf,
                   int y_bf);
int irx_bn_forward_t(const float* x, int n, int c, float eps, float momentum, const float* gamma, const float* beta,
                     const float* residual, int relu, float* mean, float* invstd, float* running_mean, float* running_var,
                     float* y, void* workspace, size_t workspace_bytes, void* stream, int x_bf, int res_bf, int y_bf);
int irx_bn_forward_slabs_t(const float* slabs, int S, int n, int c, float eps, float momentum, const float* gamma, const float* beta,
                           const float* residual, int relu, float* mean, float* invstd, float* running_mean, float* running_var,
                           float* x_out, float* y, void* workspace, size_t workspace_bytes, void* stream, int x_bf, int res_bf,
                           int y_bf);
// the incoming gradient still as the S offset-split fp32 slabs of the data-gradient convolution that produces it (IrxStore::
// slabs_out): irx_bn_backward_t folds them (dy = (acc ? dy : 0) + sum of the slabs, stored to dy) in its statistics pass
struct IrxDySlabs {
  const float* slabs = nullptr;
  int S = 0, acc = 0;
};
int irx_bn_backward_t(const float* x, const float* y, const float* dy, int n, int c, const float* mean,
                      const float* invstd, const float* gamma, int relu, float* dx, float* dgamma, float* dbeta,
                      float* dresidual, void* workspace, size_t workspace_bytes, void* stream, int x_bf, int y_bf,
                      int dy_bf, int dx_bf, int dres_bf, int phases = 3, const float* all_sum_g = nullptr,
                      const float* all_sum_gx = nullptr, double all_count = 0.0, const double* count_dev = nullptr,
                      const float* beta = nullptr, IrxDySlabs dys = IrxDySlabs());

int irx_bn_sums_t(const float* x, int n, int c, double* sums, void* workspace, size_t workspace_bytes, void* stream, int x_bf,
                  double* count_slot);
int irx_bn_pack_sums(const float* sum_g, const float* sum_gx, int c, float* out, void* stream);

// ---- stem (small-Cin) launchers (irx_stem.hip) ------------------------------------------------------
bool irx_stem_supported(int K, int cin, int cout);
int irx_stem_fwd_launch(const float* x, const float* w, const int32_t* nbr, int ld, int n_out, int K, int cin,
                        float* y, hipStream_t st, int ldx = 0, int y_bf = 0);
int irx_stem_wgrad_blocks(int n_out);
int irx_stem_wgrad_launch(const float* x, const float* dy, const int32_t* nbr, int ld, int n_out, int cin,
                          int blocks, float* part, hipStream_t st, int ldx = 0, int dy_bf = 0);
// multiview stem: 3^3 conv, 129..136 input channels -> 32 (irx_spconv.hip "wide stem")
bool irx_wide_stem(int K, int cin, int cout);
