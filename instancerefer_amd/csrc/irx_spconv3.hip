// irx_spconv3.hip — third-generation sparse-conv forward / data-gradient for the bf16 STORAGE mode (irx_set_compute_dtype(2):
// x is a bf16 tensor in HBM, BASELINE configs[2]-[4]).  Reference semantics: models/basic_blocks.py:10-95 (spnn.Conv3d).
//
// k_spconv2 in bf16 is the fp32 design with narrower operands: per (64-row tile, offset) it compacts pairs, stages the
// gathered rows in LDS, runs 16-pair MFMA groups and read-modify-writes an fp32 LDS output tile — round 3's ablation
// priced that skeleton at 52 of 131 us on the largest 128-channel level with 4 us of matrix-core time.  At 2.5 PFLOP/s the
// matrix core is the one resource this path has to spare, so this kernel spends it to delete the skeleton:
//   * workgroup = NW waves, wave w owns 32 consecutive (Morton-ordered) output rows x ALL output channels; its fp32
//     accumulators (COUT/32 x 16 VGPRs) stay in registers for the whole offset loop: no LDS output tile, no
//     read-modify-write, no pair compaction, no pair lists;
//   * A operand: lane l gathers 16 bytes of row nbr[k][row l & 31] straight into the A-fragment layout of
//     v_mfma_f32_32x32x16_bf16 (8 reduction channels per lane and step) with raw BUFFER loads — a missing neighbour is an
//     out-of-range offset and returns zeros without a memory access, the wasted MFMA rows are free;  a wave whose 32 rows
//     have no neighbour at an offset skips that offset's MFMAs altogether (wave-uniform bit test);
//   * B operand: W[k] as a bf16 image in B-fragment order ([step][column block][lane][8]: one ds_read_b128 per fragment,
//     conflict-free, shared by all waves), DOUBLE-buffered in LDS: offset k+1's image is requested at the top of offset k's
//     MFMA chain and written behind it; one barrier per offset;
//   * epilogue: the wave transposes its accumulators through LDS once per tile and stores whole rows (16 B per lane);
//   * small levels split the offsets over blockIdx.y (fp32 slabs summed by k_wgrad_reduce), as k_spconv2 does.
#include <stdlib.h>
#include <type_traits>
#include "irx_common.h"

typedef __bf16 s3_bf16x8 __attribute__((ext_vector_type(8)));
typedef float s3_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned s3_u32x4 __attribute__((ext_vector_type(4)));

// dev ablation (timing only, results wrong): 1 = no A gather (offsets all out of range), 2 = no MFMA, 4 = no W staging,
// 8 = no B-fragment LDS reads, 16 = no W instructions at all (loads and LDS stores), 32 = no row-load instructions,
// 64 = no per-offset barrier
#ifndef IRX_S3_ABL
#define IRX_S3_ABL 0
#endif
// channel -> (step s, lane half h, element j) of the A / B fragments: 0: c = 16 s + 8 h + j (the two halves of a row read
// adjacent 16-byte pieces), 1: c = (CIN/2) h + 8 s + j (each lane reads its own contiguous CIN bytes over the steps)
#ifndef IRX_S3_AMAP
#define IRX_S3_AMAP 0
#endif

template <int CIN>
__host__ __device__ constexpr int s3_chan(int s, int h, int j) {
  return IRX_S3_AMAP ? (CIN / 2) * h + 8 * s + j : 16 * s + 8 * h + j;
}

// dev: per-wave s_memtime stamps (tools/micro/s3_trace.py): 256 slots per wave, buffer set by irx_debug_s3_trace
#ifndef IRX_S3_TRACE
#define IRX_S3_TRACE 0
#endif
#if IRX_S3_TRACE
__device__ unsigned long long* g_s3_trace = nullptr;
extern "C" int irx_debug_s3_trace(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_s3_trace), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define S3_EV(i_) do { if (trc && lane == 0) trc[i_] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define S3_EV(i_) do { } while (0)
#endif

#define S3_OOB 0x80000000u
#ifndef IRX_S3_SKIP
#define IRX_S3_SKIP 1
#endif

// s_waitcnt vmcnt(n) with expcnt / lgkmcnt untouched (gfx9 encoding: vmcnt = simm16[3:0] | simm16[15:14] << 4)
template <int N>
__device__ __forceinline__ void s3_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

// KH: an offset's reduction dimension is processed in KH items of CIN / KH channels (KH = 2 for 128 x 128: a 16 KB half image
// per item -> 46 KB of LDS and, with half the row / image registers, <= 168 VGPRs: THREE resident workgroups per CU)
template <int CIN, int COUT, int NW, int KH>
__global__ __launch_bounds__(64 * NW, (NW <= 4 ? (KH > 1 ? 3 : 2) : 1))
void k_spconv3(const unsigned short* __restrict__ x, const uint4* __restrict__ wimg, const int32_t* __restrict__ nbr,
               int ld, int n_out, int K, int flip_k, float* __restrict__ y, int k_per_split, int accumulate, int ldx,
               int y_bf, int xcd_tiles, int splits) {
  constexpr int TM = 32 * NW, NTH = 64 * NW;
  constexpr int NS = CIN / 16 / KH, NCB = COUT / 32;    // MFMA steps of one item
  constexpr int WPK = CIN * COUT * 2 / 16;          // 16-byte pieces of one offset's image
  constexpr int WP = WPK / KH;                      // ... of one item's part of it (steps are the image's major index)
  constexpr int WPT = (WP + NTH - 1) / NTH;         // ... per thread (the last pass may cover only the leading waves)
  constexpr bool WFULL = (WP % NTH) == 0;
  static_assert(WP % 64 == 0 && (CIN / 16) % KH == 0, "a staging pass is whole waves");
  constexpr int KMAX = 27;
  constexpr int LDO = 32 + 4;                       // epilogue: one 32-column block per pass
  constexpr int SM_MAIN = 2 * WP * 16 + KMAX * TM * 4;
  constexpr int SM_EPI = NW * 32 * LDO * 4;
  constexpr int SM = SM_MAIN > SM_EPI ? SM_MAIN : SM_EPI;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM];
  __shared__ unsigned sMask;
  s3_u32x4* sW = reinterpret_cast<s3_u32x4*>(smem);                          // [2][WP]
  unsigned* sOff = reinterpret_cast<unsigned*>(smem + 2 * WP * 16);    // [nk][TM] byte offsets into x, S3_OOB = none

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Work unit -> (row tile, offset split).  xcd_tiles = 0: grid (tiles, splits).  xcd_tiles = T > 0: 1-D grid of 8 * T * splits
  // workgroups; the dispatcher deals consecutive workgroups round-robin over the 8 XCDs (workgroup b runs on XCD b % 8), so
  // XCD x is given the T CONSECUTIVE tiles x T .. x T + T - 1: rows are in Morton order within a scene, hence the rows an XCD's
  // tiles gather — their own range plus a halo — are an eighth of the tensor and stay in that XCD's 4 MB L2, instead of every
  // L2 fetching the whole tensor once (measured fabric traffic of the round-robin form: ~8x the tensor per launch).
  int tile = blockIdx.x, split = blockIdx.y;
  if (xcd_tiles) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = xcd * xcd_tiles + j / splits;
    split = j % splits;
    if (tile * TM >= n_out) return;                 // (block-uniform: the last XCD's range may be short)
  }
#if IRX_S3_TRACE
  unsigned long long* trc = g_s3_trace ? g_s3_trace + ((size_t)(blockIdx.x + blockIdx.y * gridDim.x) * NW + wave) * 256 : nullptr;
  int itn = 0;
  if (trc && lane == 0) { trc[255] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); trc[254] = ((unsigned long long)tile << 32) | (unsigned)split; }
#endif
  S3_EV(0);
  const int q0 = tile * TM;
  const int kb = split * k_per_split;
  const int ke = (kb + k_per_split < K) ? kb + k_per_split : K;
  const int nk = ke - kb;
  y += (size_t)split * n_out * COUT;           // offset-split slabs (fp32; y_bf and accumulate are 0 then)
  const unsigned rowbytes = (unsigned)ldx * 2u;

  // ---- setup: the tile's table columns as byte offsets; per-wave and per-tile activity masks ----
  // (all of a thread's table reads are issued before the first one is used: as a plain load-then-store loop they went out one at a
  // time — 17 k cycles of dependent round trips on the 81 k-row level, s_memtime stamps of tools/micro/s3_trace.py)
  {
    constexpr int NLD = (KMAX * TM + NTH - 1) / NTH;
    int idxv[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int e = tid + j * NTH;
      const int kk = e / TM, r = e % TM;
      const int k = kb + kk;
      const int kt = flip_k ? (K - 1 - k) : k;
      idxv[j] = (e < nk * TM && q0 + r < n_out) ? nbr[(size_t)kt * ld + q0 + r] : -1;
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int e = tid + j * NTH;
      if (e < nk * TM) sOff[e] = (idxv[j] >= 0 && !(IRX_S3_ABL & 1)) ? (unsigned)idxv[j] * rowbytes : S3_OOB;
    }
  }
  if (tid == 0) sMask = 0;
  __syncthreads();
  S3_EV(1);
  unsigned wm = 0;                                   // offsets at which this wave's 32 rows have a neighbour
  for (int kk = 0; kk < nk; ++kk) {
    const unsigned o = sOff[kk * TM + wave * 32 + (lane & 31)];
    if (__ballot(o != S3_OOB) != 0ull) wm |= 1u << kk;
  }
  if (IRX_S3_ABL & 1) wm = (1u << nk) - 1u;
  wm = __builtin_amdgcn_readfirstlane(wm);
  if (lane == 0 && wm) atomicOr(&sMask, wm);
  __syncthreads();
  S3_EV(2);
  unsigned act = __builtin_amdgcn_readfirstlane(sMask);

  s3_f32x16 acc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;

  if (act) {
    // Every global read of the offset loop is a raw BUFFER load issued unconditionally: a missing neighbour, a wave without
    // pairs at the next offset and "no next offset" are all an out-of-range offset (zeros, no memory access).  With no load
    // behind a branch the number of loads in flight is exact at every point, so the two explicit s_waitcnt below are all
    // the vector-memory waits of the loop (a conditional prefetch made the compiler wait for the loads it had just issued).
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)wimg, 0, (unsigned)K * (unsigned)(WPK * 16), 0x00020000);
    const unsigned ahalf = IRX_S3_AMAP ? (unsigned)(lane >> 5) * (unsigned)CIN : (unsigned)(lane >> 5) * 16u;
    constexpr unsigned ASTEP = IRX_S3_AMAP ? 16u : 32u;
    const unsigned* myoff = sOff + wave * 32 + (lane & 31);
    const unsigned wlane = (unsigned)tid * 16u;
    s3_u32x4 A[2][NS];
    s3_u32x4 wreg[WPT];
    // an item = (offset kk, part h of its reduction channels); kk < 0: nothing (requests forced out of range)
    struct Cur { int kk, h; };
    auto pop = [&]() __attribute__((always_inline)) {
      int k = -1;
      if (act) { k = __builtin_ctz(act); act &= act - 1; }
      return k;
    };
    auto next = [&](Cur c) __attribute__((always_inline)) {
      Cur n;
      if (KH > 1 && c.kk >= 0 && c.h + 1 < KH) { n.kk = c.kk; n.h = c.h + 1; }
      else { n.kk = pop(); n.h = 0; }
      return n;
    };
    // piece i_ of item c_'s image -> wreg[i_]
#define S3_LOAD_W(c_, i_)                                                                                             \
    do {                                                                                                                \
      const unsigned voff_ =                                                                                            \
          wlane | (((c_).kk < 0 || (IRX_S3_ABL & 4) || (!WFULL && tid + (i_) * NTH >= WP)) ? S3_OOB : 0u);              \
      const unsigned soff_ = ((unsigned)(kb + ((c_).kk < 0 ? 0 : (c_).kk)) * (unsigned)KH + (unsigned)(c_).h) * (unsigned)(WP * 16); \
      if (!(IRX_S3_ABL & 16))                                                                                           \
        wreg[i_] = __builtin_bit_cast(s3_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, voff_, soff_ + (i_) * NTH * 16, 0)); \
    } while (0)
#define S3_STORE_W(b_, i_)                                                                                            \
    do {                                                                                                                \
      if (!(IRX_S3_ABL & 16) && (WFULL || tid + (i_) * NTH < WP)) sW[(b_) * WP + tid + (i_) * NTH] = wreg[i_];          \
    } while (0)
#define S3_OFF_A(c_) ((myoff[((c_).kk < 0 ? 0 : (c_).kk) * TM] + ahalf + (unsigned)(c_).h * (NS * ASTEP)) | ((c_).kk < 0 ? S3_OOB : 0u))
#define S3_BARRIER()                                                                                                  \
    do {                                                                                                                \
      if (!(IRX_S3_ABL & 64)) __syncthreads();                                                                          \
    } while (0)
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if (IRX_S3_ABL & (16 | 32)) {
#pragma unroll
      for (int i = 0; i < WPT; ++i) wreg[i] = (s3_u32x4){1u, 2u, 3u, 4u};
#pragma unroll
      for (int s = 0; s < NS; ++s) A[0][s] = A[1][s] = (s3_u32x4){(unsigned)lane, 2u, 3u, 4u};
    }

    // Software pipeline over the tile's items u_0, u_1, ...; item i has parity C = i & 1, T = C ^ 1:
    //   on entry  sW[C] = image of u_i (visible: barrier), A[C] = rows for u_i and wreg = image of u_i+1 in flight (requested
    //             during item i - 1);
    //   step s    waits for exactly A[C][s] and wreg[s], writes wreg[s] to sW[T] (last read in item i - 1, behind a barrier),
    //             requests the same piece of u_i+2's image into the register it has just stored (an in-place ring: one
    //             register set) and A[T][s] = its piece of the rows for u_i+1 — every request has one whole item to land —
    //             then runs the step's MFMAs over sW[C];  one barrier ends the item.
    const int n_off = __builtin_popcount(act);
    Cur cur = next(Cur{-1, 0});
    Cur nx = next(cur);
#pragma unroll
    for (int i = 0; i < WPT; ++i) S3_LOAD_W(cur, i);
    {
      const unsigned off0 = S3_OFF_A(cur);
      if (!(IRX_S3_ABL & 32)) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
          A[0][s] = __builtin_bit_cast(s3_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off0 + s * ASTEP, 0, 0));
      }
    }
    s3_wait_vmcnt<(IRX_S3_ABL & 32) ? 0 : NS>();
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      S3_STORE_W(0, i);
      S3_LOAD_W(nx, i);
    }
    s3_wait_vmcnt<0>();                               // once per tile: the per-step waits below count from a clean slate
    S3_BARRIER();
    S3_EV(3);

    auto item = [&](auto C_, auto T_, Cur c, Cur cn, Cur cn2) __attribute__((always_inline)) {
      constexpr int C = decltype(C_)::value, T = decltype(T_)::value;
      // The NS .. NS + WPT vector-memory instructions of the item are spread over the MFMA chain, one or two per step: a wave
      // that issues them back to back sits in the texture-address queue (~25 cycles per 1 KiB instruction and CU) with its
      // MFMA pipe idle.  IRX_S3_SKIP: a wave whose 32 rows have no neighbour at this offset skips the MFMAs (per step,
      // wave-uniform; the requests stay unconditional so that the counts below are exact).
      S3_EV(4 + 4 * itn);
      const unsigned offA = S3_OFF_A(cn);
      const bool live = c.kk >= 0 && (!IRX_S3_SKIP || ((wm >> c.kk) & 1u));
      const s3_u32x4* bw = sW + C * WP + lane;
      // B fragments of step s + 1 are requested before the MFMAs of step s
      s3_u32x4 bq[2][NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) bq[0][cb] = (IRX_S3_ABL & 8) ? A[C][cb % NS] : bw[cb * 64];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const s3_bf16x8 a = __builtin_bit_cast(s3_bf16x8, A[C][s]);
        if (s + 1 < NS) {
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb)
            bq[(s + 1) & 1][cb] = (IRX_S3_ABL & 8) ? A[C][(s + 1 + cb) % NS] : bw[((s + 1) * NCB + cb) * 64];
        }
        // exact wait: A[C][s] and wreg[s] were requested at step s of the previous item; the loads issued since then — the rest
        // of that item and steps 0 .. s-1 of this one — stay in flight (every request has one whole item to land)
        if (!(IRX_S3_ABL & 48)) {
          if (s < WPT) s3_wait_vmcnt<NS + WPT - 2>();
          else s3_wait_vmcnt<NS + WPT - 1>();
        }
        if (s == 0) S3_EV(5 + 4 * itn);
        if (s < WPT) {
          S3_STORE_W(T, s);
          S3_LOAD_W(cn2, s);
        }
        if (!(IRX_S3_ABL & 32))
          A[T][s] = __builtin_bit_cast(s3_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, offA + s * ASTEP, 0, 0));
        __builtin_amdgcn_sched_barrier(0);            // (keeps this step's memory instructions between the MFMA blocks)
        if (live) {
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) {
            if constexpr ((IRX_S3_ABL & 2) != 0) {
              acc[cb][0] += __uint_as_float(bq[s & 1][cb][0] ^ A[C][s][0]);
            } else {
              acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(s3_bf16x8, bq[s & 1][cb]), acc[cb], 0, 0, 0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      S3_EV(6 + 4 * itn);
      S3_BARRIER();
      S3_EV(7 + 4 * itn);
#if IRX_S3_TRACE
      ++itn;
#endif
    };
    // Items go in pairs (register-set parity is a compile-time constant) and the loop has ONE exit, at the bottom: an odd number
    // of items ends with an empty one (kk < 0: no MFMAs, all requests out of range).  A mid-loop exit made the compiler
    // route a never-taken edge from the first item back to the loop header, and its waitcnt pass then waited at the header
    // for the loads the first item had just issued.
    const int npairs = (n_off * KH + 1) >> 1;
    for (int ip = 0; ip < npairs; ++ip) {
      Cur nx2 = next(nx);
      item(I0{}, I1{}, cur, nx, nx2);
      cur = nx; nx = nx2;
      nx2 = next(nx);
      item(I1{}, I0{}, cur, nx, nx2);
      cur = nx; nx = nx2;
    }
    s3_wait_vmcnt<0>();
    if (IRX_S3_ABL & (16 | 32)) {
#pragma unroll
      for (int i = 0; i < WPT; ++i) acc[0][i] += __uint_as_float(wreg[i][0]);
    }
  }

  // ---- epilogue: D layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  One 32-column block at a time through
  // the wave's OWN LDS region (no workgroup barrier: the loop's last barrier is behind every read of sW / sOff), then whole
  // 32-column row pieces go out with 16 B (fp32) / 8 B (bf16) per lane ----
  S3_EV(250);
#if IRX_S3_TRACE
  if (trc && lane == 0) trc[253] = (unsigned long long)itn;
#endif
  float* so = reinterpret_cast<float*>(smem) + wave * 32 * LDO;
  const int col = lane & 31, r4 = 4 * (lane >> 5);
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) so[((reg & 3) + 8 * (reg >> 2) + r4) * LDO + col] = acc[cb][reg];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int f = lane + 64 * it;                  // 32 rows x 8 float4
      const int row = f >> 3, cc = (f & 7) * 4;
      const int gr = q0 + wave * 32 + row;
      if (gr < n_out) {
        float4 o = *reinterpret_cast<const float4*>(&so[row * LDO + cc]);
        const size_t off = (size_t)gr * COUT + cb * 32 + cc;
        if (accumulate) {
          const float4 e = irx_ld4(y, off, y_bf);
          o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
        }
        irx_st4(y, off, y_bf, o);
      }
    }
  }
  S3_EV(251);
}

// LDS-DMA issue as inline asm.  With the builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) hipcc treats every later ds_read as
// possibly aliasing the pending DMA and puts s_waitcnt vmcnt(0) in front of it — the request is then waited for where it is
// issued.  As asm the requests are invisible to the compiler's counter model (its own waits can only over-wait: vmcnt retires in
// order), and the kernel's explicit vmcnt(0) at the item barrier is the only wait they get.  M0 (the LDS destination) is saved and
// restored around a group.  `pre_lgkm0`: s_waitcnt lgkmcnt(0) first (the fragment reads of the buffer being refilled are back).
typedef int s4_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s4_i32x4 s4_rsrc(const void* base, unsigned num_records) {
  const unsigned long long b = (unsigned long long)base;
  return (s4_i32x4){(int)(unsigned)b, (int)((unsigned)(b >> 32) & 0xffffu), (int)num_records, 0x00020000};
}
__device__ __forceinline__ unsigned s4_lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}
// The accumulate-in-place form of the MFMA as asm: with the builtin inside k_spconv4's conditional (wave-skip) item body the
// register allocator gave every accumulator a second home (D != C, 64 more VGPRs -> spills at three workgroups per CU).  The
// operands come from ds_read (the compiler's lgkmcnt covers asm inputs); dependent MFMAs on one accumulator are interlocked by
// the hardware; the accumulators are next read behind the tile's last barrier.
__device__ __forceinline__ void s4_mfma(s3_f32x16& acc, const s3_u32x4& a, const s3_u32x4& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// one instruction: LDS address lds, per-lane byte offset v, scalar offset soff
__device__ __forceinline__ void s4_dma1(s4_i32x4 rs, unsigned lds, unsigned v, unsigned soff) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds), "v"(v), "s"(rs), "s"(soff)
      : "memory");
}
// the row instructions of one set, as many as it has groups of 8 present rows (np rows: ceil(np / 8) of the four); the count is
// tested INSIDE the asm: branches around the requests in the C++ made the register allocator split the accumulators' live ranges
__device__ __forceinline__ void s4_dma_rows_n(s4_i32x4 rs, unsigned lds0, unsigned v0, unsigned v1, unsigned v2, unsigned v3,
                                              unsigned soff, int np) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_cmp_lt_i32 %8, 1\n\t"
      "s_cbranch_scc1 s4_rows_done_%=\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %2, %6, %7 offen lds\n\t"
      "s_cmp_lt_i32 %8, 9\n\t"
      "s_cbranch_scc1 s4_rows_done_%=\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %3, %6, %7 offen lds\n\t"
      "s_cmp_lt_i32 %8, 17\n\t"
      "s_cbranch_scc1 s4_rows_done_%=\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %4, %6, %7 offen lds\n\t"
      "s_cmp_lt_i32 %8, 25\n\t"
      "s_cbranch_scc1 s4_rows_done_%=\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %5, %6, %7 offen lds\n"
      "s4_rows_done_%=:\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds0), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(rs), "s"(soff), "s"(np)
      : "memory", "scc");
}
// four row instructions: LDS lds0 + {0, 1, 2, 3} KB, per-lane byte offsets v0..v3, one scalar offset
__device__ __forceinline__ void s4_dma_rows(s4_i32x4 rs, unsigned lds0, unsigned v0, unsigned v1, unsigned v2, unsigned v3,
                                            unsigned soff) {
  unsigned keep;
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %2, %6, %7 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %3, %6, %7 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %4, %6, %7 offen lds\n\t"
      "s_add_u32 m0, m0, 0x400\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %5, %6, %7 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds0), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(rs), "s"(soff)
      : "memory", "scc");
}
// N image instructions: LDS lds0 + i * 4 KB, scalar offsets soff0 + i * 4 KB, per-lane offset v
template <int N>
__device__ __forceinline__ void s4_dma_img(s4_i32x4 rs, unsigned lds0, unsigned v, unsigned soff0) {
  unsigned keep, so;
  if constexpr (N == 1) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %3, %4, %5 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(so)
        : "s"(lds0), "v"(v), "s"(rs), "s"(soff0)
        : "memory", "scc");
  } else if constexpr (N == 2) {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_add_u32 %1, %5, 0x1000\n\t"
        "buffer_load_dwordx4 %3, %4, %5 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %3, %4, %1 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(so)
        : "s"(lds0), "v"(v), "s"(rs), "s"(soff0)
        : "memory", "scc");
  } else {
    static_assert(N == 4, "1, 2 or 4 image instructions per wave");
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_add_u32 %1, %5, 0x1000\n\t"
        "buffer_load_dwordx4 %3, %4, %5 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %3, %4, %1 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_add_u32 %1, %1, 0x1000\n\t"
        "buffer_load_dwordx4 %3, %4, %1 offen lds\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_add_u32 %1, %1, 0x1000\n\t"
        "buffer_load_dwordx4 %3, %4, %1 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(so)
        : "s"(lds0), "v"(v), "s"(rs), "s"(soff0)
        : "memory", "scc");
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_spconv4 — the same operator with every global read of the offset loop as an LDS-DMA (buffer_load ... lds) and the gathered
// rows fetched ROW-shaped.  What the probe tools/micro/ta/ta_rate.hip measured on gfx950 (cycles of the CU's vector-memory path
// per 64-lane x 16-byte instruction, operands L2-resident): 1 KB contiguous 21; four whole 256-byte rows ~25; the A-FRAGMENT
// shape k_spconv3 gathers with (32 rows x 32 bytes: 32 cache lines per instruction) 75, 35 with 62 % of the rows missing — and
// k_spconv3's item time on the 81 k-row level (2 650 cycles per half offset, s_memtime stamps of tools/micro/s3_trace.py) is
// exactly its 48 row + 48 image instructions per CU at those prices: the kernel is bound by the instruction rate of the texture
// path, not by bytes.  So here
//   * an item = (offset, 64 reduction channels): a wave's 32 rows x 128 bytes arrive as FOUR instructions of 8 rows x one
//     128-byte line each (8 lanes per row), straight into the wave's own 4 KB of LDS (no VGPRs, no ds_write); a missing
//     neighbour is an out-of-range offset: zeros are written, no memory access;  the 16-byte pieces of a row are permuted on the
//     SOURCE side (piece p of row r lands at position p ^ ((r >> 1) & 7): the DMA destination is lane-linear), which makes the
//     A-fragment ds_read_b128 of the MFMA steps conflict-free;
//   * the item's part of the W image (B-fragment order, as k_spconv3) arrives the same way, NCB instructions per wave, into a
//     double buffer shared by the workgroup;
//   * A fragments go LDS -> registers at the top of an item, so the wave's row buffer is free for the next item's DMA during the
//     MFMA chain: one row buffer per wave, 48 KB of LDS per workgroup at 128 x 128 -> three resident workgroups per CU;
//   * the tile's table columns stay in REGISTERS (14 per lane: lane l holds row l & 31 of offsets 2 j + (l >> 5)); an offset's
//     row addresses are a select chain + four ds_bpermute: no table in LDS, no dependent global read in the loop;
//   * everything an item requests is waited for (vmcnt(0)) at its closing barrier: no counted waits, one barrier per item.
// NW waves per workgroup, RS sets of 32 rows per wave: a B fragment read from LDS feeds RS MFMAs (at RS = 1 the B-fragment
// reads alone are half of the CU's LDS bandwidth and, with the rows now going through LDS too, LDS is what bounds the item).
// dev ablation of k_spconv4 (timing only, results wrong): 1 = no row requests, 2 = no MFMA, 4 = no image requests, 8 = no B-fragment
// reads, 16 = no item barrier / wait, 32 = no A-fragment reads
#ifndef IRX_S4_ABL
#define IRX_S4_ABL 0
#endif
template <int CIN, int COUT, int NW, int RS>
__global__ __launch_bounds__(64 * NW, (RS == 1 ? 3 : (NW == 2 ? 3 : 2)))
void k_spconv4(const unsigned short* __restrict__ x, const uint4* __restrict__ wimg, const int32_t* __restrict__ nbr,
               int ld, int n_out, int K, int flip_k, float* __restrict__ y, int k_per_split, int accumulate, int ldx,
               int y_bf, int xcd_tiles, int splits) {
  constexpr int WR = 32 * RS, TM = WR * NW;         // rows per wave / per workgroup
  constexpr int KH = CIN / 64, NS = 4, NCB = COUT / 32;
  constexpr int WPK = CIN * COUT * 2 / 16;          // 16-byte pieces of one offset's image
  constexpr int WP = WPK / KH;                      // ... of one item's part
  constexpr int WI = WP / 64;                       // 1 KB DMA instructions per item image (= NS * NCB)
  static_assert(CIN % 64 == 0 && WI % NW == 0, "items are 64 reduction channels; image instructions split evenly over the waves");
  constexpr int WIW = WI / NW;
  constexpr int NA = 4 * RS;                        // row instructions per wave and item
  constexpr int LDO = 32 + 4;
  constexpr int SA = 33 * 128;                      // a row set in LDS: 32 row slots + one row of zeros (absent neighbours read it)
  constexpr int SM_MAIN = 2 * WP * 16 + NW * RS * SA;
  constexpr int SM_EPI = NW * 32 * LDO * 4;
  constexpr int SM = SM_MAIN > SM_EPI ? SM_MAIN : SM_EPI;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SM];
  __shared__ unsigned sMask;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile = blockIdx.x, split = blockIdx.y;
  if (xcd_tiles) {                                   // XCD-contiguous tile ranges (k_spconv3's work-unit comment)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = xcd * xcd_tiles + j / splits;
    split = j % splits;
    if (tile * TM >= n_out) return;
  }
#if IRX_S3_TRACE
  unsigned long long* trc = g_s3_trace ? g_s3_trace + ((size_t)(blockIdx.x + blockIdx.y * gridDim.x) * NW + wave) * 256 : nullptr;
  int itn = 0;
  if (trc && lane == 0) { trc[255] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); trc[254] = ((unsigned long long)tile << 32) | (unsigned)split; }
#endif
  S3_EV(0);
  const int q0 = tile * TM;
  const int kb = split * k_per_split;
  const int ke = (kb + k_per_split < K) ? kb + k_per_split : K;
  const int nk = ke - kb;
  y += (size_t)split * n_out * COUT;
  const unsigned rowbytes = (unsigned)ldx * 2u;

  // ---- the wave's rows of every table column of this split: lanes 0-31 hold the even offsets, 32-63 the odd ones ----
  const int r32 = lane & 31, hsel = lane >> 5;
  int idxr[RS][14];
#pragma unroll
  for (int t = 0; t < RS; ++t) {
    const int grow = q0 + wave * WR + 32 * t + r32;
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      const int kk = 2 * j + hsel;
      int v = -1;
      if (kk < nk && grow < n_out) {
        const int k = kb + kk;
        v = nbr[(size_t)(flip_k ? (K - 1 - k) : k) * ld + grow];
      }
      idxr[t][j] = v;
    }
  }
  if (tid == 0) sMask = 0;
  __syncthreads();
  unsigned wm = 0;                                   // offsets at which this wave's rows have a neighbour
#pragma unroll
  for (int t = 0; t < RS; ++t)
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      const unsigned long long b = __ballot(idxr[t][j] >= 0);
      if ((unsigned)b) wm |= 1u << (2 * j);
      if ((unsigned)(b >> 32)) wm |= 2u << (2 * j);
    }
  wm = __builtin_amdgcn_readfirstlane(wm);
  if (lane == 0 && wm) atomicOr(&sMask, wm);
  __syncthreads();
  S3_EV(1);
  unsigned act = __builtin_amdgcn_readfirstlane(sMask);

  s3_f32x16 acc[RS][NCB];
#pragma unroll
  for (int t = 0; t < RS; ++t)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][cb][i] = 0.f;

  if (act) {
    const s4_i32x4 rs_x = s4_rsrc(x, 0x7FFFFFF0u);
    const s4_i32x4 rs_w = s4_rsrc(wimg, (unsigned)K * (unsigned)(WPK * 16));
    unsigned char* sW = smem;                                        // [2][WP * 16]
    unsigned char* sAw = smem + 2 * WP * 16 + wave * (RS * SA);      // this wave's rows: [RS][33][128 B], pieces permuted
    // DMA instruction jj of a row set covers rows 8 jj + (lane >> 3); lane & 7 = position of the piece in the LDS row
    const unsigned pc0 = (unsigned)((lane & 7) ^ ((lane >> 4) & 7)) * 16u;            // jj even: (r >> 1) & 7 = lane >> 4
    const unsigned pc1 = (unsigned)((lane & 7) ^ (((lane >> 4) + 4) & 7)) * 16u;      // jj odd
    // The rows of an offset are COMPACTED: the j-th present row of a set goes to row slot j, so that an offset with p present rows
    // costs ceil(p / 8) row instructions (an instruction whose lanes are all out of range costs ~1 cycle, one with a single
    // valid lane the full ~21-25).  A fragment of step s, row r32: piece 2 s + hsel of slot rank(r32) (or of the zero row,
    // slot 32), at position (2 s + hsel) ^ ((slot >> 1) & 7).
    const unsigned sAwa = s4_lds_addr(sAw);
    if (lane < 32) {
#pragma unroll
      for (int t = 0; t < RS; ++t) *reinterpret_cast<unsigned*>(sAw + t * SA + 4096 + lane * 4) = 0u;
    }
    const unsigned wlane = (unsigned)lane * 16u;
    const unsigned ldsA = __builtin_amdgcn_readfirstlane(s4_lds_addr(sAw));
    const unsigned ldsW = __builtin_amdgcn_readfirstlane(s4_lds_addr(sW) + (unsigned)wave * 1024u);

    auto pop = [&]() __attribute__((always_inline)) {
      int k = -1;
      if (act) { k = __builtin_ctz(act); act &= act - 1; }
      return k;
    };
    // row byte offsets of offset kk for the row instructions (S3_OOB: nothing) and the fragment read addresses of its rows
    auto rowoffs = [&](int kk, unsigned (&ro)[NA], unsigned (&fa)[RS][NS], int (&np)[RS]) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < RS; ++t) {
        int v = -1;
#pragma unroll
        for (int j = 0; j < 14; ++j) v = ((kk >> 1) == j) ? idxr[t][j] : v;
        const bool mine = hsel == (kk & 1);
        const unsigned long long bal = __ballot(mine && v >= 0);
        const unsigned m32 = (kk & 1) ? (unsigned)(bal >> 32) : (unsigned)bal;      // present rows of the set (wave-uniform)
        const int npres = __builtin_popcount(m32);
        np[t] = npres;
        const int rank = __builtin_popcount(m32 & ((1u << r32) - 1u));
        const bool pres = (m32 >> r32) & 1u;
        // row index of the j-th present row -> lane j (lanes that have nothing to send target the upper half: never read)
        const int comp = __builtin_amdgcn_ds_permute(((mine && pres) ? rank : 32 + r32) * 4, v);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int slot = 8 * jj + (lane >> 3);
          const int id = __shfl(comp, slot, 64);
          ro[4 * t + jj] = slot < npres ? (unsigned)id * rowbytes + ((jj & 1) ? pc1 : pc0) : S3_OOB;
        }
        const int slot = pres ? rank : 32;
#pragma unroll
        for (int s = 0; s < NS; ++s)
          fa[t][s] = sAwa + (unsigned)(t * SA + slot * 128 + (((2 * s + hsel) ^ ((slot >> 1) & 7)) * 16));
      }
    };
    typedef __attribute__((address_space(3))) const s3_u32x4* lds_frag_ptr;

    int kcur = pop(), knext = pop();
    unsigned roC[NA], roN[NA], faC[RS][NS], faN[RS][NS];
    int npC[RS], npN[RS];
    rowoffs(kcur, roC, faC, npC);
#pragma unroll
    for (int t = 0; t < RS; ++t) npN[t] = 0;
#pragma unroll
    for (int jj = 0; jj < NA; ++jj) roN[jj] = S3_OOB;
#pragma unroll
    for (int t = 0; t < RS; ++t)
#pragma unroll
      for (int s = 0; s < NS; ++s) faN[t][s] = faC[t][s];
    if (knext >= 0) rowoffs(knext, roN, faN, npN);
    // row instruction jj of set t is issued only if the set has more than 8 jj present rows: an LDS-DMA whose lanes are all out
    // of range still moves 1 KB of zeros over the CU's 64 B/clk return path
    auto issue_rows = [&](const unsigned (&ro)[NA], const int (&np)[RS], unsigned soff) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < RS; ++t)
        if (!(IRX_S4_ABL & 1)) s4_dma_rows_n(rs_x, ldsA + (unsigned)(t * SA), ro[4 * t], ro[4 * t + 1], ro[4 * t + 2], ro[4 * t + 3], soff,
                      __builtin_amdgcn_readfirstlane(np[t]));
    };
    {
      const unsigned wbase = __builtin_amdgcn_readfirstlane((unsigned)(kb + kcur) * (unsigned)(KH * WP * 16) + (unsigned)wave * 1024u);
#pragma unroll
      for (int i = 0; i < WIW; ++i) s4_dma1(rs_w, ldsW + (unsigned)i * (NW * 1024u), wlane, wbase + (unsigned)i * (NW * 1024u));
      issue_rows(roC, npC, 0u);
    }
    s3_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    S3_EV(3);
    int par = 0;
    // One loop iteration = one offset = KH items.  Whether this wave has rows at the offset is decided ONCE per iteration and the
    // two forms of the item body are straight-line code (a request that is not wanted gets an out-of-range offset: ~1 cycle).
    while (kcur >= 0) {
      // (an empty asm that "modifies" the accumulators at the head of every iteration: without it the allocator keeps them in one
      // register set at loop level and in another inside the branch below — 64 copies per item, and spills)
#pragma unroll
      for (int t = 0; t < RS; ++t)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) asm volatile("" : "+v"(acc[t][cb]));
      const bool liveC = (wm >> kcur) & 1u;
      const unsigned oobN = knext >= 0 ? 0u : S3_OOB;
      if (liveC) {
#pragma unroll
        for (int h = 0; h < KH; ++h) {
          S3_EV(4 + 4 * itn);
          // next item: (kcur, h + 1) or (knext, 0); nothing after the last one
          const bool last = h + 1 == KH;
          const unsigned wbase = __builtin_amdgcn_readfirstlane(
              (((unsigned)(kb + (last ? (knext < 0 ? 0 : knext) : kcur)) * (unsigned)KH + (unsigned)(last ? 0 : h + 1)) *
               (unsigned)(WP * 16)) + (unsigned)wave * 1024u);
          const unsigned wlds = ldsW + (unsigned)(par ^ 1) * (unsigned)(WP * 16);
          const unsigned wv = wlane | (last ? oobN : 0u);
          const unsigned asoff = last ? 0u : (unsigned)(h + 1) * 128u;
          s3_u32x4 a[RS][NS];
#pragma unroll
          for (int t = 0; t < RS; ++t)
#pragma unroll
            for (int s = 0; s < NS; ++s) a[t][s] = (IRX_S4_ABL & 32) ? (s3_u32x4){faC[t][s], 1u, 2u, 3u} : *(lds_frag_ptr)(unsigned long long)faC[t][s];
          const s3_u32x4* bw = reinterpret_cast<const s3_u32x4*>(sW + par * (WP * 16)) + lane;
          // MFMA groups of GC column blocks: group g = (step g / NG, blocks GC (g % NG) ..); the B fragments of group g + 1 are
          // requested before the MFMAs of group g (a ring of 2 x GC fragments), and the item's NA + WIW requests go out a few per
          // group, in the shadow of the MFMA chain
          constexpr int GC = NCB >= 2 ? 2 : 1, NG = NCB / GC, NGT = NS * NG;
#ifndef IRX_S4_BD
#define IRX_S4_BD 1
#endif
          constexpr int BD = (RS == 1) ? IRX_S4_BD : 1;   // groups of B fragments requested ahead of their MFMAs (LDS latency >> one group)
          s3_u32x4 bq[BD + 1][GC];
#pragma unroll
          for (int g0 = 0; g0 < BD && g0 < NGT; ++g0)
#pragma unroll
            for (int c = 0; c < GC; ++c) bq[g0][c] = (IRX_S4_ABL & 8) ? (s3_u32x4){(unsigned)lane, 1u, 2u, 3u} : bw[((g0 / NG) * NCB + (g0 % NG) * GC + c) * 64];
          // the rows are in registers before their buffer is handed to the next item's DMA
          __builtin_amdgcn_s_waitcnt(0xC07F);         // lgkmcnt(0)
          __builtin_amdgcn_sched_barrier(0);
          if (last) issue_rows(roN, npN, 0u); else issue_rows(roC, npC, asoff);
          __builtin_amdgcn_sched_barrier(0);
          S3_EV(5 + 4 * itn);
#pragma unroll
          for (int g = 0; g < NGT; ++g) {
            const int sg = g / NG, cg = (g % NG) * GC;
            if (g + BD < NGT) {
              const int sn = (g + BD) / NG, cn = ((g + BD) % NG) * GC;
#pragma unroll
              for (int c = 0; c < GC; ++c) bq[(g + BD) % (BD + 1)][c] = (IRX_S4_ABL & 8) ? (s3_u32x4){(unsigned)lane, 1u, 2u, (unsigned)g} : bw[(sn * NCB + cn + c) * 64];
            }
            {
              // the image instructions go out one per group, in the shadow of the MFMA chain
              if (g < WIW && !(IRX_S4_ABL & 4)) s4_dma1(rs_w, wlds + (unsigned)g * (NW * 1024u), wv, wbase + (unsigned)g * (NW * 1024u));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < GC; ++c)
#pragma unroll
              for (int t = 0; t < RS; ++t) { if (!(IRX_S4_ABL & 2)) s4_mfma(acc[t][cg + c], a[t][sg], bq[g % (BD + 1)][c]); else acc[t][cg + c][0] += __uint_as_float(a[t][sg][0] ^ bq[g % (BD + 1)][c][0]); }
            __builtin_amdgcn_sched_barrier(0);
          }
          S3_EV(6 + 4 * itn);
          if (!(IRX_S4_ABL & 16)) { s3_wait_vmcnt<0>(); __builtin_amdgcn_s_barrier(); }
          S3_EV(7 + 4 * itn);
#if IRX_S3_TRACE
          ++itn;
#endif
          par ^= 1;
        }
      } else {
#pragma unroll
        for (int h = 0; h < KH; ++h) {
          S3_EV(4 + 4 * itn);
          const bool last = h + 1 == KH;
          const unsigned wbase = __builtin_amdgcn_readfirstlane(
              (((unsigned)(kb + (last ? (knext < 0 ? 0 : knext) : kcur)) * (unsigned)KH + (unsigned)(last ? 0 : h + 1)) *
               (unsigned)(WP * 16)) + (unsigned)wave * 1024u);
          const unsigned wlds = ldsW + (unsigned)(par ^ 1) * (unsigned)(WP * 16);
          const unsigned wv = wlane | (last ? oobN : 0u);
          S3_EV(5 + 4 * itn);
#pragma unroll
          for (int i = 0; i < WIW; ++i) if (!(IRX_S4_ABL & 4)) s4_dma1(rs_w, wlds + (unsigned)i * (NW * 1024u), wv, wbase + (unsigned)i * (NW * 1024u));
          if (last) issue_rows(roN, npN, 0u);         // (compile-time) the rows of the next offset; this offset has none
          S3_EV(6 + 4 * itn);
          if (!(IRX_S4_ABL & 16)) { s3_wait_vmcnt<0>(); __builtin_amdgcn_s_barrier(); }
          S3_EV(7 + 4 * itn);
#if IRX_S3_TRACE
          ++itn;
#endif
          par ^= 1;
        }
      }
      kcur = knext;
#pragma unroll
      for (int jj = 0; jj < NA; ++jj) { roC[jj] = roN[jj]; roN[jj] = S3_OOB; }
#pragma unroll
      for (int t = 0; t < RS; ++t)
#pragma unroll
        for (int s = 0; s < NS; ++s) faC[t][s] = faN[t][s];
#pragma unroll
      for (int t = 0; t < RS; ++t) { npC[t] = npN[t]; npN[t] = 0; }
      knext = pop();
      if (knext >= 0) rowoffs(knext, roN, faN, npN);
    }
  }

  S3_EV(250);
#if IRX_S3_TRACE
  if (trc && lane == 0) trc[253] = (unsigned long long)itn;
#endif
  // ---- epilogue (as k_spconv3): one 32 x 32 block at a time through the wave's own LDS region ----
  __syncthreads();
  float* so = reinterpret_cast<float*>(smem) + wave * 32 * LDO;
  const int col = lane & 31, r4 = 4 * (lane >> 5);
#pragma unroll
  for (int t = 0; t < RS; ++t)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) so[((reg & 3) + 8 * (reg >> 2) + r4) * LDO + col] = acc[t][cb][reg];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int f = lane + 64 * it;
        const int row = f >> 3, cc = (f & 7) * 4;
        const int gr = q0 + wave * WR + 32 * t + row;
        if (gr < n_out) {
          float4 o = *reinterpret_cast<const float4*>(&so[row * LDO + cc]);
          const size_t off = (size_t)gr * COUT + cb * 32 + cc;
          if (accumulate) {
            const float4 e = irx_ld4(y, off, y_bf);
            o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
          }
          irx_st4(y, off, y_bf, o);
        }
      }
    }
  S3_EV(251);
}

// ---- weight image: [K][NS][NCB][64 lanes][8 bf16]; lane l of fragment (s, cb) holds W[k][c = chan(s, l >> 5, j)][n = 32 cb + (l & 31)]
//   forward      : W[k][c][n] = w[(k*cin + c)*cout + n]
//   data-gradient: W[k][c][n] = w[(k*cout + n)*cin + c]      (kernel-relative cin = reduction, cout = outputs)
__device__ __forceinline__ void s3_image_piece(const float* __restrict__ w, int cin, int cout, int trans_w, size_t piece,
                                               uint4* __restrict__ dst) {
  const int ncb = cout / 32, ns = cin / 16;
  const int lane = (int)(piece & 63);
  size_t r = piece >> 6;
  const int cb = (int)(r % ncb); r /= ncb;
  const int s = (int)(r % ns); r /= ns;
  const int k = (int)r;
  const int n = cb * 32 + (lane & 31), h = lane >> 5;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = IRX_S3_AMAP ? (cin / 2) * h + 8 * s + j : 16 * s + 8 * h + j;
    v[j] = trans_w ? w[((size_t)k * cout + n) * cin + c] : w[((size_t)k * cin + c) * cout + n];
  }
  dst[piece] = make_uint4(irx_pk_bf16(v[0], v[1]), irx_pk_bf16(v[2], v[3]), irx_pk_bf16(v[4], v[5]), irx_pk_bf16(v[6], v[7]));
}

__global__ void k_permute_w3(const float* __restrict__ w, int K, int cin, int cout, int trans_w, uint4* __restrict__ dst) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (size_t)K * cin * cout / 8) return;
  s3_image_piece(w, cin, cout, trans_w, p, dst);
}

// all third-generation layers of an encoder pass in one launch (end4 counts float4 = 4 weights; a piece holds 8)
__global__ void k_permute_w3_multi(IrxPermuteJobs J, int trans_w) {
  const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int j = 0;
  size_t begin = 0;
  while (j < J.n) {
    const size_t pieces = (size_t)J.K[j] * J.cin[j] * J.cout[j] / 8;
    if (f < begin + pieces) break;
    begin += pieces;
    ++j;
  }
  if (j >= J.n) return;
  s3_image_piece(J.w[j], J.cin[j], J.cout[j], trans_w, f - begin, reinterpret_cast<uint4*>(J.dst[j]));
}

// Dev: resident workgroups per CU the runtime reports (tools/micro/occupancy.py)
extern "C" int irx_debug_occupancy_s3(int which) {
  int n = -1;
  hipError_t e = hipSuccess;
  switch (which) {
    case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spconv3<128, 128, 4, 2>, 256, 0); break;
    case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spconv3<64, 64, 4, 1>, 256, 0); break;
    case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spconv3<128, 128, 4, 1>, 256, 0); break;
    case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spconv3<64, 128, 4, 1>, 256, 0); break;
    default: return -2;
  }
  return e == hipSuccess ? n : -1;
}

// ---------------------------------------------------------------------------- host side -----
bool irx_spconv3_supported(int cin, int cout) {
  auto ok = [](int c) { return c == 32 || c == 64 || c == 128; };
  return ok(cin) && ok(cout) && cin * cout >= 2048;
}

// IRX_SPCONV3=0 / irx_debug_set_knob("spconv3", 0): the bf16-storage convs go back to k_spconv2 (A/B and the equality test)
bool irx_spconv3_enabled() { return irx_knob(IRX_KNOB_SPCONV3) != 0; }

// x_rows: rows of the bf16 input tensor (0 = unknown -> not taken: rows are addressed with 32-bit byte offsets)
bool irx_spconv3_use(int cin, int cout, int x_bf, long long x_rows, int ldx) {
  if (!x_bf || !irx_spconv3_enabled() || !irx_spconv3_supported(cin, cout) || x_rows <= 0) return false;
  if (ldx <= 0) ldx = cin;
  return x_rows * (long long)ldx * 2 < 0x7FFFFFF0ll;
}

// waves per workgroup (32 output rows each): dev knob IRX_S3_NW = 4 | 8
static int s3_nw() {
  static const int v = getenv("IRX_S3_NW") ? atoi(getenv("IRX_S3_NW")) : 4;
  return v == 8 ? 8 : 4;
}
int irx_spconv3_tile() { return 32 * s3_nw(); }

// offsets are split over workgroups when a level has too few 128-row tiles to fill the 256 CUs twice
int irx_spconv3_splits(int n_out, int K) {
  static const char* e = getenv("IRX_SPCONV3_KSPLIT");
  if (e) { int s = atoi(e); return s < 1 ? 1 : (s > K ? K : s); }
  const int tiles = irx_cdiv(n_out, irx_spconv3_tile());
  static const int full = getenv("IRX_SPCONV3_SPLIT_BELOW") ? atoi(getenv("IRX_SPCONV3_SPLIT_BELOW")) : 384;
  if (tiles >= full || K < 4) return 1;
  static const int target = getenv("IRX_SPCONV3_SPLIT_TARGET") ? atoi(getenv("IRX_SPCONV3_SPLIT_TARGET")) : 448;
  int s = irx_cdiv(target, tiles);
  if (s > 9) s = 9;
  if (s > K) s = K;
  const int kps = irx_cdiv(K, s);
  return irx_cdiv(K, kps);
}

int irx_permute_w3_launch(const float* w, int K, int cin, int cout, int trans_w, float* dst, hipStream_t st) {
  const size_t pieces = (size_t)K * cin * cout / 8;
  k_permute_w3<<<irx_cdiv((long long)pieces, 256), 256, 0, st>>>(w, K, cin, cout, trans_w, reinterpret_cast<uint4*>(dst));
  IRX_CHECK_LAUNCH("irx_spconv_fwd(permute v3)");
  return IRX_OK;
}

int irx_permute_w3_multi_launch(const IrxPermuteJobs& jobs, int trans_w, hipStream_t st) {
  if (jobs.n == 0) return IRX_OK;
  size_t pieces = 0;
  for (int j = 0; j < jobs.n; ++j) pieces += (size_t)jobs.K[j] * jobs.cin[j] * jobs.cout[j] / 8;
  k_permute_w3_multi<<<irx_cdiv((long long)pieces, 256), 256, 0, st>>>(jobs, trans_w);
  IRX_CHECK_LAUNCH("irx_encoder(permute v3)");
  return IRX_OK;
}

template <int CIN, int NW, int KH>
static void launch3(int cout, dim3 grid, hipStream_t st, const unsigned short* x, const uint4* wimg, const int32_t* nbr, int ld,
                    int n_out, int K, int flip_k, float* y, int kps, int acc, int ldx, int y_bf, int xt, int sp) {
  if (cout == 128) k_spconv3<CIN, 128, NW, KH><<<grid, 64 * NW, 0, st>>>(x, wimg, nbr, ld, n_out, K, flip_k, y, kps, acc, ldx, y_bf, xt, sp);
  else if constexpr (KH == 1) {
    if (cout == 64) k_spconv3<CIN, 64, NW, 1><<<grid, 64 * NW, 0, st>>>(x, wimg, nbr, ld, n_out, K, flip_k, y, kps, acc, ldx, y_bf, xt, sp);
    else if constexpr (CIN >= 64) k_spconv3<CIN, 32, NW, 1><<<grid, 64 * NW, 0, st>>>(x, wimg, nbr, ld, n_out, K, flip_k, y, kps, acc, ldx, y_bf, xt, sp);
  }
}

// x: bf16 [rows][ldx]; wimg: irx_permute_w3* image; y: result (splits == 1; accumulate adds to it; y_bf = bf16 tensor) or
// `splits` fp32 slabs [n_out][cout]
int irx_spconv3_launch(const float* x, const float* wimg, const int32_t* nbr, int ld, int n_out, int K, int cin, int cout,
                       int flip_k, float* y, int splits, int accumulate, hipStream_t st, int ldx, int y_bf) {
  if (ldx <= 0) ldx = cin;
  IRX_REQUIRE(irx_spconv3_supported(cin, cout), "irx_spconv3: channels (%d, %d) unsupported", cin, cout);
  IRX_REQUIRE(K <= 27, "irx_spconv3: K = %d > 27", K);
  IRX_REQUIRE(splits == 1 || (!accumulate && !y_bf), "irx_spconv3: offset-split slabs are plain fp32");
  // rows per workgroup of the instantiation launched below: 32 rows per wave; Cin = 32 always runs 4 waves (IRX_S3_NW=8 only has
  // 64- / 128-channel instantiations — the grid must follow the launch, not the knob: ADVICE r4)
  const int tile = (s3_nw() == 8 && cin != 32) ? 256 : 128;
  const int tiles = irx_cdiv(n_out, tile);
  // XCD-contiguous tile ranges (k_spconv3's work-unit comment) from "spconv3_xcd_min" tiles on; below that a level's rows fit
  // every L2 anyway
  const long xmin = irx_knob(IRX_KNOB_SPCONV3_XCD_MIN);
  const int xt = (xmin > 0 && tiles >= xmin) ? irx_cdiv(tiles, 8) : 0;
  const dim3 grid = xt ? dim3(8 * xt * splits) : dim3(tiles, splits);
  const int kps = irx_cdiv(K, splits);
  const unsigned short* xb = reinterpret_cast<const unsigned short*>(x);
  const uint4* wi = reinterpret_cast<const uint4*>(wimg);
  // fourth generation (k_spconv4: all-LDS-DMA, row-shaped gathers) for 64 / 128 input channels; "spconv4" = 0: k_spconv3
  // policy ("spconv4"): 1 = the 128 -> 128 layers only (measured, kernel alone, B = 16 pyramid: 69-73 vs 79 us on the 81 k-row
  // level, 28.0 vs 31.0 on 20 k rows, 19.8 vs 20.0 on 4.6 k; the 64-channel and 128 -> 64 shapes are SLOWER than k_spconv3:
  // 90 vs 75 us and 48 vs 39 us — a 128-byte row is one instruction either way and the image is small), 2 = those layers with
  // two row sets per wave (dev: 87 us), 3 = every 64 / 128-input-channel shape (dev)
  const long s4 = irx_knob(IRX_KNOB_SPCONV4);
  if (s4 != 0 && (cin == 64 || cin == 128) && s3_nw() == 4 && (s4 == 3 || (cin == 128 && cout == 128))) {
    irx_bracket_begin(st);
#define S4_GO(CIN_, COUT_) do {                                                                                                      \
      if (rs2) k_spconv4<CIN_, COUT_, 2, 2><<<grid, 128, 0, st>>>(xb, wi, nbr, ld, n_out, K, flip_k, y, kps, accumulate, ldx, y_bf, xt, splits); \
      else k_spconv4<CIN_, COUT_, 4, 1><<<grid, 256, 0, st>>>(xb, wi, nbr, ld, n_out, K, flip_k, y, kps, accumulate, ldx, y_bf, xt, splits);  \
    } while (0)
    // "spconv4" = 2: two waves of 64 rows (two row sets: every B fragment read feeds two MFMAs); 1: four waves of 32 rows
    const bool rs2 = irx_knob(IRX_KNOB_SPCONV4) == 2 && cout >= 64;
    if (cin == 128) { if (cout == 128) S4_GO(128, 128); else if (cout == 64) S4_GO(128, 64); else k_spconv4<128, 32, 4, 1><<<grid, 256, 0, st>>>(xb, wi, nbr, ld, n_out, K, flip_k, y, kps, accumulate, ldx, y_bf, xt, splits); }
    else { if (cout == 128) S4_GO(64, 128); else if (cout == 64) S4_GO(64, 64); else k_spconv4<64, 32, 4, 1><<<grid, 256, 0, st>>>(xb, wi, nbr, ld, n_out, K, flip_k, y, kps, accumulate, ldx, y_bf, xt, splits); }
#undef S4_GO
    irx_bracket_end(st);
    IRX_CHECK_LAUNCH("irx_spconv_fwd(v4)");
    return IRX_OK;
  }
  irx_bracket_begin(st);
#define S3_GO(CIN_, NW_, KH_) launch3<CIN_, NW_, KH_>(cout, grid, st, xb, wi, nbr, ld, n_out, K, flip_k, y, kps, accumulate, ldx, y_bf, xt, splits)
  // 128 -> 128: the reduction in two half-items (three resident workgroups per CU); dev knob IRX_S3_KH=1: whole offsets
  static const int kh_env = getenv("IRX_S3_KH") ? atoi(getenv("IRX_S3_KH")) : 2;
  if (s3_nw() == 8) {
    if (cin == 128) S3_GO(128, 8, 1);
    else if (cin == 64) S3_GO(64, 8, 1);
    else S3_GO(32, 4, 1);
  } else {
    if (cin == 128 && cout == 128 && kh_env == 2) S3_GO(128, 4, 2);
    else if (cin == 128) S3_GO(128, 4, 1);
    else if (cin == 64) S3_GO(64, 4, 1);
    else S3_GO(32, 4, 1);
  }
#undef S3_GO
  irx_bracket_end(st);
  IRX_CHECK_LAUNCH("irx_spconv_fwd(v3)");
  return IRX_OK;
}
