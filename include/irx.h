/*
 * irx.h — C-ABI of libirx.so: the MI355X (gfx950 / CDNA4) native operator library behind
 * InstanceRefer's data-parallel hot path (sparse-voxel 3D conv backbone, per-instance
 * extractors, language-instance matching).
 *
 * What this boundary replaces (reference = CurryYuan/InstanceRefer, paths relative to its root):
 * the reference owns no native code; its hot path calls pybind11 back-ends of un-vendored
 * third-party packages:
 *   torchsparse  (hash / hash-query / neighbour-map / gather-GEMM-scatter conv / voxelise)
 *       call sites  models/basic_blocks.py:14-21,32-44,182   models/attribute_module.py:20,65-70,101,105
 *                   models/scene_module.py:20                lib/dataset.py:229-234,256-261,458
 *   torch_cluster.knn + torch_scatter max  (through torch_geometric)
 *       call sites  models/basic_blocks.py:100,120,125
 * Every entry point below names the call site(s) it stands in for.
 *
 * Conventions (all functions):
 *   - `extern "C"`, plain pointers + sizes; no torch / STL types cross the boundary.
 *   - Every pointer is a DEVICE pointer unless the parameter name ends in `_host`.
 *   - The CALLER owns every buffer, including workspaces (sizes from the `*_workspace_bytes`
 *     queries, which are pure host arithmetic). Functions never allocate device memory,
 *     never synchronise the device and never throw.
 *   - Work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream).
 *   - Return 0 on success, a negative irx_status otherwise; `irx_last_error()` returns a
 *     thread-local, human-readable message for the last failure on the calling thread.
 *   - Re-entrant across streams; safe to call from PyTorch's autograd worker thread.
 *
 * Data layout in HBM:
 *   coords   int32  [N][4]   (x, y, z, batch) — torchsparse's `C` layout, batch LAST
 *                            (models/basic_blocks.py:176,233 index columns 0..2 / 3).
 *                            x,y,z are in ORIGINAL-resolution voxel units (multiples of the
 *                            tensor stride), each within [-32768, 32767]; batch in [0, 32767].
 *   feats    float  [N][C]   row-major, C contiguous (torchsparse's `F`).
 *   keys     uint64 [N]      (batch << 48) | morton3(x+32768, y+32768, z+32768): the library's
 *                            canonical order is ascending key (Z-order inside each batch item).
 *   nbr      int32  [K][ld]  output-stationary neighbour table: nbr[k][q] = row index of the
 *                            INPUT voxel at coord(q) + offset_k, or -1.  (k-major, ld >= n_out.)
 *   weights  float  [K][Cin][Cout]   torchsparse `Conv3d.kernel` layout (state-dict compatible).
 *   Kernel-offset enumeration follows torchsparse: odd kernel sizes
 *   enumerate x fastest ({-1,0,1}*stride), even sizes enumerate z fastest ({0,1}*stride).
 */
#ifndef IRX_H_
#define IRX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRX_VERSION_MAJOR 0
#define IRX_VERSION_MINOR 1

typedef enum irx_status {
  IRX_OK = 0,
  IRX_ERR_INVALID_ARG = -1,   /* bad size / null pointer / unsupported channel count      */
  IRX_ERR_LAUNCH = -2,        /* hipLaunchKernel or hipMemsetAsync reported an error       */
  IRX_ERR_WORKSPACE = -3,     /* workspace too small                                       */
  IRX_ERR_NO_DEVICE = -4      /* no HIP device visible                                     */
} irx_status;

/* ---- library ------------------------------------------------------------------------- */
int irx_version(void);                       /* major*1000 + minor                          */
const char* irx_last_error(void);            /* thread-local message of the last failure    */
/* Fills up to 8 ints: {CU count, wavefront size, LDS bytes/CU, L2 bytes, clock kHz,
 * gcn arch number (950), total HBM MiB, 0}. Host call; returns IRX_ERR_NO_DEVICE w/o GPU.  */
int irx_device_props(int device, int* out8_host);

/* ---- coordinates, hashing, voxelisation --------------------------------------------- */

/* keys[i] = (b<<48)|morton(x,y,z) for coords[i]. */
int irx_coords_to_keys(const int32_t* coords, int n, uint64_t* keys, void* stream);
/* ... and back: coords[i] = (x, y, z, b) of keys[i] (int32 [n][4]). The voxeliser takes the coordinate rows of its
 * Morton-sorted voxels from the sorted keys (a streaming pass) instead of gathering them through the sort permutation. */
int irx_keys_to_coords(const uint64_t* keys, int n, int32_t* coords, void* stream);

/* Stable ascending radix sort (8-bit digits, LSD) of 64-bit keys by their bits [begin_bit, end_bit), with the permutation
 * it applies: keys_out[i] = keys[order_out[i]]. The ordering step of the voxeliser: rows of every SparseTensor are kept
 * in ascending Morton-key order — what torchsparse obtains from `np.unique(hash, return_index=True)` inside
 * sparse_quantize (reference call sites models/attribute_module.py:65-69, lib/dataset.py:229-233,256-260), and the grouping
 * of sampled points by instance slot in the device-side input pipeline (lib/dataset.py:207-213: np.nonzero per instance id).
 * n_dev != NULL: only the first *n_dev elements are real (the count lives on the device: no host sync); the remaining
 * n - *n_dev elements are treated as the key `pad` (must compare above every real key within the sorted bits) and come out
 * behind the real ones. Deterministic (no global atomics). workspace: irx_sort_workspace_bytes(n).
 * NOTE (whole digits): the sort runs ceil((end_bit - begin_bit) / 8) passes of 8 bits, so it orders by bits
 * [begin_bit, begin_bit + 8 * passes) — key bits between end_bit and that digit boundary TAKE PART in the ordering (unlike
 * CUB's begin/end-bit semantics). Callers keep those bits zero in real keys (the in-tree ones do: Morton keys carry nothing
 * above their batch field) or want them sorted too (the `pad` key). */
size_t irx_sort_workspace_bytes(int n);
int irx_sort_pairs_u64(const uint64_t* keys, int n, const int32_t* n_dev, uint64_t pad, int begin_bit, int end_bit,
                       uint64_t* keys_out, int32_t* order_out, void* workspace, size_t workspace_bytes, void* stream);


/* Quantise points to voxel coordinates exactly as torchsparse.utils.sparse_quantize does
 * before hashing: coord = floor(xyz / voxel) evaluated in float64 (no min-shift; negatives
 * kept) — replaces the `np.floor(coords / quantization_size)` step reached from
 * models/attribute_module.py:65-69 and lib/dataset.py:229-233,256-260.
 * xyz: [n][3] (float64 when xyz_is_f64 else float32), batch: int32 [n] or NULL (=0).
 * Writes coords int32 [n][4] (skipped when coords == NULL) and keys uint64 [n]. A point whose voxel coordinate falls outside [-32768, 32768) (or
 * whose batch index is outside [0, 32768), or that is NaN) cannot be keyed: it receives a poison key, and
 * irx_voxel_select reports it by returning a NEGATIVE count (sign bit set) — callers must treat that as an error. */
int irx_quantize(const void* xyz, int xyz_is_f64, const int32_t* batch, int n,
                 double voxel_x, double voxel_y, double voxel_z,
                 int32_t* coords, uint64_t* keys, void* stream);

/* Capacity (number of slots, a power of two >= 2n) of the open-addressing table for n keys. */
size_t irx_hash_capacity(int n);

/* Voxel de-duplication with torchsparse's FIRST-OCCURRENCE rule (np.unique(return_index=True)
 * keeps the smallest original index of each voxel): inserts every key with
 * atomicMin(point index); on return table_vals[slot] = smallest point index of that voxel.
 * table_keys/table_vals: [capacity]; the function clears them itself. */
int irx_voxel_insert(const uint64_t* keys, int n, uint64_t* table_keys, int32_t* table_vals,
                     size_t capacity, void* stream);

/* Wave-ballot + prefix-sum compaction of the winners of irx_voxel_insert: writes the point
 * indices whose voxel they represent to winners[0 .. *count) (unordered; the caller sorts
 * by key) and the number of voxels to count[0] (device int32, cleared by the function). */
int irx_voxel_select(const uint64_t* keys, int n, const uint64_t* table_keys,
                     const int32_t* table_vals, size_t capacity, int32_t* winners,
                     int32_t* count, void* stream);

/* Build key -> row index table for n UNIQUE keys (torchsparse `sphash` + hashmap insert). */
int irx_hash_build(const uint64_t* keys, int n, uint64_t* table_keys, int32_t* table_vals,
                   size_t capacity, void* stream);

/* Neighbour table for a stride-preserving convolution (torchsparse `sphash(coords, offsets)`
 * + `sphashquery`, reached from spnn.Conv3d at models/basic_blocks.py:14,32,39).
 * kernel_size must be 3 (K = 27). tensor_stride = current stride s; offsets {-s,0,s}^3,
 * x fastest.  nbr: int32 [27][ld]. */
int irx_kmap_build_s1(const int32_t* coords, int n, int tensor_stride,
                      const uint64_t* table_keys, const int32_t* table_vals, size_t capacity,
                      int32_t* nbr, int ld, void* stream);

/* Hash tables AND 27-neighbour tables of up to 8 levels of one coordinate pyramid in ONE call (three launches in total: table fill,
 * inserts, neighbour search; round 5 took four launches per level). Per level l: keys[l] uint64 [n[l]] (Morton keys, ASCENDING — the
 * rows of every level of a pyramid built by irx_pyramid_build are), coords[l] int32 [n[l]][4], tensor_stride[l]; the level's hash
 * table tk[l] / tv[l] of capacity cap[l] (irx_hash_capacity(n[l])) is (re)built as irx_hash_build does; nbr[l] int32 [27][ld[l]] as
 * irx_kmap_build_s1. Levels with n[l] == 0 are skipped. The neighbour search stages a window of each workgroup's sorted keys in LDS
 * and only consults the hash table for probes outside that window's key range (csrc/irx_coords.hip, k_kmap_win); irx_kmap_build_s1
 * runs the same kernel for one level. Reference: torchsparse's `sphash` + `sphashquery` per Conv3d (models/basic_blocks.py:14-19). */
int irx_kmaps_build_multi(int nlev, const uint64_t* const* keys, const int32_t* const* coords, const int* n,
                          const int* tensor_stride, uint64_t* const* table_keys, int32_t* const* table_vals, const size_t* capacity,
                          int32_t* const* nbr, const int* ld, void* stream);

/* The 27-neighbour tables of EVERY level of a coordinate pyramid built by irx_pyramid_build, by octree descent: the coarsest level
 * (index nlev - 1) is searched like irx_kmaps_build_multi does (its hash table tk_top / tv_top of capacity cap_top is built and used
 * only when that level has more than 2048 rows; otherwise the three may be NULL / 0), every finer level l is derived from level l + 1:
 * nbr_l[d][q] = child_l[(b + d) mod 2][ nbr_{l+1}[floor((b + d) / 2)][parent_l[q]] ] with b = the child position of row q inside its
 * parent (koff_l[q] = x * 4 + y * 2 + z) — parent_l int32 [n[l]], koff_l uint8 [n[l]], child_l int32 [8][child_ld[l]] are exactly the
 * arrays irx_pyramid_build / irx_downsample wrote for the map l -> l + 1. No hash table and no key arithmetic below the top: one launch
 * per level, reads of rows that are neighbours in Morton order. n[l], tensor_stride[l], nbr[l] int32 [27][ld[l]] per level (finest
 * first). Same tables as irx_kmap_build_s1 (tests/test_kmaps_gpu.py). Reference: torchsparse's kernel-map construction per
 * Conv3d (models/basic_blocks.py:14-19). */
int irx_kmaps_build_pyramid(int nlev, const int* n, const int* tensor_stride, const uint64_t* keys_top, const int32_t* coords_top,
                            uint64_t* table_keys_top, int32_t* table_vals_top, size_t capacity_top, const int32_t* const* parent,
                            const uint8_t* const* koff, const int32_t* const* child, const int* child_ld, int32_t* const* nbr,
                            const int* ld, void* stream);

/* Strided (kernel 2, stride 2) down-sampling of a key-sorted coordinate set
 * (torchsparse `spdownsample` + kernel map, reached from BasicConvolutionBlock(ks=2,
 * stride=2) at models/basic_blocks.py:68,73,78,83): out coords = unique(floor(c/(2s))*2s, b).
 * Because rows are in ascending Morton-key order, parents are a segmented scan:
 *   parent[i]   int32 [n]      row of the output voxel that input row i falls into
 *   koff[i]     uint8 [n]      kernel-offset index of input i inside its parent (z fastest)
 *   out_coords  int32 [cap][4], out_keys uint64 [cap]   (cap >= n is always enough)
 *   child       int32 [8][ld]  child[k][p] = input row at parent p + offset_k, or -1
 *                              (the function fills it with -1 first; ld >= cap)
 *   n_out       int32 [1]      device counter
 * workspace: irx_downsample_workspace_bytes(n). */
size_t irx_downsample_workspace_bytes(int n);
int irx_downsample(const uint64_t* keys, const int32_t* coords, int n, int tensor_stride,
                   int32_t* parent, uint8_t* koff, int32_t* out_coords, uint64_t* out_keys,
                   int32_t* child, int ld, int32_t* n_out, void* workspace,
                   size_t workspace_bytes, void* stream);

/* The whole `levels`-deep coordinate pyramid of one tensor (irx_downsample applied `levels` times) in ONE call and
 * without a host sync between levels: level l takes its row count from counts[l-1] ON THE DEVICE. Host arrays of
 * `levels` device pointers; every per-level buffer holds n0 rows (the upper bound), child tables int32 [8][ld], ld >= n0.
 * counts int32 [levels] (device): rows of each down-sampled level; one D2H copy of it replaces a sync per level.
 * n0_dev (device, may be NULL): the input's actual row count when n0 is only an upper bound (un-synchronised voxeliser).
 * workspace: irx_downsample_workspace_bytes(n0). */
int irx_pyramid_build(const uint64_t* keys0, const int32_t* coords0, int n0, int stride0, int levels,
                      int32_t* const* parent, uint8_t* const* koff, int32_t* const* out_coords,
                      uint64_t* const* out_keys, int32_t* const* child, int ld, int32_t* counts,
                      const int32_t* n0_dev, void* workspace, size_t workspace_bytes, void* stream);

/* Transposed table of a down-sampling map for its data-gradient: tbl[k][i] = parent[i] if
 * koff[i]==k else -1  (int32 [8][ld], ld >= n). */
int irx_kmap_down_transpose(const int32_t* parent, const uint8_t* koff, int n, int32_t* tbl,
                            int ld, void* stream);

/* Table for SparseCrop + ToDenseBEVConvolution (models/basic_blocks.py:174-243,
 * models/scene_module.py:22-27): dense cell c = (b, ix, iy), ix < nx, iy < ny; for each
 * z-bin k < nz looks up voxel (ix*s, iy*s, k*s, b) -> tbl[k][c] (int32 [nz][ld]); also the
 * transposed form for the data-gradient: cell_of_row int32 [n] (-1 when outside the crop
 * window) and zbin_of_row uint8 [n]. */
int irx_bev_table(const int32_t* coords, int n, int tensor_stride, int batch_size, int nx,
                  int ny, int nz, const uint64_t* table_keys, const int32_t* table_vals,
                  size_t capacity, int32_t* tbl, int ld, int32_t* cell_of_row,
                  uint8_t* zbin_of_row, void* stream);

/* ---- sparse convolution (spnn.Conv3d fwd + bwd; models/basic_blocks.py:14-19,32-43) --- */

/* Compute dtype of the MFMA sparse-conv kernels (channel counts 32/64/128; forward, data- and weight-gradient):
 * 0 (default) = exact fp32 (v_mfma_f32_16x16x4_f32: the reference's dtype, the 1e-4 parity gate);
 * 1 = bf16 operands with fp32 accumulation (BASELINE configs[2]-[4]: x, w, dy are rounded to bf16, round-to-nearest-
 * even, as they enter the matrix core; every tensor in HBM, BatchNorm statistics and all other kernels stay fp32).
 * Process-wide; set it before a step, not concurrently with running calls. */
int irx_set_compute_dtype(int bf16);
int irx_get_compute_dtype(void);

/* y[q][:] = sum_k x[nbr[k'][q]][:] * W[k]    (k' = K-1-k when flip_k, else k)
 * with W[k] = w[k][cin][cout]               when !trans_w   (forward)
 *      W[k] = transpose(w[k][cout'][cin'])  when  trans_w   (data-gradient: pass x = dy,
 *             cin = conv's Cout, cout = conv's Cin, w = the forward weight unchanged)
 * Output-stationary: every output row is written exactly once (no atomics; deterministic).
 * fp32 in / fp32 accumulate on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).
 * Any cin / cout; channel counts in {32,64,128} take the pair-compacting, weight-stationary fast path, which
 * needs a workspace (irx_spconv_fwd_workspace_bytes) for a fragment-major weight image and, for small layers,
 * the partial-sum slabs of the offset splits (summed deterministically).
 * y: [n_out][cout]. */
size_t irx_spconv_fwd_workspace_bytes(int n_out, int K, int cin, int cout, int trans_w);
int irx_spconv_fwd(const float* x, const float* w, const int32_t* nbr, int ld, int n_out,
                   int K, int cin, int cout, int flip_k, int trans_w, float* y, void* workspace,
                   size_t workspace_bytes, void* stream);

/* irx_spconv_fwd with the element types of x and y stated (0 = float32, 1 = bf16 stored as uint16; a bf16 tensor needs
 * irx_set_compute_dtype(1 | 2)) — the convolution of the encoder executor's bf16 STORAGE mode as a single operator
 * (BASELINE configs[2]-[4]; reference dtype fp32, models/basic_blocks.py:10-95). x: [n_in][cin], y: [n_out][cout];
 * accumulate != 0: y += result. A bf16 x with channel counts in {32,64,128} (cin*cout >= 2048) and n_in*cin*2 < 2 GiB
 * takes the third-generation kernel (irx_spconv3.hip: 128-row output tiles with register accumulators, rows gathered
 * straight into v_mfma_f32_32x32x16_bf16 operands, W[k] double-buffered in LDS); IRX_SPCONV3=0 keeps it on the second
 * generation. Same workspace query as irx_spconv_fwd. */
int irx_spconv_fwd_t(const void* x, const float* w, const int32_t* nbr, int ld, int n_in, int n_out, int K, int cin,
                     int cout, int flip_k, int trans_w, void* y, int accumulate, int x_bf, int y_bf, void* workspace,
                     size_t workspace_bytes, void* stream);

/* dw[k][ci][co] = sum_q x[nbr[k][q]][ci] * dy[q][co].  Deterministic two-stage reduction
 * through `workspace` (irx_spconv_wgrad_workspace_bytes). */
size_t irx_spconv_wgrad_workspace_bytes(int n_out, int K, int cin, int cout);
int irx_spconv_wgrad(const float* x, const float* dy, const int32_t* nbr, int ld, int n_out,
                     int K, int cin, int cout, float* dw, void* workspace,
                     size_t workspace_bytes, void* stream);

/* Per-offset compacted pair lists of a neighbour table, built on the device without a host sync:
 * in_list / out_list int32 [K][ldp] (ldp >= n_out): for offset k the valid entries of nbr[k][0..n_out) in
 * ascending output-row order; counts int32 [K] (device). Reused by every weight-gradient on that table. */
size_t irx_pairs_workspace_bytes(int n_out, int K);
int irx_pairs_build(const int32_t* nbr, int ld, int n_out, int K, int32_t* in_list, int32_t* out_list,
                    int ldp, int32_t* counts, void* workspace, size_t workspace_bytes, void* stream);
/* irx_pairs_build for up to 16 tables in one call (two launches in total): host arrays of per-table arguments.
 * workspace: the sum of irx_pairs_workspace_bytes(n_out[t], K[t]). */
int irx_pairs_build_multi(int n_tables, const int32_t* const* nbr, const int* ld, const int* n_out, const int* K,
                          int32_t* const* in_list, int32_t* const* out_list, const int* ldp, int32_t* const* counts,
                          void* workspace, size_t workspace_bytes, void* stream);
/* Weight-gradient over pair lists in DENSE MFMA stages (cin, cout in {32,64,128}); same result as
 * irx_spconv_wgrad up to fp32 summation order; deterministic. x: [n_in][cin], dy: [n_out][cout]; every entry of
 * in_list must be < n_in (the fp32 kernel addresses both tensors with 32-bit byte offsets when they are below 2 GiB). */
size_t irx_spconv_wgrad_pairs_workspace_bytes(int n_out, int K, int cin, int cout);
int irx_spconv_wgrad_pairs(const float* x, const float* dy, const int32_t* in_list, const int32_t* out_list,
                           int ldp, const int32_t* counts, int n_in, int n_out, int K, int cin, int cout, float* dw,
                           void* workspace, size_t workspace_bytes, void* stream);
/* irx_spconv_wgrad_pairs with the element type of x AND dy stated (rows_bf: 0 = float32, 1 = bf16 stored as uint16, needs
 * irx_set_compute_dtype(2)) — the weight gradient of the encoder executor's bf16 STORAGE mode as a single operator. bf16
 * rows with cin, cout in {64,128} take k_wgrad3 (rows stay bf16 in LDS, ds_read_b64_tr_b16 operand reads,
 * v_mfma_f32_32x32x16_bf16; IRX_WGRAD3=0 keeps the widening kernel); dw is float32 either way. */
int irx_spconv_wgrad_pairs_t(const void* x, const void* dy, const int32_t* in_list, const int32_t* out_list, int ldp,
                             const int32_t* counts, int n_in, int n_out, int K, int cin, int cout, float* dw, int rows_bf,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ---- BatchNorm(+residual)(+ReLU) over voxel rows (spnn.BatchNorm / spnn.ReLU and the
 *      residual add at models/basic_blocks.py:20-21,37-38,44,52,55) ----------------------- */

/* Train-mode statistics of x [n][c]: mean, biased var -> invstd = rsqrt(var+eps); updates
 * running_mean/var with `momentum` (unbiased var) when running_mean != NULL — nn.BatchNorm1d
 * semantics. workspace: irx_bn_workspace_bytes(n, c). */
size_t irx_bn_workspace_bytes(int n, int c);
int irx_bn_stats(const float* x, int n, int c, float eps, float momentum, float* mean,
                 float* invstd, float* running_mean, float* running_var, void* workspace,
                 size_t workspace_bytes, void* stream);

/* y = act( (x - mean) * invstd * gamma + beta (+ residual) ), act = ReLU when relu != 0. */
int irx_bn_apply(const float* x, int n, int c, const float* mean, const float* invstd,
                 const float* gamma, const float* beta, const float* residual, int relu,
                 float* y, void* stream);

/* irx_bn_stats followed by irx_bn_apply as ONE call (the form spnn.BatchNorm + spnn.ReLU (+ the shortcut add) take in training,
 * models/basic_blocks.py:20-21,37-38,52-55): mean / invstd are written for the backward, the running statistics updated, y =
 * act((x - mean) * invstd * gamma + beta (+ residual)). With relu != 0 a tensor small enough to stay on-die (IRX_BN_SLICE_BYTES,
 * default 6 MB) is ONE launch — a workgroup per 16-byte channel column walks all rows twice — instead of three dependent ones;
 * irx_bn_backward takes the same single-launch form for such a tensor. workspace: irx_bn_workspace_bytes(n, c). */
int irx_bn_forward(const float* x, int n, int c, float eps, float momentum, const float* gamma, const float* beta,
                   const float* residual, int relu, float* mean, float* invstd, float* running_mean,
                   float* running_var, float* y, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of irx_bn_apply∘irx_bn_stats (train mode). g = dy * (y > 0) when relu.
 * Outputs: dx [n][c]; dgamma, dbeta [c]; dresidual [n][c] (= g) when dresidual != NULL. */
int irx_bn_backward(const float* x, const float* y, const float* dy, int n, int c,
                    const float* mean, const float* invstd, const float* gamma, int relu,
                    float* dx, float* dgamma, float* dbeta, float* dresidual, void* workspace,
                    size_t workspace_bytes, void* stream);

/* irx_bn_forward / irx_bn_backward on tensors whose element type is chosen per tensor: 0 = float32, 1 = bf16 (round-to-nearest-
 * even on store, exact widening on load; statistics, arithmetic and parameter gradients stay float32 / float64) — the form the
 * encoder executor's bf16 storage mode runs (irx_set_compute_dtype(2)). bf16 tensors need c % 4 == 0 and 16-byte aligned
 * pointers. irx_bn_backward_ex, beta != NULL with relu and no dresidual: the layer had no shortcut, so y = relu(fma(x, invstd *
 * gamma, fma(-mean, invstd * gamma, beta))) — the ReLU mask is recomputed from x and y is not read. */
int irx_bn_forward_ex(const void* x, int n, int c, float eps, float momentum, const float* gamma, const float* beta,
                      const void* residual, int relu, float* mean, float* invstd, float* running_mean,
                      float* running_var, void* y, void* workspace, size_t workspace_bytes, void* stream, int x_bf,
                      int res_bf, int y_bf);
int irx_bn_backward_ex(const void* x, const void* y, const void* dy, int n, int c, const float* mean,
                       const float* invstd, const float* gamma, const float* beta, int relu, void* dx, float* dgamma,
                       float* dbeta, void* dresidual, void* workspace, size_t workspace_bytes, void* stream, int x_bf,
                       int y_bf, int dy_bf, int dx_bf, int dres_bf);

/* Sync BatchNorm (optional; the reference has no multi-GPU path — torch.nn.SyncBatchNorm is the contract):
 * statistics over the rows of ALL ranks. The library computes this rank's sums, the caller folds them over the ranks
 * (torch.distributed.all_reduce) and hands the folded sums back:
 *   forward : irx_bn_sums (sums[0..c) = sum x, sums[c..2c) = sum x^2, float64) -> all_reduce(sums, count) ->
 *             irx_bn_stats_from_sums (mean, invstd, running statistics with the GLOBAL count) -> irx_bn_apply;
 *   backward: irx_bn_backward_sums (dbeta = sum g, dgamma = sum g*xhat over THIS rank's rows: the parameter gradients, which
 *             the gradient all-reduce sums like any other) -> all_reduce(copies of both) -> irx_bn_backward_apply with the
 *             folded sums and the global row count.
 * The folded row count can stay on the device: count < 1 means "read it from sums[2 c]" (irx_bn_stats_from_sums: the caller
 * appends its row count to the sums before the fold) / from *count_dev (irx_bn_backward_apply) — no host round trip per layer.
 * workspace: irx_bn_workspace_bytes(n, c). n == 0 is legal everywhere (a rank without rows still joins the collectives). */
int irx_bn_sums(const float* x, int n, int c, double* sums, void* workspace, size_t workspace_bytes, void* stream);
int irx_bn_stats_from_sums(const double* sums, double count, int c, float eps, float momentum, float* mean,
                           float* invstd, float* running_mean, float* running_var, void* stream);
int irx_bn_backward_sums(const float* x, const float* y, const float* dy, int n, int c, const float* mean,
                         const float* invstd, int relu, float* dgamma, float* dbeta, void* workspace,
                         size_t workspace_bytes, void* stream);
int irx_bn_backward_apply(const float* x, const float* y, const float* dy, int n, int c, const float* mean,
                          const float* invstd, const float* gamma, int relu, const float* sum_g, const float* sum_gx,
                          double count, const double* count_dev, float* dx, float* dresidual, void* stream);

/* Measurement aid for bench.py's `roofline` object: the next dominant sparse-conv kernel launched by THIS host thread
 * (the MFMA kernel of irx_spconv_fwd / irx_spconv_wgrad / irx_spconv_wgrad_pairs, not the weight-permute or
 * split-reduce helpers that share the call) is bracketed with the two caller-owned HIP events on its launch stream.
 * One-shot: cleared after that kernel. Pass NULL, NULL to cancel. */
int irx_profile_next_kernel(void* ev_start, void* ev_stop);

/* Dev / test knobs. Each knob takes its default from an environment variable read ONCE (first use) and can then be changed
 * only through this setter (atomic; safe against the library's lane threads): "spconv3" (IRX_SPCONV3, 1: bf16-input convs
 * on the third-generation kernel), "updgrad" (IRX_UPDGRAD, 1) / "updgrad_min" (IRX_UPDGRAD_MIN, 40000: k_updgrad and its
 * size threshold), "wgrad_v1" (IRX_WGRAD_V1, 0: fp32 pair-list weight-gradient on the first-generation kernel), "wgrad3"
 * (IRX_WGRAD3, 1: bf16-row pair-list weight-gradient on k_wgrad3), "wgrad3_units" (IRX_WGRAD3_UNITS, 448: workgroups of its
 * XCD-segment mapping), "wgrad3_xcd_min" (IRX_WGRAD3_XCD_MIN, 200000: table entries n_out * K from which that mapping is used),
 * "wgrad_xcd_f32" (IRX_WGRAD_XCD_F32, 0: number of XCD-segment units for the fp32 pair-list kernels, 0 = their (share, offset) grid),
 * "enc_fold_slabs" (IRX_ENC_FOLD_SLABS, 1: inside irx_encoder_forward the offset-split slabs of a small level's convolution are
 * folded by the BatchNorm statistics pass instead of a reduce launch of their own; bit-identical), "enc_abl" (IRX_ENC_ABL, 0: TIMING
 * ONLY, results wrong — irx_encoder_backward without its weight-gradient (bit 0) / data-gradient (bit 1) launches), "stem_mfma"
 * (IRX_STEM_MFMA, 1: the 7-channel stem forward as an im2col tile + fp32 MFMA; 0: the vector-ALU kernel).
 * irx_debug_get_knob returns the value in force (-1: unknown name). Not part of the reference-facing surface. */
int irx_debug_set_knob(const char* name, long value);
long irx_debug_get_knob(const char* name);

/* ---- whole-encoder executor ------------------------------------------------------------
 * SparseConvEncoder.forward / BEVEncoder.forward (models/basic_blocks.py:59-95,136-171) and their backward as ONE
 * call per direction: for every layer  c = conv(x);  (mean, invstd) = stats(c);  y = relu(bn(c) (+ y[res]))  — the
 * same kernels, in the same order, as the per-layer entry points above (bit-identical results), walked by the host
 * side of the library instead of the Python caller.
 *
 * desc: host array [n_layers][IRX_ENC_NFIELDS] of int64 (sizes, indices and DEVICE pointers cast to int64);
 * fdesc: host array [n_layers][2] of double = {eps, momentum}.  The caller owns every buffer named in the table. */
enum {
  IRX_ENC_K = 0, IRX_ENC_CIN, IRX_ENC_COUT, IRX_ENC_N_IN, IRX_ENC_N_OUT,
  IRX_ENC_RES,                      /* index of the layer whose OUTPUT is added before the ReLU, or -1 */
  IRX_ENC_TBL, IRX_ENC_LD,          /* forward table int32 [K][ld]: nbr27 (stride 1) or child (2^3 stride 2) */
  IRX_ENC_TBL_B, IRX_ENC_LD_B, IRX_ENC_FLIP_B,   /* data-gradient table: same table + flip (stride 1) or child_T.
                                     * PRECONDITION for a 2^3 / stride-2 layer (TBL_B != TBL, K == 8): every input row is the child of
                                     * exactly one output row (true for irx_downsample / irx_pyramid_build maps). The fp32 parent-tiled
                                     * data-gradient (k_updgrad) writes each input-gradient row from its parent and does not zero rows
                                     * no parent references. */
  IRX_ENC_PAIR_IN, IRX_ENC_PAIR_OUT, IRX_ENC_PAIR_COUNTS, IRX_ENC_LD_PAIRS,   /* irx_pairs_build lists, or 0 */
  IRX_ENC_W, IRX_ENC_GAMMA, IRX_ENC_BETA, IRX_ENC_RUNNING_MEAN, IRX_ENC_RUNNING_VAR,
  IRX_ENC_X, IRX_ENC_C, IRX_ENC_Y,  /* layer input [n_in][cin], conv output and layer output [n_out][cout] */
  IRX_ENC_MEAN, IRX_ENC_INVSTD,     /* [cout] each */
  IRX_ENC_DW, IRX_ENC_DGAMMA, IRX_ENC_DBETA,     /* backward outputs */
  IRX_ENC_GY,                       /* backward: gradient w.r.t. the layer output [n_out][cout] */
  IRX_ENC_STORE,                    /* 0: every tensor fp32. 1 (needs irx_set_compute_dtype(1 | 2)): bf16 STORAGE — C, Y, GY
                                     * and the dc scratch are bf16 arrays (2 bytes / element) except the FIRST layer's X, the
                                     * LAST layer's Y and GY and dx0, which stay fp32; statistics / parameter gradients fp32 */
  IRX_ENC_MODE,                     /* compute mode of this pass (the values of irx_set_compute_dtype: 0 fp32, 1 bf16 operands, 2 +
                                     * bf16 storage), recorded by the caller when it builds the table for the forward pass and
                                     * carried into the backward pass: the executor uses THIS, not the process-wide setting */
  IRX_ENC_ORDER,                    /* 0, or a device pointer to the launch order of this layer's 64-row output tiles
                                     * (irx_tile_order over TBL; stride-1 layers only: the same order serves the data-gradient,
                                     * whose table is TBL with flipped offsets) */
  IRX_ENC_PROF,                     /* 0, or a HOST pointer to 6 event handles (hipEvent_t): start / stop around this layer's
                                     * dominant forward, data-gradient and weight-gradient kernel (measurement aid, bench.py) */
  IRX_ENC_DC2,                      /* row 0 only, backward only: 0, or a device pointer to a SECOND gradient scratch of the size of
                                     * dc_scratch. The pass then issues every layer's weight gradient on a stream of its own (the
                                     * library's, one per issuing thread) beside the BatchNorm-backward -> data-gradient chain of the next
                                     * layers, the two scratches alternating from layer to layer; results are bit-identical (same kernels,
                                     * same operands), everything is joined on the caller's stream before the call returns */
  IRX_ENC_WSTREAM,                  /* row 0 only, with IRX_ENC_DC2: 0 = a stream the library creates per issuing thread, else the
                                     * hipStream_t the weight gradients are issued on (a stream the caller already runs: every additional
                                     * stream of a process risks sharing a hardware queue with a busy one — measured: two library streams
                                     * beside the model's four HALVED the bf16 step) */
  IRX_ENC_WROWS,                    /* row 0 only, with IRX_ENC_DC2: 0 = every layer's weight gradient goes to the second stream, else only
                                     * those of layers with fewer than this many output rows (the latency-bound small levels, where the
                                     * GPU idles between the chain's short kernels; on the large levels two saturating kernel families
                                     * only take CUs from each other) */
  IRX_ENC_NFIELDS
};
/* Stream `to` continues behind everything enqueued on stream `from` so far (event record + wait; the library owns the events).
 * For callers that issue independent halves of one operator on two streams — plumbing, no reference counterpart. */
int irx_stream_fork(void* from, void* to);

/* Launch order of the 64-row output tiles of a stride-1 convolution over table `nbr` (int32 [K][ld], the irx_kmap_build_s1
 * table): order[i] = i-th tile to start, heaviest cost class first (cost = per active offset a fixed part + one part per
 * 16-pair group), ties in tile order; deterministic. Pure scheduling aid for irx_encoder_forward / _backward (IRX_ENC_ORDER):
 * torchsparse has no counterpart (its gather-GEMM-scatter launches are per offset, reference models/basic_blocks.py:32-44 reach
 * them through spnn.Conv3d); the convolution's results do not depend on it. workspace: irx_tile_order_workspace_bytes(n_out). */
size_t irx_tile_order_workspace_bytes(int n_out);
int irx_tile_order(const int32_t* nbr, int ld, int n_out, int K, int32_t* order, void* workspace, size_t workspace_bytes,
                   void* stream);
size_t irx_encoder_workspace_bytes(const int64_t* desc, const double* fdesc, int n_layers, int backward);
int irx_encoder_forward(const int64_t* desc, const double* fdesc, int n_layers, void* workspace,
                        size_t workspace_bytes, void* stream);
/* On entry GY of the last layer = d(loss)/d(output). Writes DW/DGAMMA/DBETA of every layer and, when dx0 != NULL,
 * dx0 [n_in0][cin0]; GY of the other layers and dc_scratch [max n_out*cout] are scratch. */
int irx_encoder_backward(const int64_t* desc, const double* fdesc, int n_layers, float* dc_scratch, float* dx0,
                         void* workspace, size_t workspace_bytes, void* stream);
/* Sync BatchNorm inside the one-call executor (torch.nn.SyncBatchNorm's contract — the reference has no multi-GPU
 * path): the same passes with the cross-rank fold of every layer's statistics done by the CALLER through `allreduce`, which the
 * library calls on the calling thread between a layer's statistics and apply pass (forward: 2 cout + 1 float64 = sum x, sum x^2,
 * row count; backward: 2 cout float32 = sum g, sum g xhat): it must enqueue an in-place SUM over the ranks of buf[0..n) on
 * `stream` (or complete it) and return 0. sums: device [n_layers][IRX_ENC_SYNC_STRIDE] float64, written by the forward and read
 * back by the backward (the folded row counts); gsums: device [n_layers][IRX_ENC_SYNC_STRIDE] float32 scratch. Parameter
 * gradients stay this rank's own sums. Every rank must call with the same layer list. Not available through irx_encoder_submit. */
#define IRX_ENC_SYNC_STRIDE 264
typedef int (*irx_allreduce_fn)(void* user, void* buf, int n, int is_double, void* stream);
int irx_encoder_forward_sync(const int64_t* desc, const double* fdesc, int n_layers, void* workspace, size_t workspace_bytes,
                             void* stream, double* sums, irx_allreduce_fn allreduce, void* user);
int irx_encoder_backward_sync(const int64_t* desc, const double* fdesc, int n_layers, float* dc_scratch, float* dx0,
                              void* workspace, size_t workspace_bytes, void* stream, double* sums, float* gsums,
                              irx_allreduce_fn allreduce, void* user);

/* Asynchronous issue of the two calls above: a lane (0..3) is a library thread that performs the call from a copy of
 * the descriptor table while the caller's thread goes on issuing independent work (Python: the ctypes call returns at
 * once). backward == 0: irx_encoder_forward (dc_scratch / dx0 ignored). Jobs of a lane run in submission order. The
 * caller keeps every device buffer alive, and enqueues nothing that depends on the pass on its stream — nor records an
 * event there — before irx_encoder_wait(lane) returned; that call blocks until the lane is idle and returns the first
 * failing status since the previous wait (message in irx_last_error()). */
int irx_encoder_submit(int lane, int backward, const int64_t* desc, const double* fdesc, int n_layers,
                       float* dc_scratch, float* dx0, void* workspace, size_t workspace_bytes, void* stream);
int irx_encoder_wait(int lane);

/* Orders the backward passes of TWO encoders of one training step (reference: models/instancerefer.py:38-53 runs the scene and the
 * candidate encoder one after the other; here their passes run on two streams).  Applies to the NEXT irx_encoder_backward /
 * irx_encoder_submit(backward) of the calling thread.  role 1 (recorder): an event is recorded on that pass's stream where its
 * levels of fewer than `rows` rows are done (at its end if it has no larger level); role 2 (waiter): that pass's stream waits
 * for the event before its first level of `rows` or more rows.  `token` pairs the two calls of a step (ascending from step to
 * step).  A waiter whose recorder has not been issued within 1.5 ms goes on without the gate.  role 0 clears. */
int irx_encoder_gate_next(int role, long long rows, unsigned long long token);


/* ---- segmented reductions ------------------------------------------------------------- */

/* spnn.GlobalMaxPooling (models/attribute_module.py:20,105) and the `aggr='max'` of
 * MessagePassing (models/basic_blocks.py:100,125): rows of segment s are
 * [offsets[s], offsets[s+1]).  y [nseg][c]; argmax int32 [nseg][c] (row index, -1 and y=0
 * for an empty segment — torch_scatter's fill). */
int irx_segment_max(const float* x, const int32_t* offsets, int nseg, int c, float* y,
                    int32_t* argmax, void* stream);
/* dx must be zero-filled by the caller: dx[argmax[s][j]][j] = dy[s][j]. */
int irx_segment_max_backward(const float* dy, const int32_t* argmax, int nseg, int c,
                             float* dx, void* stream);
/* Mean over equal-length segments (per-instance mean of the (1024, C0) instance points,
 * models/relation_module.py:67-68): x [nseg][len][c] -> y [nseg][c], fp32 in, fp64 accumulate. */
int irx_segment_mean(const float* x, int nseg, int len, int c, float* y, void* stream);

/* offsets[b] = first row whose batch index >= b, for b in [0, nseg]  (rows sorted by batch). */
int irx_batch_offsets(const int32_t* coords, int n, int nseg, int32_t* offsets, void* stream);

/* ---- per-sample input pipeline on a resident scan (reference lib/dataset.py:124,154-181,207-232;
 *      SURVEY.md 8(f) rank 1). elem_bytes = 4 (float32, the dtype of ScanNet's *_aligned_vert.npy) or 8. ------ */

/* random_sampling + augmentation (lib/dataset.py:124,154-181, _translate :440-453): dst[r] = src[choices[r]] with
 * the xyz columns flipped (x and/or y negated), rotated by n_rot (0..3) row-major 3x3 float64 matrices applied in
 * order (new = R p == numpy dot(p, R.T), each result rounded to the storage type like the in-place numpy assignment),
 * then shifted (shift may be NULL). The caller draws choices / angles / shift from the reference's RNG streams. */
int irx_scene_sample(const void* src, int n_src, int c, const int32_t* choices, int n, int flip_x, int flip_y,
                     const double* rot, int n_rot, const double* shift, void* dst, int elem_bytes, void* stream);
/* irx_scene_sample for a whole batch in one call (host arrays of per-sample arguments; choices are int64 device arrays,
 * e.g. torch.randperm; flip_xy = 2 ints, rot = 27 doubles, shift = 3 doubles per sample): dst [n_samples][n][c]. When
 * gslot != NULL it also gathers the sampled points' labels: gslot[b][r] = slot_src[b][choices[b][r]] + slot_base[b],
 * sem[b][r] = sem_src[b][choices[b][r]] (int64, ready for bincount / sort on the device). */
int irx_scene_sample_batch(int n_samples, const void* const* src, const int* n_src, int c, const int64_t* const* choices,
                           int n, const int* flip_xy, const double* rot, const int* n_rot, const double* shift,
                           const int* has_shift, void* dst, const int32_t* const* slot_src, const int32_t* const* sem_src,
                           const int64_t* slot_base, int64_t* gslot, int64_t* sem, int elem_bytes, void* stream);
/* Counter-based random draws for the fully device-side mode (no generator state, no sort). irx_random_subset:
 * out[b][r] (int64, [n_samples <= 64][n]) = a size-n subset of [0, n_src[b]) WITHOUT replacement when n_src[b] >= n —
 * the first n values of a keyed pseudo-random permutation (4-round Feistel network, cycle-walked) — else n hashed
 * uniform draws with replacement: the two cases of random_sampling (utils/pc_utils.py:32-40), same distribution family,
 * not numpy's stream. irx_resample_rows: rows[i][s] = order[seg[i] + j] with j drawn the same way among the
 * seg[i+1]-seg[i] rows of slot i (distinct when the slot has >= n_sample rows); an empty slot gets 0. */
int irx_random_subset(int n_samples, const int* n_src, int n, const uint64_t* seeds, int64_t* out, void* stream);
int irx_resample_rows(const int32_t* order, const int32_t* seg, int n_slots, int n_sample, uint64_t seed, int32_t* rows,
                      void* stream);
/* The instance loop of lib/dataset.py:207-232 for one sampled cloud pts [n][c]: instance i owns the rows
 * order[seg[i] .. seg[i+1]) (ascending point index = np.nonzero(labels == id)); obbs[i] = (0.5*(lo+hi), hi-lo, 0)
 * computed in the storage type and widened to float64; inst_points[i][s] = pts[rows[i][s]] (the 1024-point resample,
 * rows index the sampled cloud); extent (optional, 6 storage-type values) = min xyz, max xyz of the whole cloud
 * (point_min / point_max, lib/dataset.py:267-268). An empty segment gets an all-zero box. order / seg / rows are device
 * arrays, so the whole batch can go through ONE call (pts = all sampled clouds back to back, global row indices). */
int irx_instance_split(const void* pts, int n, int c, const int32_t* order, const int32_t* seg, int n_inst,
                       const int32_t* rows, int n_sample, void* inst_points, double* obbs, void* extent,
                       int elem_bytes, void* stream);

/* ---- language encoder recurrence (nn.GRU on a packed sequence, reference models/lang_module.py:22-28,53-57) -- */

/* One GRU layer, ndir (1|2) directions, hidden size H in {64,128}: the sequential part only.
 * gi [B][T][ndir][3H] = x W_ih^T + b_ih (gate order r,z,n), lengths int32 [B], w_hh [ndir][3H][H],
 * b_hh [ndir][3H]. out [B][T][ndir*H] (zero at t >= length, reverse direction starts at t = length-1);
 * gates [B][T][ndir][4H] = (r, z, n, W_hn h + b_hn) saved for the backward pass. */
int irx_gru_forward(const float* gi, const int32_t* lengths, const float* w_hh, const float* b_hh,
                    int B, int T, int ndir, int H, float* out, float* gates, void* stream);
/* BPTT of the recurrence: dout [B][T][ndir*H] -> dgi, dgh [B][T][ndir][3H] (gradients of the input /
 * hidden projections, biases included); dW_ih, dW_hh, dX are GEMMs of these in the host layer. */
int irx_gru_backward(const float* dout, const float* out, const float* gates, const int32_t* lengths,
                     const float* w_hh, int B, int T, int ndir, int H, float* dgi, float* dgh,
                     void* stream);

/* The four weight gradients of a GRU layer from irx_gru_backward's dgi / dgh in one launch (per direction d: dW_ih[d] [3H][I] =
 * dgi_d^T x, dW_hh[d] [3H][H] = dgh_d^T h_prev_d, db_ih[d] = column sums of dgi_d, db_hh[d] of dgh_d; x [B*T][I] the layer's input,
 * out [B][T][ndir*H] its output — h_prev is read from it with the direction's shift). The *1 pointers are the reverse direction's
 * (NULL when ndir = 1). fp32 FMA tiles, deterministic. */
int irx_gru_wgrad(const float* dgi, const float* dgh, const float* x, const float* out, int B, int T, int I, int ndir, int H,
                  float* dw_ih0, float* dw_ih1, float* dw_hh0, float* dw_hh1, float* db_ih0, float* db_ih1, float* db_hh0,
                  float* db_hh1, void* stream);

/* DynamicEdgeConv of the relation module (models/basic_blocks.py:98-133: MessagePassing(aggr='max') with
 * message = mlp([x_i, weight([pos_j - pos_i, cls_i, cls_j]), x_j])) as one launch per direction over a fixed
 * (n_query, k) neighbour grid (irx_knn_batched's output, -1 = no neighbour). feats [S][fin], pos [S][3]: support rows;
 * qidx int64 [nq]: the support row of every query; cls = the last nc channels of a feature row.
 * params[8] = weight.0.weight [hid][3+2nc], weight.0.bias, weight.2.weight [fin][hid], weight.2.bias, mlp.0.weight
 * [fout][3 fin], mlp.0.bias, mlp.2.weight [fout][fout], mlp.2.bias (nn.Linear layouts, state-dict order).
 * fwd: out [nq][fout] = max over the valid neighbours, arg int32 [nq][fout] = the winning neighbour slot (first maximum).
 * bwd: grads[8] (shapes of params) from dout [nq][fout] + arg; dmin_out (optional): [nq][k][3 fin] = d(message input) per
 * edge, thirds (d x_i, d edge weight, d x_j), followed by [nq][k][3 + 2 nc] = d(edge-weight input) per edge, for callers
 * that need the gradient of the node features. k <= 16, hid <= 128.
 * Deterministic (per-workgroup slabs in the workspace, summed in order). */
size_t irx_edgeconv_workspace_bytes(int nq, int k, int fin, int nc, int hid, int fout);
int irx_edgeconv_max_fwd(const float* feats, const float* pos, const int64_t* qidx, const int32_t* nbr, int nq, int k,
                         int fin, int nc, int hid, int fout, const float* const* params, float* out, int32_t* arg,
                         void* stream);
int irx_edgeconv_max_bwd(const float* feats, const float* pos, const int64_t* qidx, const int32_t* nbr, int nq, int k,
                         int fin, int nc, int hid, int fout, const float* const* params, const float* dout,
                         const int32_t* arg, float* const* grads, float* dmin_out, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ---- box labels of get_loss / get_eval --------------------------------------- */

/* lib/loss_helper.py:233-258 without the per-sample host loop: IoU (float64, axis-aligned: utils/box_util.py:154-175,
 * 310-333 with heading 0) of every same-class candidate box against its scene's ground-truth box, and the one-hot label
 * of the best candidate (np.argmax: first maximum). obbs [S][7] double = (centre, size, heading) of every instance;
 * filtered int64 [total] = rows of obbs, scene-major; starts int64 [n_scenes + 1]; gt_obb [n_scenes][7] double.
 * Outputs: labels float [total] (cluster_label of every scene); for scenes with scored_pos[i] >= 0 (>= 2 candidates)
 * the same labels at lab[scored_pos[i] + j] (aligned with the score vectors) and keep[scored_row[i]] = (max IoU >= 0.2);
 * best_iou double [n_scenes] (optional). Bit-identical to the numpy evaluation. scored_pos / scored_row may be NULL. */
int irx_iou_labels(const double* obbs, const int64_t* filtered, const int64_t* starts, const double* gt_obb,
                   int n_scenes, const int64_t* scored_pos, const int64_t* scored_row, float* labels, float* lab,
                   float* keep, double* best_iou, void* stream);

/* lib/eval_helper.py:52-100 on the device: per scene the arg-max of the summed scores (s1 + s2) + s3 over its
 * candidates (scenes with >= 2), the arg-max of its labels, the chosen box (the only candidate / a zero box for 1 / 0
 * candidates) and its IoU against the ground-truth box. out [n_scenes][10] double = (pred, tgt, iou, chosen obb[7]);
 * pred = tgt = -1 for scenes with < 2 candidates. */
int irx_eval_select(const float* s1, const float* s2, const float* s3, const float* labels, const double* obbs,
                    const int64_t* filtered, const int64_t* starts, const int64_t* scored_pos, const double* gt_obb,
                    int n_scenes, double* out, void* stream);

/* ---- multiview back-projection (lib/projection.py:191-279) ----------------------------- */

/* ProjectionHelper.compute_projection for one frame: which of the n_points world points (float [n][3]) fall on which
 * pixel of the depth frame (float [height][width]): inside the viewing frustum (six rounded plane tests), projected by
 * world_to_camera + intrinsics onto a pixel inside the image whose depth lies in [depth_min, depth_max] and within
 * `accuracy` of the point's camera depth. ind3d / ind2d: int64 [n_points + 1], the reference's format — [0] = number of
 * correspondences m (it stays on the device: no host sync), [1 .. m] = point indices ascending / their pixel indices
 * (y * width + x), zeros behind. params: HOST array of IRX_PROJ_NPARAMS floats = 6 inward plane normals [6][3],
 * corner_coords[2][:3], corner_coords[4][:3], world_to_camera [4][4] row-major, fx, fy, cx, cy, depth_min, depth_max,
 * accuracy (the 8-corner algebra of projection.py:49-122 is tiny host work). */
#define IRX_PROJ_NPARAMS 47
size_t irx_project_workspace_bytes(int n_points);
int irx_project_points(const float* points, int n_points, const float* depth, int width, int height,
                       const float* params, int64_t* ind3d, int64_t* ind2d, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ProjectionHelper.project / Projection.forward (lib/projection.py:255-279,285-305): out float [channels][n_points] =
 * label[:, ind2d[1..m]] scattered to columns ind3d[1..m], zero elsewhere; m is read on the device. */
int irx_project_features(const float* label, int channels, int n_pixels, const int64_t* ind3d, const int64_t* ind2d,
                         int n_points, float* out, void* stream);

/* ---- optimizer -------------------------------------------------------------------------------------- */

/* torch.optim.Adam(lr, betas, eps, weight_decay) (reference scripts/train.py:121) as ONE launch over flat,
 * 16-byte aligned fp32 buffers of n elements; `step` counts from 1; gradients are multiplied by grad_scale
 * first (1/world_size after a sum all-reduce). */
int irx_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                  void* stream);

/* ---- instance graph ------------------------------------------------------------------- */

/* torch_cluster.knn(x=support, y=query, k, batch_x, batch_y) (models/basic_blocks.py:120):
 * for every query the k nearest support rows with the same batch id (Euclidean, fp32; ties
 * -> lower support index). support rows of one batch id are contiguous:
 * sup_offsets int32 [nbatch+1]. nbr_idx int32 [nq][k] (-1 padded when a batch item has < k
 * support rows), ascending distance. */
int irx_knn_batched(const float* sup_xyz, const int32_t* sup_offsets, const float* qry_xyz,
                    const int32_t* qry_batch, int nq, int k, int32_t* nbr_idx, void* stream);

/* The two-layer head MLPs  y = W2 . D(relu(N(W1 x + b1))) + b2  (reference models/attribute_module.py:26-34, relation_module.py:18-27,
 * scene_module.py:38-42: nn.Sequential(Linear, BatchNorm1d | LayerNorm, ReLU, [Dropout], Linear)) as one operator each way.
 * x [rows][din], w1 [dh][din], w2 [dout][dh] (nn.Linear layouts); norm: 1 = BatchNorm1d with batch statistics (running_mean /
 * running_var, when given, are updated with `momentum` and the unbiased variance), 2 = BatchNorm1d with the running statistics,
 * 3 = LayerNorm over dh, 4 = none (gamma / beta may be NULL: the word projection of models/lang_module.py:22-23, Linear -> ReLU ->
 * Dropout -> Linear -> ReLU), + 8: ReLU on the output y (the backward then expects dy already masked by y > 0);
 * drop_p > 0: dropout of the hidden activations, decided by a counter-based hash of (seed, element).
 * saved: irx_mlp2_saved_floats(rows, dh) floats the backward reads back (h, a, statistics). Backward: dhid = scratch
 * [rows][dh]; dx may be NULL; drop_scale = 1 / (1 - drop_p). fp32 FMA tiles, deterministic, any row count. */
size_t irx_mlp2_saved_floats(int rows, int dh);
int irx_mlp2_fwd(const float* x, int rows, int din, int dh, int dout, const float* w1, const float* b1, int norm,
                 const float* gamma, const float* beta, float eps, float* running_mean, float* running_var, float momentum,
                 float drop_p, unsigned long long seed, const float* w2, const float* b2, float* saved, float* y, void* stream);
int irx_mlp2_bwd(const float* x, const float* dy, int rows, int din, int dh, int dout, const float* w1, int norm,
                 const float* gamma, const float* w2, const float* saved, float drop_scale, float* dhid, float* dx, float* dw1,
                 float* db1, float* dgamma, float* dbeta, float* dw2, float* db2, void* stream);

/* nn.Dropout in training mode (models/scene_module.py:31, between the two Conv2d of vis_emb_fc) without a mask tensor: element o of
 * the flat tensor x [n] is kept iff a counter-based hash of (seed, o) falls at or above p (the decision irx_mlp2_fwd uses), kept
 * elements are scaled by 1 / (1 - p). The backward is the same call on the incoming gradient with the same seed. x == y allowed. */
int irx_dropout_flat(const float* x, size_t n, float p, unsigned long long seed, float* y, void* stream);

/* ---- language-instance matching scores ---------------------------------------------------
 * models/attribute_module.py:122-126 (normalize + normalize + row dot), relation_module.py:104-105 and
 * scene_module.py:104-106 (cosine_similarity): score[i] = <a_i, b_j> / (max(|a_i|, eps) * max(|b_j|, eps)) with
 * j = idx[i] (idx NULL = identity), a [n][d], b [m][d]; norms [n][2] = (|a_i|, |b_j|) is kept for the backward.
 * Backward: da [n][d] and db [m][d] (either may be NULL); db_j sums its candidates in ascending i (deterministic). */
int irx_cosine_rows_fwd(const float* a, const float* b, const int64_t* idx, int n, int d, float eps, float* score,
                        float* norms, void* stream);
int irx_cosine_rows_bwd(const float* a, const float* b, const int64_t* idx, const float* score, const float* norms,
                        const float* dscore, int n, int m, int d, float eps, float* da, float* db, void* stream);

/* Batched ContrastiveLoss of the matching step (lib/loss_helper.py:93-107,248-258): for every scored scene s = rows
 * [seg_off[s], seg_off[s+1]) of the candidate list, x = gamma*(s1+s2+s3), loss_s = keep_s * max(logsumexp(x*(1-lab)) -
 * sum(x*lab) + margin, 0); out[0] = sum_s loss_s (fixed order). per / act / lse [nseg] are scratch kept for the backward,
 * which writes ds [n] = d out / d s1 = d out / d s2 = d out / d s3 (scaled by dout[0]). */
int irx_contrastive_fwd(const float* s1, const float* s2, const float* s3, const float* lab, const int64_t* seg_off,
                        const float* keep, int nseg, float gamma, float margin, float* out, float* per, float* act,
                        float* lse, void* stream);
int irx_contrastive_bwd(const float* s1, const float* s2, const float* s3, const float* lab, const int64_t* seg_off,
                        const float* act, const float* lse, const float* dout, int nseg, float gamma, float* ds,
                        void* stream);

/* The whole training loss of get_loss (lib/loss_helper.py:196-269) in one launch, gradients included: lang_loss = cross-entropy of
 * lang_scores [B][n_lang] (:110-118), seg_loss / seg_acc = cross-entropy / accuracy of seg_scores [B][n_seg] (:121-150), ref_loss =
 * the batched ContrastiveLoss above / batch_size (:248-260), loss = ref_weight * ref_loss + lang_loss + seg_loss (:263).
 * out[5] = loss, ref_loss, lang_loss, seg_loss, seg_acc;  d_lang [B][n_lang], d_seg [B][n_seg], d_s [seg_off[nscored]] =
 * d loss / d input (d_s is the gradient of EACH of s1, s2, s3) for an upstream gradient of 1. nscored == 0: no scored scene
 * (ref_loss = 0, the score pointers may be NULL). */
int irx_total_loss(const float* lang_scores, const int64_t* lang_label, int B, int n_lang, const float* seg_scores,
                   const int64_t* seg_label, int n_seg, const float* s1, const float* s2, const float* s3,
                   const float* lab, const int64_t* seg_off, const float* keep, int nscored, float gamma,
                   float margin, float ref_weight, int batch_size, float* out, float* d_lang, float* d_seg, float* d_s,
                   void* stream);

/* Language-guided attention pooling of the scene head (models/scene_module.py:84-93): logit[b][i] = <feats[b][i], lang[b]> * scale
 * (scale = 1 / sqrt(d)), atten[b] = softmax over the n cells, out[b] = sum_i atten[b][i] feats[b][i]. feats [B][n][d], lang [B][d],
 * atten [B][n], out [B][d]. Backward: dfeats [B][n][d], dlang [B][d] from dout [B][d] and (optional, may be NULL) datten [B][n].
 * One workgroup per scene, deterministic. */
int irx_attn_pool_fwd(const float* feats, const float* lang, int B, int n, int d, float scale, float* atten, float* out,
                      void* stream);
int irx_attn_pool_bwd(const float* feats, const float* lang, const float* atten, const float* dout, const float* datten,
                      int B, int n, int d, float scale, float* dfeats, float* dlang, void* stream);

/* The four attention heads of the language module (models/lang_module.py:61-83) in one launch each way: logit[b][t][h] =
 * <feats[b][t], w[h]> + bias[h][0]; p = softmax over ALL T positions; q = p * [t < len[b]]; att = q / sum_t q; pooled[b][h] =
 * sum_t att[b][t][h] embed[b][t]. feats [B][T][O], embed [B][T][E], len [B] int64, w / bias: 4 pointers each (the fc_a, fc_cls,
 * fc_rel, fc_scene parameters in the caller's order). Outputs att [B][T][4], pooled [B][4][E]; prob [B][T][4] and qsum [B][4] are
 * kept for the backward, which returns dfeats [B][T][O], dembed [B][T][E], dw [4][O], db [4] from dpooled [B][4][E] and (optional)
 * datt [B][T][4]; part: scratch of B * 4 * (O + 1) floats. Deterministic. */
int irx_lang_pool_fwd(const float* feats, const float* embed, const int64_t* len, int B, int T, int O, int E,
                      const float* const* w, const float* const* bias, float* att, float* prob, float* qsum, float* pooled,
                      void* stream);
int irx_lang_pool_bwd(const float* feats, const float* embed, const int64_t* len, int B, int T, int O, int E,
                      const float* const* w, const float* att, const float* prob, const float* qsum, const float* dpooled,
                      const float* datt, float* dfeats, float* dembed, float* dw, float* db, float* part, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IRX_H_ */
