// irx_project.hip — multiview back-projection (reference lib/projection.py:191-279: ProjectionHelper.compute_projection
// + project): which points of a cloud fall on which pixel of a depth frame, and the gather / scatter
// of image features onto those points.  The reference runs ~25 torch ops, 3 boolean-mask compactions and 3 host syncs
// per frame; here one frame is two launches (flags + per-tile counts, ordered compaction) and the feature transfer one,
// with the correspondence count staying on the device.
//
// Float32 arithmetic follows the reference's torch ops one for one (same operation order, explicit fused
// multiply-adds where the BLAS kernels behind torch.mm fuse, round-half-even like torch.round), so the index lists are
// the reference's:   frustum:  round(<p - c, n_k> * 100) / 100 < 0  for the six inward plane normals (projection.py:
// 141-147; dot = fma(y, n.y, x * n.x) + z * n.z), camera = world_to_camera . (x, y, z, 1) as an fma chain in column
// order, pixel = round(cam.xy * f / cam.z + c), then the image-range, depth-range and |depth - cam.z| <= accuracy tests.
#include "irx_common.h"

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

#define PJ_TILE 256

struct IrxProjParams {
  float normal[6][3];     // inward plane normals (compute_frustum_normals)
  float c2[3], c4[3];     // corner_coords[2], corner_coords[4]: reference points of planes 0-2 / 3-5
  float w2c[4][4];        // torch.inverse(camera_to_world)
  float fx, fy, cx, cy;
  int width, height;
  float depth_min, depth_max, accuracy;
};

__device__ __forceinline__ bool pj_point(const IrxProjParams& P, const float* __restrict__ pts,
                                         const float* __restrict__ depth, int i, int& pix) {
  const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
  bool keep = true;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float* c = (k < 3) ? P.c2 : P.c4;
    const float dx = __fsub_rn(x, c[0]), dy = __fsub_rn(y, c[1]), dz = __fsub_rn(z, c[2]);
    const float d = __fadd_rn(__fmaf_rn(dy, P.normal[k][1], __fmul_rn(dx, P.normal[k][0])), __fmul_rn(dz, P.normal[k][2]));
    keep = keep && (__fdiv_rn(rintf(__fmul_rn(d, 100.f)), 100.f) < 0.f);
  }
  if (!keep) return false;
  float cam[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float a = __fmul_rn(P.w2c[r][0], x);
    a = __fmaf_rn(P.w2c[r][1], y, a);
    a = __fmaf_rn(P.w2c[r][2], z, a);
    cam[r] = __fmaf_rn(P.w2c[r][3], 1.f, a);
  }
  const float u = __fadd_rn(__fdiv_rn(__fmul_rn(cam[0], P.fx), cam[2]), P.cx);
  const float v = __fadd_rn(__fdiv_rn(__fmul_rn(cam[1], P.fy), cam[2]), P.cy);
  const float ru = rintf(u), rv = rintf(v);                  // torch.round: half to even
  if (!(ru >= 0.f && rv >= 0.f && ru < (float)P.width && rv < (float)P.height)) return false;
  pix = (int)rv * P.width + (int)ru;
  const float dv = depth[pix];
  return dv >= P.depth_min && dv <= P.depth_max && fabsf(__fsub_rn(dv, cam[2])) <= P.accuracy;
}

// pass 1: per-point flag + pixel, per-tile count
__global__ __launch_bounds__(PJ_TILE) void k_project_flags(IrxProjParams P, const float* __restrict__ pts,
                                                           const float* __restrict__ depth, int n,
                                                           int32_t* __restrict__ pix_of, int32_t* __restrict__ tile_count) {
  const int i = blockIdx.x * PJ_TILE + threadIdx.x;
  int pix = -1;
  bool ok = false;
  if (i < n) ok = pj_point(P, pts, depth, i, pix);
  if (i < n) pix_of[i] = ok ? pix : -1;
  const int c = __syncthreads_count(ok);
  if (threadIdx.x == 0) tile_count[blockIdx.x] = c;
}

// pass 2: ordered compaction (ascending point index, as the reference's boolean-mask indexing)
__global__ __launch_bounds__(PJ_TILE) void k_project_write(const int32_t* __restrict__ pix_of,
                                                           const int32_t* __restrict__ tile_count, int n, int ntiles,
                                                           int64_t* __restrict__ ind3d, int64_t* __restrict__ ind2d) {
  __shared__ int s_part[PJ_TILE];
  __shared__ int s_wave[PJ_TILE / 64];
  int acc = 0;
  for (int t = threadIdx.x; t < (int)blockIdx.x; t += PJ_TILE) acc += tile_count[t];
  s_part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = PJ_TILE / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) s_part[threadIdx.x] += s_part[threadIdx.x + s];
    __syncthreads();
  }
  const int base = s_part[0];
  const int i = blockIdx.x * PJ_TILE + threadIdx.x;
  const int pix = (i < n) ? pix_of[i] : -1;
  const bool ok = pix >= 0;
  const unsigned long long b = __ballot(ok);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_wave[wave] = __popcll(b);
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += s_wave[w];
  if (ok) {
    const int dst = base + woff + __popcll(b & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
    ind3d[1 + dst] = i;
    ind2d[1 + dst] = pix;
  }
  if (blockIdx.x == (unsigned)ntiles - 1 && threadIdx.x == 0) {
    int total = base;
    for (int w = 0; w < PJ_TILE / 64; ++w) total += s_wave[w];
    ind3d[0] = total;
    ind2d[0] = total;
  }
}

extern "C" size_t irx_project_workspace_bytes(int n_points) {
  if (n_points <= 0) return 0;
  return ((size_t)n_points + (size_t)irx_cdiv(n_points, PJ_TILE)) * sizeof(int32_t);
}

extern "C" int irx_project_points(const float* points, int n_points, const float* depth, int width, int height,
                                  const float* params, int64_t* ind3d, int64_t* ind2d, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  IRX_REQUIRE(n_points >= 0 && width > 0 && height > 0 && ind3d && ind2d && params, "irx_project_points: bad arguments");
  IRX_CHECK_HIP(hipMemsetAsync(ind3d, 0, ((size_t)n_points + 1) * sizeof(int64_t), S(stream)), "irx_project_points(memset)");
  IRX_CHECK_HIP(hipMemsetAsync(ind2d, 0, ((size_t)n_points + 1) * sizeof(int64_t), S(stream)), "irx_project_points(memset)");
  if (n_points == 0) return IRX_OK;
  IRX_REQUIRE(points && depth, "irx_project_points: null pointer");
  if (!workspace || workspace_bytes < irx_project_workspace_bytes(n_points)) {
    irx_set_error("irx_project_points: workspace %zu < %zu", workspace_bytes, irx_project_workspace_bytes(n_points));
    return IRX_ERR_WORKSPACE;
  }
  IrxProjParams P;
  const float* q = params;               // host array of IRX_PROJ_NPARAMS floats, layout of include/irx.h
  for (int k = 0; k < 6; ++k) for (int d = 0; d < 3; ++d) P.normal[k][d] = *q++;
  for (int d = 0; d < 3; ++d) P.c2[d] = *q++;
  for (int d = 0; d < 3; ++d) P.c4[d] = *q++;
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) P.w2c[r][c] = *q++;
  P.fx = *q++; P.fy = *q++; P.cx = *q++; P.cy = *q++;
  P.depth_min = *q++; P.depth_max = *q++; P.accuracy = *q++;
  P.width = width; P.height = height;
  const int ntiles = irx_cdiv(n_points, PJ_TILE);
  int32_t* pix_of = (int32_t*)workspace;
  int32_t* tile_count = pix_of + n_points;
  k_project_flags<<<ntiles, PJ_TILE, 0, S(stream)>>>(P, points, depth, n_points, pix_of, tile_count);
  IRX_CHECK_LAUNCH("irx_project_points(flags)");
  k_project_write<<<ntiles, PJ_TILE, 0, S(stream)>>>(pix_of, tile_count, n_points, ntiles, ind3d, ind2d);
  IRX_CHECK_LAUNCH("irx_project_points(write)");
  return IRX_OK;
}

// out[c][ind3d[1 + j]] = label[c][ind2d[1 + j]], j < ind3d[0] (read on the device); out is zero elsewhere.
__global__ void k_project_features(const float* __restrict__ label, int channels, int n_pixels,
                                   const int64_t* __restrict__ ind3d, const int64_t* __restrict__ ind2d, int n_points,
                                   float* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = (int)ind3d[0];
  if (j >= m) return;
  const size_t p = (size_t)ind3d[1 + j], q = (size_t)ind2d[1 + j];
  for (int c = blockIdx.y; c < channels; c += gridDim.y) out[(size_t)c * n_points + p] = label[(size_t)c * n_pixels + q];
}

extern "C" int irx_project_features(const float* label, int channels, int n_pixels, const int64_t* ind3d,
                                    const int64_t* ind2d, int n_points, float* out, void* stream) {
  IRX_REQUIRE(channels >= 1 && n_pixels >= 1 && n_points >= 0 && out, "irx_project_features: bad arguments");
  IRX_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)channels * n_points * sizeof(float), S(stream)), "irx_project_features(memset)");
  if (n_points == 0) return IRX_OK;
  IRX_REQUIRE(label && ind3d && ind2d, "irx_project_features: null pointer");
  dim3 grid(irx_cdiv(n_points, 256), channels < 64 ? channels : 64);
  k_project_features<<<grid, 256, 0, S(stream)>>>(label, channels, n_pixels, ind3d, ind2d, n_points, out);
  IRX_CHECK_LAUNCH("irx_project_features");
  return IRX_OK;
}
