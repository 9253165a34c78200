// irx_labels.hip — the box arithmetic of get_loss / get_eval on the device:
//   irx_iou_labels : IoU of every same-class candidate box against its scene's ground-truth box and the one-hot
//                    "cluster label" of the best one (reference lib/loss_helper.py:233-258: get_3d_box_batch +
//                    box3d_iou_batch from utils/box_util.py:154-175,310-333, np.argmax, the IoU >= 0.2 gate);
//   irx_eval_select: per scene the candidate with the highest summed score, the label arg-max, the chosen box and its
//                    IoU (reference lib/eval_helper.py:52-100).
// ScanNet boxes are axis-aligned (heading 0), so get_3d_box's rotation is the identity and the 8 corners are
// centre +- size/2 exactly; every operation below is the float64 operation numpy performs, in numpy's order
// (products left to right, (v1 + v2 - inter) + 1e-8), so IoUs and labels are BIT-IDENTICAL to the host code.
// One thread per scene: B <= a few hundred scenes of <= a few dozen boxes — launch latency is the whole cost.
#include "irx_common.h"

// Built with -ffp-contract=off (instancerefer_amd/_build.py EXTRA_FLAGS): every product must be rounded before it is
// added, as numpy does — with the library-wide -ffp-contract=fast the backend turns `v1 + v2 - inter` into
// fma(-a, e, v1 + v2) and the IoU loses bit-equality (a file-scope `#pragma clang fp contract(off)` does not prevent it).

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

// obb = (cx, cy, cz, sx, sy, sz, heading)
__device__ static inline double iou_aabb(const double* __restrict__ p, const double* __restrict__ g) {
  double inter = 1.0, v1 = 1.0, v2 = 1.0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const double h1 = p[3 + d] / 2, h2 = g[3 + d] / 2;
    const double mn1 = -h1 + p[d], mx1 = h1 + p[d];
    const double mn2 = -h2 + g[d], mx2 = h2 + g[d];
    const double lo = fmax(mn1, mn2), hi = fmin(mx1, mx2);
    const double e = fmax(hi - lo, 0.0);
    // left-to-right products: ((a * b) * c) with a leading exact "1.0 *"
    inter = (d == 0) ? e : inter * e;
    v1 = (d == 0) ? (mx1 - mn1) : v1 * (mx1 - mn1);
    v2 = (d == 0) ? (mx2 - mn2) : v2 * (mx2 - mn2);
  }
  return inter / (v1 + v2 - inter + 1e-8);
}

// filtered[starts[i] .. starts[i+1]) = rows of `obbs` that are scene i's same-class candidates (instance order).
// labels[starts[i] + j] = 1 for the first candidate with the highest IoU, else 0        (cluster_label, all scenes)
// scored_pos[i] >= 0: scene i has >= 2 candidates and its labels are also written to lab[scored_pos[i] + j];
// scored_row[i] >= 0: keep[scored_row[i]] = (max IoU >= 0.2)                               (loss_helper.py:249)
__global__ void k_iou_labels(const double* __restrict__ obbs, const int64_t* __restrict__ filtered,
                             const int64_t* __restrict__ starts, const double* __restrict__ gt, int nscene,
                             const int64_t* __restrict__ scored_pos, const int64_t* __restrict__ scored_row,
                             float* __restrict__ labels, float* __restrict__ lab, float* __restrict__ keep,
                             double* __restrict__ best_iou) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nscene) return;
  const int lo = (int)starts[i], hi = (int)starts[i + 1];
  double best = -1.0;
  int arg = -1;
  for (int j = lo; j < hi; ++j) {
    const double v = iou_aabb(obbs + 7 * filtered[j], gt + 7 * (size_t)i);
    if (arg < 0 || v > best) {            // np.argmax: first maximum
      best = v;
      arg = j;
    }
  }
  const long sp = scored_pos ? (long)scored_pos[i] : -1;
  for (int j = lo; j < hi; ++j) {
    const float l = (j == arg) ? 1.f : 0.f;
    labels[j] = l;
    if (sp >= 0) lab[sp + (j - lo)] = l;
  }
  if (scored_row && scored_row[i] >= 0) keep[scored_row[i]] = (best >= 0.2) ? 1.f : 0.f;
  if (best_iou) best_iou[i] = (hi > lo) ? best : 0.0;
}

extern "C" int irx_iou_labels(const double* obbs, const int64_t* filtered, const int64_t* starts, const double* gt_obb,
                              int n_scenes, const int64_t* scored_pos, const int64_t* scored_row, float* labels,
                              float* lab, float* keep, double* best_iou, void* stream) {
  IRX_REQUIRE(n_scenes >= 0, "irx_iou_labels: n_scenes < 0");
  if (n_scenes == 0) return IRX_OK;
  IRX_REQUIRE(obbs && filtered && starts && gt_obb && labels, "irx_iou_labels: null pointer");
  IRX_REQUIRE(!scored_pos || lab, "irx_iou_labels: scored_pos without lab");
  IRX_REQUIRE(!scored_row || keep, "irx_iou_labels: scored_row without keep");
  k_iou_labels<<<irx_cdiv(n_scenes, 64), 64, 0, S(stream)>>>(obbs, filtered, starts, gt_obb, n_scenes, scored_pos,
                                                              scored_row, labels, lab, keep, best_iou);
  IRX_CHECK_LAUNCH("irx_iou_labels");
  return IRX_OK;
}

// Per scene i with c = starts[i+1] - starts[i] candidates:
//   c >= 2: pred = argmax_j (s1 + s2) + s3 over its scores [scored_pos[i], +c) (torch.argmax: first maximum),
//           tgt = argmax_j labels; chosen box = candidate pred;      c == 1: the only candidate;      c == 0: zero box.
//   out[i] = (pred, tgt, IoU(chosen, gt), chosen obb[7])  as 10 doubles  — ONE D2H copy serves the whole batch.
__global__ void k_eval_select(const float* __restrict__ s1, const float* __restrict__ s2, const float* __restrict__ s3,
                              const float* __restrict__ labels, const double* __restrict__ obbs,
                              const int64_t* __restrict__ filtered, const int64_t* __restrict__ starts,
                              const int64_t* __restrict__ scored_pos, const double* __restrict__ gt, int nscene,
                              double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nscene) return;
  const int lo = (int)starts[i], hi = (int)starts[i + 1], c = hi - lo;
  int pred = -1, tgt = -1;
  double box[7] = {0, 0, 0, 0, 0, 0, 0};
  if (c >= 2) {
    const long sp = (long)scored_pos[i];
    float bs = 0.f, bl = 0.f;
    for (int j = 0; j < c; ++j) {
      const float s = (s1[sp + j] + s2[sp + j]) + s3[sp + j];
      if (pred < 0 || s > bs) { bs = s; pred = j; }
      const float l = labels[lo + j];
      if (tgt < 0 || l > bl) { bl = l; tgt = j; }
    }
  }
  if (c >= 1) {
    const double* src = obbs + 7 * filtered[lo + (c >= 2 ? pred : 0)];
#pragma unroll
    for (int d = 0; d < 7; ++d) box[d] = src[d];
  }
  double* o = out + 10 * (size_t)i;
  o[0] = (double)pred;
  o[1] = (double)tgt;
  o[2] = iou_aabb(box, gt + 7 * (size_t)i);
#pragma unroll
  for (int d = 0; d < 7; ++d) o[3 + d] = box[d];
}

extern "C" int irx_eval_select(const float* s1, const float* s2, const float* s3, const float* labels, const double* obbs,
                               const int64_t* filtered, const int64_t* starts, const int64_t* scored_pos,
                               const double* gt_obb, int n_scenes, double* out, void* stream) {
  IRX_REQUIRE(n_scenes >= 0, "irx_eval_select: n_scenes < 0");
  if (n_scenes == 0) return IRX_OK;
  IRX_REQUIRE(starts && scored_pos && gt_obb && out, "irx_eval_select: null pointer");
  k_eval_select<<<irx_cdiv(n_scenes, 64), 64, 0, S(stream)>>>(s1, s2, s3, labels, obbs, filtered, starts, scored_pos,
                                                               gt_obb, n_scenes, out);
  IRX_CHECK_LAUNCH("irx_eval_select");
  return IRX_OK;
}
