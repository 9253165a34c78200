"""TEST ORACLE — numpy restatement of the reference's lib/projection.py:191-279 (ProjectionHelper.compute_projection,
project) with every float32 operation written out in the order the reference's torch ops perform it (pinned bit for bit
to tests/golden/projection.npz = the reference's own output, tests/test_oracle_cpu.py). Only tests may import this."""
import numpy as np

f32 = np.float32


def _fma(a, b, c):
    """float32 fused multiply-add (one rounding): exact product in float64, one add, one rounding to float32."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def compute_projection(points, depth, params, width, height):
    """points (N,3) f32; depth (H,W) f32; params = the 47 floats of include/irx.h (normals, c2, c4, w2c, fx, fy, cx, cy,
    depth_min, depth_max, accuracy). -> (ind3d, ind2d) int64 arrays of the valid correspondences, ascending point order."""
    p = np.asarray(params, f32)
    normals, c2, c4, w2c = p[:18].reshape(6, 3), p[18:21], p[21:24], p[24:40].reshape(4, 4)
    fx, fy, cx, cy, dmin, dmax, acc = p[40:47]
    pts = np.asarray(points, f32)
    keep = np.ones(len(pts), bool)
    for k in range(6):                                           # projection.py:141-154
        d = pts - (c2 if k < 3 else c4)
        dot = _fma(d[:, 1], normals[k, 1], d[:, 0] * normals[k, 0]) + d[:, 2] * normals[k, 2]
        keep &= (np.rint(dot * f32(100)) / f32(100)) < 0
    idx = np.nonzero(keep)[0]
    x, y, z = pts[idx, 0], pts[idx, 1], pts[idx, 2]
    cam = []
    for r in range(3):                                           # projection.py:218 (torch.mm as an fma chain)
        a = w2c[r, 0] * x
        a = _fma(w2c[r, 1], y, a)
        a = _fma(w2c[r, 2], z, a)
        cam.append(_fma(w2c[r, 3], f32(1), a))
    with np.errstate(divide="ignore", invalid="ignore"):
        u = np.rint((cam[0] * fx) / cam[2] + cx)                 # projection.py:221-223 (torch.round: half to even)
        v = np.rint((cam[1] * fy) / cam[2] + cy)
    ok = (u >= 0) & (v >= 0) & (u < width) & (v < height)
    idx, u, v, cz = idx[ok], u[ok].astype(np.int64), v[ok].astype(np.int64), cam[2][ok]
    pix = v * width + u
    dv = np.asarray(depth, f32).reshape(-1)[pix]
    m = (dv >= dmin) & (dv <= dmax) & (np.abs(dv - cz) <= acc)   # projection.py:235-236
    return idx[m].astype(np.int64), pix[m].astype(np.int64)


def project(label, ind3d, ind2d, num_points):
    """label (C,H,W) -> (C, num_points): label[:, ind2d] scattered to columns ind3d (projection.py:255-279)."""
    c = label.shape[0]
    out = np.zeros((c, num_points), f32)
    out[:, ind3d] = label.reshape(c, -1)[:, ind2d]
    return out
