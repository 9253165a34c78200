"""ProjectionHelper — drop-in for the reference's lib/projection.py:5-279 (multiview back-projection: which scan points
a depth frame sees, and the transfer of per-pixel ENet features onto them; used offline by
scripts/project_multiview_features.py:28-29,96-118,188 to build the 128 extra input channels of the stress config).

Same constructor, same `compute_projection(points, depth, camera_to_world) -> (indices_3d, indices_2d) | None` and
`project(label, lin_indices_3d, lin_indices_2d, num_points)` contracts and index-list format ([0] = count, then the
indices, zero padded to num_points + 1). The per-point work runs in csrc/irx_project.hip (irx_project_points: frustum +
projection + depth tests + ORDERED compaction in two launches; irx_project_features: one gather/scatter launch); the
8-corner frustum algebra and the 4x4 inverse stay tiny host torch ops, evaluated exactly as the reference writes them.
`compute_projection_batch` does all frames of a scan with no host sync at all (the reference syncs 3x per frame)."""
import numpy as np
import torch

from . import _lib


class ProjectionHelper:
    def __init__(self, intrinsic, depth_min, depth_max, image_dims, accuracy, cuda=True):
        self.intrinsic = intrinsic
        self.depth_min = depth_min
        self.depth_max = depth_max
        self.image_dims = image_dims
        self.accuracy = accuracy
        self.cuda = cuda
        self._compute_corner_points()

    # ---- host algebra on 8 corner points (projection.py:17-122), float32 torch on the CPU ----
    def depth_to_skeleton(self, ux, uy, depth):
        x = (ux - self.intrinsic[0][2]) / self.intrinsic[0][0]
        y = (uy - self.intrinsic[1][2]) / self.intrinsic[1][1]
        return torch.Tensor([depth * x, depth * y, depth])

    def skeleton_to_depth(self, p):
        x = (p[0] * self.intrinsic[0][0]) / p[2] + self.intrinsic[0][2]
        y = (p[1] * self.intrinsic[1][1]) / p[2] + self.intrinsic[1][2]
        return torch.Tensor([x, y, p[2]])

    def _compute_corner_points(self):
        w, h = self.image_dims[0] - 1, self.image_dims[1] - 1
        cp = torch.ones(8, 4)
        for i, (ux, uy, d) in enumerate([(0, 0, self.depth_min), (w, 0, self.depth_min), (w, h, self.depth_min),
                                         (0, h, self.depth_min), (0, 0, self.depth_max), (w, 0, self.depth_max),
                                         (w, h, self.depth_max), (0, h, self.depth_max)]):
            cp[i][:3] = self.depth_to_skeleton(ux, uy, d)
        self.corner_points = cp

    def compute_frustum_corners(self, camera_to_world):
        c2w = camera_to_world.detach().float().cpu()
        return torch.bmm(c2w.repeat(8, 1, 1), self.corner_points.unsqueeze(2))

    def compute_frustum_normals(self, corner_coords):
        c = corner_coords
        pairs = [(3, 0, 1, 0), (2, 1, 5, 1), (3, 2, 6, 2), (0, 3, 7, 3), (1, 0, 4, 0), (6, 5, 4, 5)]
        normals = c.new_zeros(6, 3)
        for k, (a, b, d, e) in enumerate(pairs):
            v1 = (c[a][:3] - c[b][:3]).view(-1)
            v2 = (c[d][:3] - c[e][:3]).view(-1)
            normals[k] = torch.linalg.cross(v1, v2)
        return normals

    def _params(self, camera_to_world):
        c2w = camera_to_world.detach().float().cpu()
        corners = self.compute_frustum_corners(c2w)
        normals = self.compute_frustum_normals(corners)
        w2c = torch.inverse(c2w)
        k = self.intrinsic
        tail = torch.tensor([k[0][0], k[1][1], k[0][2], k[1][2], self.depth_min, self.depth_max, self.accuracy],
                            dtype=torch.float32)
        p = torch.cat([normals.reshape(-1), corners[2][:3].reshape(-1), corners[4][:3].reshape(-1), w2c.reshape(-1), tail])
        return np.ascontiguousarray(p.numpy(), dtype=np.float32)

    # ---- device work ----
    def compute_projection_launch(self, points, depth, camera_to_world):
        """-> (indices_3d, indices_2d) int64 [num_points + 1] on the device, count in [0]; no host sync."""
        points = points.float().contiguous()
        depth = depth.float().contiguous()
        n = points.shape[0]
        w, h = int(self.image_dims[0]), int(self.image_dims[1])
        if depth.numel() != w * h:
            raise ValueError("depth map has %d pixels, image_dims say %d x %d" % (depth.numel(), w, h))
        dev = points.device
        i3 = torch.empty(n + 1, dtype=torch.int64, device=dev)
        i2 = torch.empty(n + 1, dtype=torch.int64, device=dev)
        wsb = int(_lib.load().irx_project_workspace_bytes(n))
        ws = torch.empty(max(wsb, 4), dtype=torch.uint8, device=dev)
        params = self._params(camera_to_world)
        _lib.call("irx_project_points", _lib.ptr(points), n, _lib.ptr(depth), w, h, params.ctypes.data, _lib.ptr(i3),
                  _lib.ptr(i2), _lib.ptr(ws), wsb, _lib.stream_ptr())
        return i3, i2

    def compute_projection(self, points, depth, camera_to_world):
        i3, i2 = self.compute_projection_launch(points, depth, camera_to_world)
        if int(i3[0]) == 0:              # the reference returns None when no point maps to the frame
            return None
        return i3, i2

    def compute_projection_batch(self, points, depths, camera_to_worlds):
        """All frames of a scan (scripts/project_multiview_features.py:96-118): (F, N + 1) index tensors, frames without
        correspondences are all-zero rows (count 0). No host sync."""
        rows = [self.compute_projection_launch(points, depths[i], camera_to_worlds[i]) for i in range(depths.shape[0])]
        return torch.stack([r[0] for r in rows]), torch.stack([r[1] for r in rows])

    @torch.no_grad()
    def project(self, label, lin_indices_3d, lin_indices_2d, num_points):
        label = label.float().contiguous()
        c = 1 if label.dim() == 2 else label.shape[0]
        out = torch.empty((c, num_points), dtype=torch.float32, device=label.device)
        _lib.call("irx_project_features", _lib.ptr(label), c, label.numel() // c, _lib.ptr(lin_indices_3d.contiguous()),
                  _lib.ptr(lin_indices_2d.contiguous()), num_points, _lib.ptr(out), _lib.stream_ptr())
        return out
