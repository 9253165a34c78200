"""Generate tests/golden/projection.npz by running the REFERENCE's own lib/projection.py (ProjectionHelper.
compute_projection + project, imported from /root/reference; build container only) on seeded synthetic frames
(instancerefer_amd.synthetic.make_frames). The fixture holds expected OUTPUTS only: per frame the number of
correspondences, the point / pixel index lists and the projected feature map; points, poses, depth maps and image
features are regenerated from the seed. CUDA hard-coding is neutralised (`.cuda()` -> identity): the arithmetic is the
reference's float32 torch code on CPU.   Usage:  python tests/golden/make_golden_projection.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [ROOT, REF]

from instancerefer_amd import synthetic as S  # noqa: E402

CFG = dict(scene_seed=4100, num_points=12000, frame_seed=4200, num_frames=5, channels=8)


def inputs(cfg=CFG):
    pts = S.make_scene(cfg["scene_seed"], num_points=cfg["num_points"], num_instances=5, num_candidates=2,
                       points_per_instance=16)["scene_points"][:, :3].astype(np.float32)
    poses, depths, feats = S.make_frames(cfg["frame_seed"], pts, cfg["num_frames"], cfg["channels"])
    return pts, poses, depths, feats


def main():
    torch.Tensor.cuda = lambda self, *a, **k: self
    from lib.projection import ProjectionHelper          # the reference's projection
    helper = ProjectionHelper(S.PROJ_INTRINSICS, S.PROJ_ARGS["depth_min"], S.PROJ_ARGS["depth_max"],
                              S.PROJ_ARGS["image_dims"], S.PROJ_ARGS["accuracy"], cuda=False)
    pts, poses, depths, feats = inputs()
    out = {"cfg": np.array([CFG[k] for k in ("scene_seed", "num_points", "frame_seed", "num_frames", "channels")])}
    n = pts.shape[0]
    for i in range(poses.shape[0]):
        res = helper.compute_projection(torch.from_numpy(pts), torch.from_numpy(depths[i]), torch.from_numpy(poses[i]))
        if res is None:
            out["count/%d" % i] = np.array(0)
            continue
        i3, i2 = res
        m = int(i3[0])
        out["count/%d" % i] = np.array(m)
        out["ind3d/%d" % i] = i3[1:1 + m].numpy().astype(np.int32)
        out["ind2d/%d" % i] = i2[1:1 + m].numpy().astype(np.int16)
        proj = helper.project(torch.from_numpy(feats[i]), i3, i2, n)        # (C, N)
        out["proj/%d" % i] = proj.numpy()[:, i3[1:1 + m].numpy()]            # the non-zero columns
        print("frame %d: %d of %d points" % (i, m, n))
    np.savez_compressed(os.path.join(HERE, "projection.npz"), **out)


if __name__ == "__main__":
    main()
