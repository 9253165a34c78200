"""Multiview back-projection on the device (instancerefer_amd/projection.py -> csrc/irx_project.hip) against
tests/golden/projection.npz — the output of the REFERENCE's own lib/projection.py:191-279 — and, at a full-size scan
(200 k points, 12 frames), against the numpy oracle that is pinned to that fixture. Index lists are integers: bit-exact."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _inputs():
    spec = importlib.util.spec_from_file_location("make_golden_projection", os.path.join(HERE, "golden", "make_golden_projection.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.inputs()


def test_projection_matches_reference_fixture(lib):
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.projection import ProjectionHelper
    g = np.load(os.path.join(HERE, "golden", "projection.npz"))
    pts, poses, depths, feats = _inputs()
    dev = torch.device("cuda")
    helper = ProjectionHelper(S.PROJ_INTRINSICS, **S.PROJ_ARGS)
    p = torch.from_numpy(pts).to(dev)
    n = pts.shape[0]
    for i in range(poses.shape[0]):
        res = helper.compute_projection(p, torch.from_numpy(depths[i]).to(dev), torch.from_numpy(poses[i]).to(dev))
        m = int(g["count/%d" % i])
        assert (res is None) == (m == 0)
        if res is None:
            continue
        i3, i2 = res
        assert i3.dtype == torch.int64 and i3.shape == (n + 1,) and int(i3[0]) == m == int(i2[0])
        assert np.array_equal(i3[1:1 + m].cpu().numpy(), g["ind3d/%d" % i])
        assert np.array_equal(i2[1:1 + m].cpu().numpy(), g["ind2d/%d" % i])
        assert not bool(i3[1 + m:].any()) and not bool(i2[1 + m:].any())
        out = helper.project(torch.from_numpy(feats[i]).to(dev), i3, i2, n)
        assert out.shape == (feats.shape[1], n)
        cols = torch.from_numpy(g["ind3d/%d" % i].astype(np.int64)).to(dev)
        assert np.array_equal(out.index_select(1, cols).cpu().numpy(), g["proj/%d" % i])
        mask = torch.ones(n, dtype=torch.bool, device=dev)
        mask[cols] = False
        assert not bool(out[:, mask].any())


def test_projection_full_size_scan_vs_oracle(lib):
    from oracle import projection_ref as PR
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.projection import ProjectionHelper
    dev = torch.device("cuda")
    pts = S.make_scene(77, num_points=200000, num_instances=32, num_candidates=4, points_per_instance=16)["scene_points"][:, :3].astype(np.float32)
    poses, depths, feats = S.make_frames(78, pts, num_frames=12, channels=128)
    helper = ProjectionHelper(S.PROJ_INTRINSICS, **S.PROJ_ARGS)
    w, h = S.PROJ_ARGS["image_dims"]
    p = torch.from_numpy(pts).to(dev)
    i3s, i2s = helper.compute_projection_batch(p, torch.from_numpy(depths).to(dev), torch.from_numpy(poses).to(dev))
    assert i3s.shape == (12, 200001)
    i3h, i2h = i3s.cpu().numpy(), i2s.cpu().numpy()
    total = 0
    for i in range(12):
        e3, e2 = PR.compute_projection(pts, depths[i], helper._params(torch.from_numpy(poses[i])), w, h)
        m = int(i3h[i, 0])
        assert m == len(e3) == int(i2h[i, 0]), (i, m, len(e3))
        assert np.array_equal(i3h[i, 1:1 + m], e3) and np.array_equal(i2h[i, 1:1 + m], e2)
        total += m
        if i == 0:
            out = helper.project(torch.from_numpy(feats[i]).to(dev), i3s[i], i2s[i], len(pts)).cpu().numpy()
            assert np.array_equal(out, PR.project(feats[i], e3, e2, len(pts)))
    assert total > 10000
