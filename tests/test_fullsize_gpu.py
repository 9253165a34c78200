"""Size-independent properties of the sparse path at BASELINE.json's FULL sizes (16 scenes x 50 k points: ~490 k voxels,
2 M neighbour pairs per level), where the CPU oracle would take minutes: sortedness / uniqueness of the coordinate
pyramid, the symmetry of the neighbour tables, idempotence of voxelisation, linearity of the convolution, and the
adjoint identities  <conv(x), g> == <x, dgrad(g)> == <w, wgrad(x, g)>  that tie the three conv kernels together.
Integer properties are exact; floating-point identities hold to the tolerance stated at each assert."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.sparse.utils import voxelize
    dev = torch.device("cuda")
    dd = S.make_batch(16, seed=321)
    pts = [torch.from_numpy(p) for p in dd["scene_points"]]
    allp = torch.cat(pts).to(dev)
    batch = torch.cat([torch.full((p.shape[0],), i, dtype=torch.int32) for i, p in enumerate(pts)]).to(dev)
    st = voxelize(allp[:, :3].contiguous(), allp.float(), batch, [0.05] * 3, 16)
    lv = st.level()
    lv.build_pyramid(4)
    return st, lv


def test_pyramid_sorted_unique_and_consistent(lib, scene):
    st, lv = scene
    assert lv.n > 300_000
    stride = 1
    for _ in range(5):
        keys = lv.keys
        assert bool((keys[1:] > keys[:-1]).all()), "Morton keys must be strictly ascending (sorted + unique)"
        c = lv.coords
        assert bool((c[:, :3] % stride == 0).all()), "coordinates stay in original-resolution units"
        if stride < 16:
            dm = lv.down()
            out = dm.out_level
            # every voxel's parent is floor(c / 2s) * 2s in the same scene; every parent has at least one child
            exp = torch.div(c[:, :3], 2 * stride, rounding_mode="floor") * (2 * stride)
            got = out.coords.index_select(0, dm.parent.long())
            assert torch.equal(got[:, :3], exp) and torch.equal(got[:, 3], c[:, 3])
            assert int(torch.unique(dm.parent).numel()) == out.n
            # child table <-> parent map
            child = dm.child[:, :out.n]
            valid = child >= 0
            assert int(valid.sum()) == lv.n
            rows = torch.arange(out.n, device=child.device, dtype=torch.int32).expand_as(child)
            assert torch.equal(dm.parent.index_select(0, child[valid].long()), rows[valid])
            lv = out
        stride *= 2


def test_neighbour_table_symmetry_and_pair_lists(lib, scene):
    _, lv = scene
    for _ in range(2):
        tbl, ld = lv.nbr27()
        n = lv.n
        t = tbl[:, :n]
        assert torch.equal(t[13], torch.arange(n, device=t.device, dtype=torch.int32)), "centre offset is the identity"
        for k in (0, 5, 12):
            j = t[k]
            v = j >= 0
            back = t[26 - k].index_select(0, j[v].long())
            assert torch.equal(back, torch.arange(n, device=t.device, dtype=torch.int32)[v]), "nbr[26-k][nbr[k][i]] == i"
        il, ol, counts, ldp = lv.pairs27()
        assert torch.equal(counts.long(), (t >= 0).sum(1)), "pair-list counts == valid table entries per offset"
        k = 3
        c = int(counts[k])
        assert torch.equal(il[k, :c], t[k][t[k] >= 0]) and bool((ol[k, 1:c] > ol[k, :c - 1]).all())
        lv = lv.down().out_level


def test_voxelisation_is_idempotent(lib, scene):
    """Voxelising one point per voxel (the voxel's own corner) reproduces exactly the same voxel set."""
    from instancerefer_amd.sparse.utils import voxelize
    st, lv = scene
    c = lv.coords
    xyz = c[:, :3].double() * 0.05 + 1e-6
    st2 = voxelize(xyz.contiguous(), xyz.float(), c[:, 3].contiguous(), [0.05] * 3, 16)
    assert st2.level().n == lv.n and torch.equal(st2.level().keys, lv.keys)


@pytest.mark.parametrize("level,cin,cout", [(1, 64, 64), (2, 128, 128)])
def test_conv_linearity_and_adjoints_full_size(lib, scene, level, cin, cout):
    from instancerefer_amd.sparse import functional as F_
    _, lv = scene
    for _ in range(level):
        lv = lv.down().out_level
    n = lv.n
    tbl, ld = lv.nbr27()
    g = torch.Generator(device="cuda").manual_seed(level)
    x1 = torch.randn(n, cin, device="cuda", generator=g)
    x2 = torch.randn(n, cin, device="cuda", generator=g)
    w = torch.randn(27, cin, cout, device="cuda", generator=g) * 0.05
    gy = torch.randn(n, cout, device="cuda", generator=g)
    conv = lambda x: F_.spconv_gather_gemm(x, w, tbl, ld, n, 27, cin, cout, 0, 0)
    y1, y2 = conv(x1), conv(x2)
    # linearity (fp32: each side is a different summation, so compare at 1e-4 of the output scale)
    lhs = conv(0.7 * x1 - 1.3 * x2)
    rhs = 0.7 * y1 - 1.3 * y2
    scale = rhs.abs().max().item()
    assert (lhs - rhs).abs().max().item() <= 1e-4 * scale
    # determinism: the same launch twice is bit-identical (no float atomics anywhere on the path)
    assert torch.equal(conv(x1), y1)
    # adjoint identities in float64 accumulation of the inner products
    dx = F_.spconv_gather_gemm(gy, w, tbl, ld, n, 27, cout, cin, 1, 1)          # data-gradient: flipped offsets, W^T
    dw = F_.spconv_wgrad_pairs(x1, gy, lv.pairs27(), n, 27, cin, cout)
    a = (y1.double() * gy.double()).sum().item()
    b = (x1.double() * dx.double()).sum().item()
    c = (w.double() * dw.double()).sum().item()
    ref = max(abs(a), 1.0)
    # the three numbers are sums of ~1e9 fp32 products each computed in a different order
    assert abs(a - b) <= 2e-5 * ref + 1e-2 and abs(a - c) <= 2e-5 * ref + 1e-2, (a, b, c)
