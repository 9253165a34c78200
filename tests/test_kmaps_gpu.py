"""The windowed, multi-level kernel-map builder (irx_kmaps_build_multi / k_kmap_win, csrc/irx_coords.hip; round 6) against an
independent statement of the same query: neighbour keys encoded by irx_coords_to_keys, located in the level's sorted key array by
torch.searchsorted. Semantics: torchsparse's `sphash(coords, offsets)` + `sphashquery` behind spnn.Conv3d (reference
models/basic_blocks.py:14-19; oracle/torchsparse/nn/functional.py:48-68): nbr[k][q] = the row holding voxel q + offset_k * stride, or
-1. Bit-exact (index work)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _expected(lv):
    """(27, n) int32 by searchsorted over the level's ascending keys"""
    from instancerefer_amd.sparse import functional as F_
    n, s = lv.n, lv.stride
    keys = lv.keys
    assert bool((keys[1:] > keys[:-1]).all()), "level keys are not strictly ascending"
    out = torch.empty((27, n), dtype=torch.int32, device=keys.device)
    c = lv.coords
    for k in range(27):
        dx, dy, dz = (k % 3) - 1, ((k // 3) % 3) - 1, (k // 9) - 1          # x fastest (odd kernel)
        nb = c.clone()
        nb[:, 0] += dx * s
        nb[:, 1] += dy * s
        nb[:, 2] += dz * s
        ok = ((nb[:, :3] >= -32768) & (nb[:, :3] < 32768)).all(1)
        nk = F_.coords_to_keys(torch.where(ok.unsqueeze(1), nb, c).contiguous())
        pos = torch.searchsorted(keys, nk).clamp(max=n - 1)
        found = (keys[pos] == nk) & ok
        out[k] = torch.where(found, pos, torch.full_like(pos, -1)).int()
    return out


def _levels(st):
    lv, out = st.level(), []
    while lv is not None:
        out.append(lv)
        lv = lv._down.out_level if lv._down is not None else None
    return out


@pytest.mark.parametrize("builder", ["descent", "window"])
@pytest.mark.parametrize("variant", ["corner", "centred"])
def test_multi_level_tables_equal_sorted_search(lib, variant, builder, monkeypatch):
    """every level of a 6-scene pyramid through ONE native call (Level.build_kmaps): by octree descent from the coarsest level
    (irx_kmaps_build_pyramid, the default) and by window search + hash table at every level (irx_kmaps_build_multi); 'centred' scenes
    straddle the coordinate origin, where biased Morton keys jump by 2^47 inside a window (the 32-bit window offsets saturate there)"""
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.sparse import tensor as T
    monkeypatch.setattr(T, "KMAP_DESCENT", builder == "descent")
    dev = torch.device("cuda")
    dd = S.to_device(S.make_batch(6, seed=11, num_points=30000, num_instances=6, num_candidates=3, points_per_instance=256,
                                  variant=variant), dev)
    st = dd["lidar"].canonical()
    st.level().build_pyramid(4)
    st.level().build_kmaps()
    lvs = _levels(st)
    assert len(lvs) == 5 and all(lv._nbr27 is not None for lv in lvs if lv.n)
    sizes = []
    for lv in lvs:
        nbr, ld = lv.nbr27()
        assert ld >= lv.n and nbr.shape == (27, ld)
        exp = _expected(lv)
        assert torch.equal(nbr[:, :lv.n], exp), "level stride %d" % lv.stride
        # the hash table the same call built answers every key with its row
        tk, tv, cap = lv.table()
        assert cap == tk.shape[0]
        sizes.append(lv.n)
    assert sizes[0] > 2048 > sizes[-1]          # both regimes: windowed levels with hash fallback, whole-level windows


def test_one_level_entry_on_unsorted_rows_falls_back_to_the_hash(lib):
    """irx_kmap_build_s1 derives the window keys from the coordinate rows; rows that are NOT in Morton order (a direct caller
    that did not sort) must still get the right table: the workgroup detects the disorder and resolves its probes in the hash table"""
    from instancerefer_amd.sparse import functional as F_
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    c = torch.randint(-40, 40, (20000, 3), generator=g)
    c = torch.unique(c, dim=0)
    c = c[torch.randperm(c.shape[0], generator=g)]
    coords = torch.cat([c, torch.zeros(c.shape[0], 1, dtype=torch.long)], 1).int().to(dev).contiguous()
    keys = F_.coords_to_keys(coords)
    nbr = F_.kmap_build_s1(coords, 1, F_.hash_build(keys))
    sk, order = torch.sort(keys)
    n = coords.shape[0]
    for k in (0, 5, 13, 14, 26):
        dx, dy, dz = (k % 3) - 1, ((k // 3) % 3) - 1, (k // 9) - 1
        nb = coords.clone()
        nb[:, 0] += dx
        nb[:, 1] += dy
        nb[:, 2] += dz
        nk = F_.coords_to_keys(nb.contiguous())
        pos = torch.searchsorted(sk, nk).clamp(max=n - 1)
        exp = torch.where(sk[pos] == nk, order[pos], torch.full_like(pos, -1)).int()
        assert torch.equal(nbr[k, :n], exp), "offset %d" % k
