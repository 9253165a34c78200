"""SparseTensor + coordinate manager — the torchsparse-shaped container the reference code builds
(`SparseTensor(feats, coords[, stride])`, attributes `.F .C .s`, `.cuda()`, `+`; reference
models/attribute_module.py:70, models/basic_blocks.py:175-182, lib/dataset.py:234,261).

MI355X-first difference from torchsparse: rows are kept in *Morton (Z-order) key order* once a tensor
enters a convolution. That makes (i) down-sampling a segmented scan instead of a hash + unique,
(ii) 64-row output tiles spatially compact so whole kernel offsets can be skipped per tile, and
(iii) gathers L2-friendly. Everything downstream in the reference (BatchNorm, max-pool, BEV
scatter-add) is invariant to row order, so only the *set* of (coord, feature) rows is contractual.
"""
import numpy as np
import torch

from . import functional as F_


class DownMap:
    """Kernel map of a kernel-2 / stride-2 convolution between two pyramid levels."""

    def __init__(self, parent, koff, child, ld, out_level):
        self.parent = parent      # (n_in,) int32
        self.koff = koff          # (n_in,) uint8
        self.child = child        # (8, ld) int32
        self.ld = ld
        self.out_level = out_level
        self._child_t = None
        self._pairs = None

    def pairs(self):
        """Compacted pair lists of the child table (for the weight-gradient); built once, no host sync."""
        if self._pairs is None:
            self._pairs = F_.pairs_build(self.child, self.ld, self.out_level.n, 8)
        return self._pairs

    def child_t(self):
        """(8, n_in) table for the data-gradient: tbl[k][i] = parent[i] if koff[i] == k else -1."""
        if self._child_t is None:
            self._child_t = F_.kmap_down_transpose(self.parent, self.koff)
        return self._child_t, max(self.parent.shape[0], 1)


import os as _os
KMAP_DESCENT = _os.environ.get("IRX_KMAP_DESCENT", "1") != "0"      # dev / test switch: 0 = every level by window search + hash table


class Level:
    """One tensor stride of a coordinate pyramid: coords in ascending Morton-key order + cached maps
    (torchsparse caches `coord_maps` / `kernel_maps` on the tensor; here they live on the level and are
    shared by every tensor of that stride, including both convs of a ResidualBlock and the backward)."""

    def __init__(self, coords, keys, stride, batch_size):
        self.coords = coords      # (n,4) int32, (x,y,z,b)
        self.keys = keys          # (n,) int64 (uint64 bit pattern), ascending
        self.stride = int(stride)
        self.batch_size = int(batch_size)
        self._table = None
        self._nbr27 = None
        self._pairs27 = None
        self._order27 = None
        self._down = None
        self._offsets = None
        self._bev = {}

    @property
    def n(self):
        return self.coords.shape[0]

    def table(self):
        if self._table is None:
            self._table = F_.hash_build(self.keys)
        return self._table

    def nbr27(self):
        if self._nbr27 is None:
            self._nbr27 = F_.kmap_build_s1(self.coords, self.stride, self.table())
        return self._nbr27, int(self._nbr27.shape[1])       # (leading dimension: the batched builder pads rows to 64 entries)

    def order27(self):
        """Launch order of the 64-row output tiles of this level's 3^3 convolutions (heaviest first; csrc/irx_sched.hip)."""
        if self._order27 is None:
            tbl, ld = self.nbr27()
            self._order27 = F_.tile_order(tbl, ld, self.n, 27)
        return self._order27

    def pairs27(self):
        """Compacted pair lists of the 27-neighbour table (weight-gradients of both convs of a ResidualBlock)."""
        if self._pairs27 is None:
            tbl, ld = self.nbr27()
            self._pairs27 = F_.pairs_build(tbl, ld, self.n, 27)
        return self._pairs27

    def down(self):
        if self._down is None:
            parent, koff, oc, ok, child, ld, m = F_.downsample(self.keys, self.coords, self.stride)
            out = Level(oc, ok, self.stride * 2, self.batch_size)
            self._down = DownMap(parent, koff, child, ld, out)
        return self._down

    def tensors(self):
        """Every device tensor reachable from this level (for stream hand-over, see SparseTensor.record_stream)."""
        out = [self.coords, self.keys]
        if self._table is not None:
            out += [self._table[0], self._table[1]]
        if self._nbr27 is not None:
            out.append(self._nbr27)
        if self._pairs27 is not None:
            out += list(self._pairs27[:3])
        if self._order27 is not None:
            out.append(self._order27)
        if self._offsets is not None:
            out.append(self._offsets)
        for v in self._bev.values():
            out += list(v)
        if self._down is not None:
            d = self._down
            out += [d.parent, d.koff, d.child]
            if d._child_t is not None:
                out.append(d._child_t)
            if d._pairs is not None:
                out += list(d._pairs[:3])
            out += d.out_level.tensors()
        return out

    def build_pyramid(self, levels):
        """Down-sample `levels` times now, with ONE host sync for the whole pyramid (F_.pyramid_build: every level reads
        its row count on the device) instead of one per level; done BEFORE any convolution is queued so that the sync
        is cheap, which leaves the rest of the forward pass free of host syncs."""
        return self.build_pyramid_finish(self.build_pyramid_launch(levels))

    def build_pyramid_launch(self, levels):
        """Enqueue the missing levels; -> opaque pending state for build_pyramid_finish(). Splitting the two lets an
        input-preparation stage put useful host work between the launch and the (then free) sync."""
        lv, have = self, 0
        while have < levels and lv._down is not None:      # levels already built (lazily or by an earlier call)
            lv = lv._down.out_level
            have += 1
        if have == levels:
            return (lv, None)
        return (lv, F_.pyramid_launch(lv.keys, lv.coords, lv.stride, levels - have))

    def build_pyramid_finish(self, pending):
        lv, pend = pending
        if pend is not None:
            for parent, koff, oc, ok, child, ld, m in pend.finish()[1]:
                out = Level(oc, ok, lv.stride * 2, lv.batch_size)
                lv._down = DownMap(parent, koff, child, ld, out)
                lv = out
        return lv

    def build_tables(self, order_min_rows=None, backward=False, pairs_level0=False):
        """Enqueue NOW, on the current stream, every table the encoder's convolutions will ask for: the 27-neighbour table
        of each level of the (already built) pyramid and the tile launch order of the levels whose 3^3 convolutions run on
        k_spconv2 with more than one round of tiles. Called by the input-preparation stage (its own stream, one step
        ahead): built lazily these ten k_kmap_s1 launches and the ordering sat at the head of the encoders' streams."""
        from . import encoder_fn
        if order_min_rows is None:
            order_min_rows = encoder_fn.TILE_ORDER_MIN_ROWS if encoder_fn.TILE_ORDER else None
        self.build_kmaps()                           # every level's table in one native call (no-op for the ones already built)
        lv, first = self, True
        while lv is not None:
            if lv.n > 0:
                lv.nbr27()
                if not first and order_min_rows is not None and lv.n >= order_min_rows:
                    lv.order27()                     # (level 0 hosts the stem convolution: no k_spconv2 layer)
            first = False
            lv = lv._down.out_level if lv._down is not None else None
        if backward:
            self.build_backward_tables(pairs_level0)

    def build_backward_tables(self, pairs_level0=False):
        """The tables only the encoders' backward passes read, for the whole (already built) pyramid in ONE native call
        (csrc/torch_nodes.cpp backward_tables): pair lists of every stride-1 level's 27-neighbour table (level 0 only when its
        convolution's weight gradient runs over pair lists: the wide stem) and of every down-sampling map's child table, plus the
        transposed child maps. Needs the 27-neighbour tables (build_kmaps). No-op for what is already there."""
        from .. import _lib, _nodes
        mod = _nodes.load() if self.coords.is_cuda else None
        if mod is None or not hasattr(mod, "backward_tables"):
            return
        lvs, lv = [], self
        while lv is not None:
            lvs.append(lv)
            lv = lv._down.out_level if lv._down is not None else None
        jobs, holders, maps = [], [], []
        for i, lv in enumerate(lvs):
            if lv.n == 0:
                continue
            if (i > 0 or pairs_level0) and lv._pairs27 is None and lv._nbr27 is not None:
                tbl, ld = lv.nbr27()
                jobs.append((tbl, ld, lv.n, 27))
                holders.append((lv, "_pairs27"))
            dm = lv._down
            if dm is not None:
                if dm._pairs is None:
                    jobs.append((dm.child, dm.ld, dm.out_level.n, 8))
                    holders.append((dm, "_pairs"))
                if dm._child_t is None:
                    maps.append(dm)
        if not jobs and not maps:
            return
        out = mod.backward_tables([j[0] for j in jobs], [j[1] for j in jobs], [j[2] for j in jobs], [j[3] for j in jobs],
                                  [d.parent for d in maps], [d.koff for d in maps], _lib.stream_ptr())
        for t, (holder, attr) in enumerate(holders):
            setattr(holder, attr, (out[3 * t], out[3 * t + 1], out[3 * t + 2], max(jobs[t][2], 1)))
        for m, dm in enumerate(maps):
            dm._child_t = out[3 * len(jobs) + m]

    def build_kmaps(self):
        """Hash table + 27-neighbour table of EVERY level of the (already built) pyramid that does not have them yet, in one call
        of the C++ nodes module (csrc/torch_nodes.cpp kmaps_build: allocation and both launches per level without the interpreter);
        falls back to the lazy per-level builders when the module is not there. Same kernels, same tables."""
        from .. import _nodes
        mod = _nodes.load() if self.coords.is_cuda else None
        if mod is None or not hasattr(mod, "kmaps_build"):
            return
        lvs, lv = [], self
        while lv is not None:
            lvs.append(lv)
            lv = lv._down.out_level if lv._down is not None else None
        have = [int(lv._nbr27 is not None or lv.n == 0) for lv in lvs]
        if all(have):
            return
        from .. import _lib
        if KMAP_DESCENT and not any(have) and len(lvs) > 1 and hasattr(mod, "kmaps_build_pyramid") and all(lv.n > 0 for lv in lvs):
            # octree descent (irx_kmaps_build_pyramid): only the coarsest level is searched, every finer level is derived from the
            # next coarser one through the down-sampling maps — no hash tables below the top
            dms = [lv._down for lv in lvs[:-1]]
            out = mod.kmaps_build_pyramid([lv.keys for lv in lvs], [lv.coords for lv in lvs], [lv.stride for lv in lvs],
                                          [d.parent for d in dms], [d.koff for d in dms], [d.child for d in dms], [d.ld for d in dms],
                                          _lib.stream_ptr())
            for i, lv in enumerate(lvs):
                tk, tv, nbr = out[3 * i:3 * i + 3]
                if tk is not None and lv._table is None:
                    lv._table = (tk, tv, int(tk.shape[0]))
                lv._nbr27 = nbr
            return
        out = mod.kmaps_build([lv.keys for lv in lvs], [lv.coords for lv in lvs], [lv.stride for lv in lvs], have, _lib.stream_ptr())
        for i, lv in enumerate(lvs):
            if not have[i]:
                tk, tv, nbr = out[3 * i:3 * i + 3]
                if lv._table is None:
                    lv._table = (tk, tv, int(tk.shape[0]))
                lv._nbr27 = nbr

    def offsets(self):
        if self._offsets is None:
            self._offsets = F_.batch_offsets(self.coords, self.batch_size)
        return self._offsets

    def bev(self, nx, ny, nz):
        key = (nx, ny, nz)
        if key not in self._bev:
            self._bev[key] = F_.bev_table(self.coords, self.stride, self.batch_size, nx, ny, nz, self.table())
        return self._bev[key]

    def bev_t(self, nx, ny, nz):
        """The transposed BEV table (8, max(n, 1)) the data gradient of ToDenseBEVConvolution gathers through; cached, so that the
        preparation stage can build it with the forward table (InstanceRefer.prepare_finish) instead of the head's backward chain."""
        key = ("t", nx, ny, nz)
        if key not in self._bev:
            _, cell, zbin = self.bev(nx, ny, nz)
            self._bev[key] = (F_.kmap_down_transpose(cell, zbin),)
        return self._bev[key][0]


class SparseTensor:
    def __init__(self, feats, coords, stride=1, batch_size=None, _level=None):
        self.F = feats
        self.C = coords
        self.s = stride
        self.coord_maps = {}
        self.kernel_maps = {}
        self._batch_size = batch_size
        self._level = _level

    # --- torchsparse surface ---------------------------------------------------------------
    def check(self):
        if self.s not in self.coord_maps:
            self.coord_maps[self.s] = self.C

    def _as_torch(self):
        F, C = self.F, self.C
        if isinstance(F, np.ndarray):
            F = torch.from_numpy(np.ascontiguousarray(F)).float()
        if isinstance(C, np.ndarray):
            C = torch.from_numpy(np.ascontiguousarray(C)).int()
        return F, C

    def to(self, device, non_blocking=True):
        F, C = self._as_torch()
        return SparseTensor(F.to(device, non_blocking=non_blocking), C.to(device, non_blocking=non_blocking),
                            self.s, self._batch_size, None)

    def cuda(self):
        if isinstance(self.C, torch.Tensor) and self.C.is_cuda:
            return self
        return self.to(torch.device("cuda", torch.cuda.current_device()))

    def cpu(self):
        F, C = self._as_torch()
        return SparseTensor(F.cpu(), C.cpu(), self.s, self._batch_size, None)

    def detach(self):
        return SparseTensor(self.F.detach(), self.C, self.s, self._batch_size, self._level)

    def __add__(self, other):
        if self._level is not other._level or self._level is None:
            raise RuntimeError("SparseTensor + SparseTensor needs both operands on the same coordinate level")
        return SparseTensor(self.F + other.F, self.C, self.s, self._batch_size, self._level)

    def __repr__(self):
        n = self.F.shape[0] if hasattr(self.F, "shape") else "?"
        return "SparseTensor(n=%s, stride=%s)" % (n, self.s)

    # --- irx ------------------------------------------------------------------------------
    @property
    def batch_size(self):
        if self._batch_size is None:
            # the reference infers it the same way (a host sync): models/basic_blocks.py:235
            self._batch_size = int(self.C[:, 3].max().item()) + 1 if self.C.shape[0] else 0
        return self._batch_size

    def level(self):
        if self._level is None:
            raise RuntimeError("SparseTensor is not canonical yet; call .canonical()")
        return self._level

    def with_feats(self, feats, level=None):
        lv = level if level is not None else self._level
        return SparseTensor(feats, lv.coords, lv.stride, lv.batch_size, lv)

    def record_stream(self, stream):
        """Tell the caching allocator that `stream` will use the tensors of this SparseTensor and of its coordinate
        pyramid (they were produced on another stream, e.g. the input-preparation side stream)."""
        ts = [t for t in (self.F, self.C) if isinstance(t, torch.Tensor)]
        if self._level is not None:
            ts += self._level.tensors()
        seen = set()
        for t in ts:
            if t.is_cuda:
                # (the levels of a pyramid are views of five shared allocations: one record per storage is what the allocator keeps)
                key = t.untyped_storage().data_ptr()
                if key not in seen:
                    seen.add(key)
                    t.record_stream(stream)

    def canonical(self):
        """Rows re-ordered to ascending Morton key (device tensors). No-op when already canonical.
        Precondition (not checked here — a check would be a host sync on the hot path): voxel coordinates in [-32768, 32768)
        and batch indices in [0, 32768), the range a key holds; the voxeliser entry points (sparse/utils.py), which see raw
        point coordinates, do check it and raise."""
        if self._level is not None:
            return self
        F, C = self._as_torch()
        if not C.is_cuda:
            raise RuntimeError("irx sparse ops need the SparseTensor on a HIP device (call .cuda())")
        C = C.int().contiguous()
        F = F.float()
        keys = F_.coords_to_keys(C)
        skeys, perm = F_.sort_keys(keys, F_.morton_bits(self._batch_size))
        C = F_.keys_to_coords(skeys)                     # a key holds its row: decoded (streaming), not gathered through perm
        F = F.index_select(0, perm)
        bs = self._batch_size if self._batch_size is not None else (int(C[-1, 3].item()) + 1 if C.shape[0] else 0)
        lv = Level(C, skeys, self.s, bs)
        return SparseTensor(F, C, self.s, bs, lv)
