"""Dev tool: the kernel maps of one B = 16 scene pyramid and one candidate pyramid, timed alone (HIP events, the batched builder
irx_kmaps_build_multi through the C++ nodes module).   python tools/kmap_bench.py [batch=16]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instancerefer_amd import _lib, _nodes, synthetic as S
from instancerefer_amd.instancerefer import InstanceRefer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda")
model = InstanceRefer(7, S.default_args()).to(dev).train()
dd = S.to_device(S.make_batch(B, seed=5, num_points=50000, num_instances=8, num_candidates=4), dev)
dd = model.prepare(dd)
mod = _nodes.load()
for name, st in (("scene", dd["lidar"]), ("candidates", dd["_attr_prepared"][0])):
    lvs, lv = [], st.level()
    while lv is not None:
        lvs.append(lv)
        lv = lv._down.out_level if lv._down is not None else None
    dms = [l._down for l in lvs[:-1]]
    builders = {
        "window + hash (irx_kmaps_build_multi)": lambda: mod.kmaps_build([l.keys for l in lvs], [l.coords for l in lvs], [l.stride for l in lvs],
                                                                         [0] * len(lvs), _lib.stream_ptr()),
        "octree descent (irx_kmaps_build_pyramid)": lambda: mod.kmaps_build_pyramid([l.keys for l in lvs], [l.coords for l in lvs], [l.stride for l in lvs],
                                                                                    [d.parent for d in dms], [d.koff for d in dms], [d.child for d in dms],
                                                                                    [d.ld for d in dms], _lib.stream_ptr())}
    for bname, fn in builders.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        n = sum(l.n for l in lvs)
        m = sum(int((out[3 * i + 2][:, :l.n] >= 0).sum()) for i, l in enumerate(lvs))
        us = e0.elapsed_time(e1) / reps * 1e3
        byts = n * 27 * 8.0 + 8.0 * m + 16.0 * n
        print("%-10s %-42s levels %s: %d voxels, %d pairs: %.1f us per pyramid, %.0f GB/s algorithmic (%.3f of 8 TB/s)" %
              (name, bname, [l.n for l in lvs], n, m, us, byts / us / 1e3, byts / us / 1e3 / 8000))
