"""Dev tool: host issue time vs GPU time of the fully device-side input stage (build_batch_device + finish)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from instancerefer_amd import _lib, synthetic as S, scene_input as SI
_lib.load(); dev = torch.device("cuda")
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dataset.npz"))
tables = SI.ClassTables(g["nyu40ids"], g["nyu40id2class"], g["mean_size_arr"])
B = 16
scans = [SI.ResidentScan(S.make_raw_scene(3000 + i, num_vertices=120000, num_instances=8, same_class=4), dev) for i in range(B)]
tb, tf, tg = [], [], []
for rep in range(12):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    p = SI.build_batch_device(scans, [0] * B, tables, dev, num_points=50000)
    t1 = time.perf_counter()
    torch.cuda.synchronize()          # so that finish() measures host work only
    t2 = time.perf_counter()
    dd = p.finish()
    t3 = time.perf_counter(); e1.record(); torch.cuda.synchronize()
    if rep >= 2: tb.append(t1 - t0); tf.append(t3 - t2); tg.append(e0.elapsed_time(e1) * 1e-3 - (t2 - t1) * 0)
print("build_batch_device host issue %.2f ms, finish() host %.2f ms per batch of %d; wall incl. GPU %.2f ms" % (1e3 * np.mean(tb), 1e3 * np.mean(tf), B, 1e3 * np.mean(tg)))
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    p = SI.build_batch_device(scans, [0] * B, tables, dev, num_points=50000); torch.cuda.synchronize(); dd = p.finish()
pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print("\n".join(l[:150] for l in s.getvalue().splitlines()[:30]))
