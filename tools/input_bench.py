"""Dev tool: throughput of the device-side input pipeline (instancerefer_amd/scene_input.py) next to the numpy
restatement of the reference's __getitem__ (oracle/dataset_ref.py) on the same host, per sample.
  python tools/input_bench.py [--vertices 120000] [--points 40000] [--instances 32] [--batch 16] [--augment]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from instancerefer_amd import _lib, synthetic as S, scene_input as SI

ap = argparse.ArgumentParser()
ap.add_argument("--vertices", type=int, default=120000)
ap.add_argument("--points", type=int, default=40000)
ap.add_argument("--instances", type=int, default=32)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--augment", action="store_true")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
_lib.load()
dev = torch.device("cuda")
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dataset.npz"))
tables = SI.ClassTables(g["nyu40ids"], g["nyu40id2class"], g["mean_size_arr"])
raws = [S.make_raw_scene(900 + i, num_vertices=a.vertices, num_instances=a.instances, same_class=8) for i in range(a.batch)]
scans = [SI.ResidentScan(r, dev) for r in raws]
np.random.seed(0); torch.manual_seed(0)
t_draw, t_dev, t_total = [], [], []
for rep in range(a.reps + 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    draws = [SI.draw_sample(sc, 0, tables, num_points=a.points, augment=a.augment) for sc in scans]
    t1 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dd = SI.build_batch(draws, dev).finish()
    e1.record(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    if rep >= 2:
        t_draw.append(t1 - t0); t_dev.append(e0.elapsed_time(e1) * 1e-3); t_total.append(t2 - t0)
B = a.batch
# fully device-side mode (device generator)
t_dev2 = []
for rep in range(a.reps + 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dd = SI.build_batch_device(scans, [0] * a.batch, tables, dev, num_points=a.points, augment=a.augment).finish()
    torch.cuda.synchronize()
    if rep >= 2: t_dev2.append(time.perf_counter() - t0)
print("fully device-side mode (device RNG): wall %.3f ms/sample -> %.0f samples/s" % (1e3 * np.mean(t_dev2) / a.batch, a.batch / np.mean(t_dev2)))
print("device pipeline: host draws %.2f ms/sample, device work %.3f ms/sample (GPU events), wall %.2f ms/sample -> %.0f samples/s"
      % (1e3 * np.mean(t_draw) / B, 1e3 * np.mean(t_dev) / B, 1e3 * np.mean(t_total) / B, B / np.mean(t_total)))
from oracle import dataset_ref as DR
n = min(B, 4)
t0 = time.perf_counter()
for r in raws[:n]:
    DR.get_item(r, 0, 2, g["nyu40ids"], g["nyu40id2class"], g["mean_size_arr"], num_points=a.points, augment=a.augment)
t = (time.perf_counter() - t0) / n
print("numpy restatement of the reference __getitem__ (1 thread): %.1f ms/sample -> %.1f samples/s" % (1e3 * t, 1 / t))
