"""Dev tool: END-TO-END training throughput — every step builds its batch from scans resident in HBM with the fully
device-side input pipeline (scene_input.build_batch_device: sub-sampling, augmentation, instance split, boxes, resample,
both voxelisations), then runs the full model forward + loss + backward + Adam. Nothing is cached between steps.
  python tools/e2e_train_bench.py [--dtype bf16] [--batch 16] [--steps 60]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import instancerefer_amd as irx
from instancerefer_amd import _lib, synthetic as S, scene_input as SI
from instancerefer_amd.instancerefer import InstanceRefer
from instancerefer_amd.loss_helper import DatasetConfig, get_loss
from instancerefer_amd.optim import FlatAdam

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--scans", type=int, default=32)
ap.add_argument("--vertices", type=int, default=120000)
ap.add_argument("--points", type=int, default=50000)
ap.add_argument("--instances", type=int, default=8)
ap.add_argument("--augment", action="store_true")
ap.add_argument("--json", action="store_true", help="print one JSON object instead of the text line (bench.py reads it)")
a = ap.parse_args()
_lib.load()
irx.set_compute_dtype("bf16" if a.dtype == "bf16" else "fp32")
dev = torch.device("cuda")
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dataset.npz"))
tables = SI.ClassTables(g["nyu40ids"], g["nyu40id2class"], g["mean_size_arr"])
scans = [SI.ResidentScan(S.make_raw_scene(3000 + i, num_vertices=a.vertices, num_instances=a.instances, same_class=4), dev)
         for i in range(a.scans)]
torch.manual_seed(0)
model = InstanceRefer(7, S.default_args()).to(dev).train()
opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
cfg = DatasetConfig(mean_size_arr=g["mean_size_arr"])
B = a.batch
rng = np.random.default_rng(0)
lang = np.zeros((B, 126, 300), np.float32); lang[:, :30] = rng.standard_normal((B, 30, 300)) * 0.4
lang_dev = torch.from_numpy(lang).to(dev)
lang_len = torch.full((B,), 30, dtype=torch.int64, device=dev)
obj_cat = torch.full((B,), 2, dtype=torch.int64, device=dev)


def make_pending(step):
    pick = [(step * B + i) % len(scans) for i in range(B)]
    return SI.build_batch_device([scans[j] for j in pick], [0] * B, tables, dev, num_points=a.points, augment=a.augment)


def finish(p):
    dd = p.finish()
    dd["lang_feat"], dd["lang_len"], dd["lang_len_max"], dd["object_cat"] = lang_dev, lang_len, 30, obj_cat
    dd["_host"]["object_cat"] = np.full(B, 2, np.int64)
    dd["unique_multiple"] = torch.ones(B, dtype=torch.int64)
    return dd


from instancerefer_amd.loss_helper import prepare_labels
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


TIMES = [0.0, 0.0, 0]      # stage_launch on the worker, main thread waiting for it, steps
WORKER = None
if os.environ.get("IRX_E2E_WORKER", "1") != "0":
    import bench as _bench
    WORKER = _bench._Worker(torch.cuda.current_device())
POST_EARLY = os.environ.get("IRX_E2E_POST") == "early"     # dev A/B: the worker's job posted at the head of the step (whole step as its window)
PROF = None
if os.environ.get("IRX_E2E_CPROFILE") == "1" and WORKER is None:       # dev: host profile of stage_launch (IRX_E2E_WORKER=0)
    import cProfile
    PROF = cProfile.Profile()
RESIDENT = None      # second leg: the sampled batch of the last end-to-end step, reused as a resident input


def stage_launch(step):
    """On the side stream: collect batch step+1's input (enqueued one step ago: its read-back has long arrived), enqueue
    its model-side preparation (candidate voxelisation, pyramids, labels), then enqueue the input pipeline of batch
    step+2. Nothing here waits for the GPU. Resident leg: the same model-side preparation on a fixed sampled batch."""
    global pend
    with torch.cuda.stream(side):
        if RESIDENT is not None:
            dd = dict(RESIDENT)
            dd["_host"] = dict(RESIDENT["_host"])
        else:
            dd = finish(pend)
        dd = model.prepare_launch(dd)
        dd["_loss_prepared"] = prepare_labels(dd, cfg, dev) if "_attr_prepared" in dd else None
        if RESIDENT is None:
            pend = make_pending(step + 2)
    return dd


def stage_finish(dd):
    with torch.cuda.stream(side):
        return model.prepare_finish(dd)


def loop(cur, n_warm, n_steps):
    t0, out = None, None
    for step in range(n_warm + n_steps):
        if step == n_warm:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        main.wait_stream(side)
        model.hand_over(cur, main)
        for t in list(cur.values()) + [cur["irx"].xyz64, cur["irx"].pts32, cur["irx"].centres]:   # made on the side stream
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(main)
        if WORKER is None:
            if PROF is not None and step >= n_warm and RESIDENT is None:
                PROF.enable()
                launched = stage_launch(step)
                PROF.disable()
            else:
                launched = stage_launch(step)
        box = {}
        if WORKER is not None and POST_EARLY:
            def job():
                t_ = time.perf_counter()
                box["dd"] = stage_launch(step)
                TIMES[0] += time.perf_counter() - t_
            WORKER.post(job)
        opt.zero_grad()
        out = get_loss(model(cur), cfg)
        if WORKER is not None and not POST_EARLY:
            # the next batches' host work (box labels, augmentation draws, ~40 small launches) on a helper thread while this
            # thread sits in the autograd engine's C++ loop (GIL released): in the host-paced bf16 mode that Python was 1.5 ms
            # of a 6.4 ms step on the training thread (round 5: end to end / resident 0.73)
            def job():
                t_ = time.perf_counter()
                box["dd"] = stage_launch(step)
                TIMES[0] += time.perf_counter() - t_
            WORKER.post(job)
        out["loss"].backward()
        if WORKER is not None:
            t_ = time.perf_counter()
            WORKER.wait()
            TIMES[1] += time.perf_counter() - t_
            TIMES[2] += 1
            launched = box["dd"]
        if os.environ.get("IRX_E2E_FINISH_EARLY") == "1":      # dev A/B: collect the level sizes (and enqueue the tables) before the optimizer
            cur = stage_finish(launched)
            opt.backward_step()
        else:
            opt.backward_step()
            cur = stage_finish(launched)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out, cur


with torch.cuda.stream(side):
    pend = make_pending(0)
    cur = model.prepare(finish(pend))
    pend = make_pending(1)
dt, out, cur = loop(cur, a.warmup, a.steps)
# the SAME process, model and scene statistics with the input pipeline taken out of the loop (a fixed sampled batch; candidate
# voxelisation, Morton sort, pyramids and labels still redone every step, as in bench.py's resident-input loop): the ratio
# of the two is what the input pipeline costs, free of box-to-box and process-to-process differences
with torch.cuda.stream(side):
    RESIDENT = finish(pend)
torch.cuda.synchronize()
dt_res, _, _ = loop(cur, max(5, a.warmup // 3), a.steps)
if PROF is not None:
    import io, pstats
    buf = io.StringIO()
    pstats.Stats(PROF, stream=buf).sort_stats("tottime").print_stats(45)
    sys.stderr.write("stage_launch over %d steps:\n%s\n" % (a.steps, buf.getvalue()))
    buf = io.StringIO()
    pstats.Stats(PROF, stream=buf).sort_stats("cumulative").print_stats(40)
    sys.stderr.write(buf.getvalue())
if os.environ.get("IRX_E2E_TIMES") and TIMES[2]:
    sys.stderr.write("worker: stage_launch %.3f ms/step, training thread waited %.3f ms/step\n" % (1e3 * TIMES[0] / TIMES[2], 1e3 * TIMES[1] / TIMES[2]))
if a.json:
    import json
    print(json.dumps({"value": B * a.steps / dt, "unit": "scenes/s", "ms_per_step": 1e3 * dt / a.steps, "steps": a.steps,
                      "warmup": a.warmup, "dtype": a.dtype, "scenes_per_step": B, "points_per_scene": a.points,
                      "resident_scans": a.scans, "vertices_per_scan": a.vertices, "augment": bool(a.augment),
                      "what": "every step builds its batch from scans resident in HBM with the device-side input pipeline "
                              "(sampling, instance split, boxes, both voxelisations), then forward + loss + backward + Adam; "
                              "nothing cached between steps", "loss": float(out["loss"]),
                      "resident_same_process": {"value": B * a.steps / dt_res, "ms_per_step": 1e3 * dt_res / a.steps,
                                                "what": "same process / model / scenes with one sampled batch reused as a resident "
                                                        "input (model-side preparation still redone every step)"},
                      "ratio_to_resident": dt_res / dt}))
else:
    print("end to end (%s, B=%d, %d pts from %d-vertex scans, input pipeline in the loop): %.1f scenes/s, %.2f ms/step, loss %.4f"
          % (a.dtype, B, a.points, a.vertices, B * a.steps / dt, 1e3 * dt / a.steps, float(out["loss"])))
    print("  same process, resident sampled batch: %.1f scenes/s, %.2f ms/step -> end to end / resident = %.3f"
          % (B * a.steps / dt_res, 1e3 * dt_res / a.steps, dt_res / dt))
