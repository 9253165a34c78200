#!/usr/bin/env python
"""bench.py — scenes/sec, forward + backward (+ gradient all-reduce + Adam step) of the InstanceRefer hot
path on synthetic ScanRefer-shaped scenes, N GPUs of one node, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 --steps 20 --warmup 5        (no torchrun: starts the 8 ranks itself, self_launch())
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task brief) with `roofline` (dominant HIP kernel, measured
with events on the launch stream during extra instrumented steps) and `cpu_baseline` (the CPU oracle — a
restatement of the reference path — timed on a bounded sample of the same workload on this host's cores).
Inputs (scene voxels source points, instance points, utterances) are resident in HBM before the timed
region; per step everything the reference does inside forward/backward is redone from scratch: Morton
sort + hash + kernel maps of the scene tensor, candidate voxelisation, 26 sparse convs, BN, heads, loss,
backward, all-reduce, Adam.

Default dtype: bf16 (operands + activation / gradient storage inside the two sparse encoders, fp32 accumulation, statistics,
parameters and heads) — the dtype BASELINE configs[2]-[4] name for this metric; the reference's own fp32 (the dtype of the 1e-4
parity gate) runs in the same process and is reported, with its own roofline, under `alt_dtype` (`--dtype f32` makes it the
headline and bf16 the alternative).

Also in the line (N = 1): `alt_dtype` (the same loop in the other dtype with its own roofline), `dense_path`
(language encoder alone against the MFMA peak) and `end_to_end` (tools/e2e_train_bench.py in its own process: the per-sample
input pipeline inside the loop, nothing resident but the raw scans).
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4). One rank uses four streams (main, scene encoder,
# language, input preparation); with more than one rank RCCL adds streams of its own, and a stream that shares a queue with another
# runs behind it — round 5 measured a HALVED step when a fifth stream appeared. 8 queues measured neutral at N = 1 (2 984 vs 2 999
# scenes/s). Must be set before the runtime initialises.
# (ranks that SHARE one GPU — the IRX_BENCH_SHARE_GPU test rig — must not: two processes x 8 hardware queues on one device
# oversubscribe its queue slots and every launch then waits for a queue switch: 31.8 s per step measured, 3.5 s with 4)
import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import instancerefer_amd as _irx_pkg
# (the package no longer touches the environment at import: the queue count is this program's decision — ADVICE r5)
if not torch.cuda.is_initialized():          # (imported by a test after the runtime started: the queue count is what it is)
    _irx_pkg.configure_hw_queues(ranks_per_device=2 if os.environ.get("IRX_BENCH_SHARE_GPU") == "1" else 1)

PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_F32_TFLOPS = 157.3      # fp32 vector == fp32-input MFMA peak
PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA peak (MI355X_MICROARCH.md; --dtype bf16 only)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="full", choices=["full", "attr", "stress"],
                    help="full = whole InstanceRefer (BASELINE configs[2]/[3] shape); attr = configs[1]; stress = configs[4] "
                         "(200k points, 64 instances, 16 candidates, multiview C0 = 135, bf16)")
    ap.add_argument("--batch", type=int, default=0, help="scenes per GPU (default 16 full / 8 attr / 8 stress)")
    ap.add_argument("--points", type=int, default=None)
    ap.add_argument("--instances", type=int, default=None)
    ap.add_argument("--candidates", type=int, default=None)
    ap.add_argument("--multiview", type=int, default=None, help="extra ENet feature channels per point (128 in the stress config)")
    ap.add_argument("--tokens", type=int, default=30)
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16", "bf16op"],
                    help="compute dtype of the sparse encoders: bf16 (default for the full / stress workloads = BASELINE "
                         "configs[2]-[4]: bf16 operands / fp32 accumulation AND bf16 storage of the activations and gradients "
                         "inside the encoders; BatchNorm statistics, parameters, heads fp32); f32 (the reference's own dtype, "
                         "the 1e-4 parity gate; default for --workload attr = configs[1]); bf16op (bf16 operands only, every "
                         "tensor fp32 in HBM)")
    ap.add_argument("--no-alt-dtype", action="store_true",
                    help="skip the extra leg in the OTHER dtype (N = 1: fp32 beside a bf16 headline, bf16 beside fp32) "
                         "reported under 'alt_dtype'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scenes", type=int, default=16)
    ap.add_argument("--cpu-timeout", type=int, default=150)
    ap.add_argument("--profile-steps", type=int, default=2, help="instrumented (serially issued) steps behind the timed region; 0 = none")
    ap.add_argument("--sync-bn", action="store_true", help="N > 1: BatchNorm statistics over all ranks (instancerefer_amd.syncbn; "
                    "the encoders stay in the one-call executor, which calls back for one small all-reduce per BatchNorm layer "
                    "and direction)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end leg (input pipeline in the loop, own process)")
    ap.add_argument("--prep-thread", action="store_true",
                    help="run the input preparation on a helper thread instead of inline behind the step (same speed: "
                         "the loop is GIL-bound, tools/micro/ab_thread.py)")
    ap.add_argument("--no-prep-at-backward", action="store_true",
                    help="default: the launch phase of the next batch's input preparation is issued by a persistent helper "
                         "thread WHILE the autograd engine runs this step's backward (the Python thread is parked in "
                         "loss.backward() and the engine's C++ nodes do not hold the GIL): +5 %% in the host-bound bf16 "
                         "mode (2127 -> 2239 scenes/s). This flag issues it inline before the forward instead")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do the input preparation of each step inline instead of on a side stream during the previous backward")
    args = ap.parse_args()
    stress = args.workload == "stress"
    for k, full, st in (("points", 50000, 200000), ("instances", 8, 64), ("candidates", 4, 16), ("multiview", 0, 128),
                        ("dtype", "bf16", "bf16")):
        if getattr(args, k) is None:
            setattr(args, k, st if stress else full)
    if args.workload == "attr" and "--dtype" not in sys.argv:
        args.dtype = "f32"                       # BASELINE configs[1] names fp32
    if stress and args.cpu_scenes == 16:
        args.cpu_scenes = 2                      # bounded CPU sample: these scenes are 4x the size
    return args


def build_model(args_ns, workload, device):
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.instancerefer import InstanceRefer
    margs = S.default_args()
    if workload == "attr":
        margs.relation_module = None
        margs.scene_module = None
    model = InstanceRefer(7 + int(getattr(args_ns, "multiview", 0) or 0), margs)
    model.load_state_dict(S.seeded_state_dict(model, 2024))
    return model.to(device).train()


_SPIN_US = float(os.environ.get("IRX_BENCH_SPIN_US", "0"))
_SPIKES = [] if os.environ.get("IRX_BENCH_SPIKES") == "1" else None   # dev: host clock at the phase boundaries of every step
_SPIKE_MARKS = []
_PREP_WORKER_ALL = os.environ.get("IRX_BENCH_PREP_WORKER", "launch") == "all"
_MARKS = None          # dev (IRX_BENCH_TIMELINE=1): list receiving (name, event on the main stream, host clock) per step


_STEP_T = [] if os.environ.get("IRX_BENCH_STEPTIMES") == "1" else None      # dev: host clock at every step start (start-up transient)
_HOSTCLK = [0.0, 0.0, 0.0, 0] if os.environ.get("IRX_BENCH_HOSTCLOCK") == "1" else None   # issue | backward() | in step_fn, steps


def _mark(name):
    if _HOSTCLK is not None:
        # dev (IRX_BENCH_HOSTCLOCK=1; multi-rank rehearsal on one box): the training thread's own clock from "step start" to
        # "optimizer issued" — what a rank's host needs to ISSUE a step, whatever the (shared) GPU is doing behind it
        now = time.perf_counter()
        if name == "step start":
            _HOSTCLK.append(now)
        elif name == "loss issued" and len(_HOSTCLK) > 4:
            _HOSTCLK.append(now)
            _HOSTCLK[0] += now - _HOSTCLK[4]          # forward + loss issued
        elif name == "backward returned" and len(_HOSTCLK) > 5:
            _HOSTCLK[1] += now - _HOSTCLK[5]          # inside backward()
            _HOSTCLK.append(now)
        elif name == "optimizer issued" and len(_HOSTCLK) > 6:
            _HOSTCLK[2] += now - _HOSTCLK[6]          # gather + all-reduce (gloo here: a blocking host copy) + Adam launch
            _HOSTCLK[3] += 1
            del _HOSTCLK[4:]
    if _MARKS is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        _MARKS.append((name, ev, time.perf_counter()))


def fresh_batch(resident):
    """The next batch as a training loop would hand it over: a FRESH un-canonical scene SparseTensor (Morton sort, hash,
    kernel maps, pair lists are rebuilt every step) and no cached candidate selection."""
    from instancerefer_amd.sparse import SparseTensor
    dd = dict(resident)
    dd["irx"]._sel_cache.clear()
    if "lidar_F" in resident:
        dd["lidar"] = SparseTensor(resident["lidar_F"], resident["lidar_C"], 1, batch_size=resident["B"])
    return dd


def prepare_next(model, resident, state, phase="all"):
    """Input preparation of the NEXT step (candidate voxelisation, Morton sort, coordinate pyramids, relation node
    features, IoU labels: everything that depends only on the inputs, including the part of the forward that needs host
    syncs) on its own HIP stream. Inline mode (default) splits it around the issue of step N: the kernels are LAUNCHED
    before step N's forward is issued (phase "launch": model.prepare_launch) and the level sizes are collected after
    step N's optimizer launch (phase "finish"), by which time they have long arrived in pinned host memory — the syncs
    cost nothing and the kernels overlap step N on the GPU. `--prep-thread` runs both phases on a helper thread instead
    (measured equal: the loop is bound by Python/dispatch work under the GIL). Multi-rank runs use the same two phases
    on the main stream (see main()). Every step still does exactly one preparation of a
    fresh batch; nothing is cached."""
    side = state.setdefault("side", torch.cuda.Stream())
    dev = torch.cuda.current_device()

    def launch():
        with torch.cuda.stream(side):
            nxt = model.prepare_launch(fresh_batch(resident))
            if state.get("labels") is not None:          # host half of get_loss (IoU labelling) is input-only too
                nxt["_loss_prepared"] = state["labels"](nxt)
            state["launched"] = nxt

    def finish():
        with torch.cuda.stream(side):
            state["next"] = model.prepare_finish(state.pop("launched"))

    def work():
        try:
            torch.cuda.set_device(dev)
            launch()
            finish()
        except BaseException as e:                      # surfaced by the training thread at join time
            state["next_error"] = e

    if state.get("threaded", True):
        import threading
        th = threading.Thread(target=work, name="irx-input-prep", daemon=True)
        state["thread"] = th
        th.start()
    elif phase == "launch":
        launch()
    elif phase == "finish":
        finish()
        state["thread"] = None
    else:
        launch()
        finish()
        state["thread"] = None


def take_prepared(model, state):
    if "thread" not in state:
        return None
    th = state.pop("thread")
    if th is not None:
        if "join_wait" in state:
            t0 = time.perf_counter()
            th.join()
            state["join_wait"].append(time.perf_counter() - t0)
        else:
            th.join()
    if "next_error" in state:
        raise state.pop("next_error")
    dd = state.pop("next")
    main = torch.cuda.current_stream()
    main.wait_stream(state["side"])                     # the prepared tensors are complete
    model.hand_over(dd, main)
    return dd


def step_fn(model, resident, workload, reducer, opt, state=None):
    """One training step on the resident batch. Returns the loss tensor (no host sync)."""
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss, compute_lang_classification_loss
    _sp = [time.perf_counter()] if _SPIKES is not None else None
    if _SPIKES is not None:
        from instancerefer_amd import instancerefer as _IR
        _fm = []
        _IR.MARK = lambda name, _fm=_fm: _fm.append((name, time.perf_counter()))
        _SPIKE_MARKS.append(_fm)
    dd = take_prepared(model, state) if state is not None else None
    if dd is None:
        dd = fresh_batch(resident)
    if _sp is not None: _sp.append(time.perf_counter())
    at_bwd = state is not None and state.get("pipeline") and state.get("at_backward")
    if state is not None and state.get("pipeline") and not at_bwd:
        # batch N+1: threaded -> the whole preparation runs beside this step; inline -> only its launch phase now
        prepare_next(model, resident, state, phase="launch")
    _mark("step start")
    if _STEP_T is not None:
        _STEP_T.append(time.perf_counter())
    opt.zero_grad()
    if _SPIN_US:                                         # dev: is the loop host- or GPU-paced? (host-only busy-wait)
        t_end = time.perf_counter() + _SPIN_US * 1e-6
        while time.perf_counter() < t_end:
            pass
    dd = model(dd)
    if _sp is not None: _sp.append(time.perf_counter())
    _mark("forward issued")
    if workload in ("full", "stress"):
        loss = get_loss(dd, step_fn.cfg)["loss"]
    else:
        # configs[1]: attribute path only — contrastive loss on the attribute scores + language CE
        # (the reference's per-sample ContrastiveLoss, lib/loss_helper.py:93-107,248-258, for all scenes in one fused launch
        # each way, IoU labels as in get_loss; the relation / scene scores of the full model are absent here)
        from instancerefer_amd.dense import ContrastiveFn
        from instancerefer_amd.loss_helper import prepare_labels
        loss = compute_lang_classification_loss(dd)
        lp = dd.pop("_loss_prepared", None) or prepare_labels(dd, step_fn.cfg, loss.device)
        if lp["total"] and lp["srow"]:
            zero = torch.zeros_like(dd["attribute_scores"])
            ref = ContrastiveFn.apply(dd["attribute_scores"], zero, zero, lp["lab"], lp["seg_off"], lp["keep_dev"], 5.0, 0.2)
            loss = loss + 10.0 * ref / lp["batch_size"]
    if at_bwd:
        w = state.get("worker")
        if w is None:
            w = state["worker"] = _Worker(torch.cuda.current_device())
        # IRX_BENCH_PREP_WORKER=all (dev A/B): the worker also FINISHES the preparation (waits for the level sizes, builds the kernel
        # maps / tables) inside the backward window instead of the training thread doing that behind the optimizer
        w.post(lambda: prepare_next(model, resident, state, phase="all" if _PREP_WORKER_ALL else "launch"))
    _mark("loss issued")
    if _sp is not None: _sp.append(time.perf_counter())
    # (an explicit unit gradient: loss.backward() alone allocates and fills a ones_like(loss) per step — ~0.1 ms of host time and a
    #  launch at the head of the backward; same arithmetic)
    one = step_fn.__dict__.get("_one")
    if one is None or one.device != loss.device or one.shape != loss.shape or one.dtype != loss.dtype:
        one = step_fn.__dict__["_one"] = torch.ones_like(loss)
    loss.backward(gradient=one)
    _mark("backward returned")
    if _sp is not None: _sp.append(time.perf_counter())
    if at_bwd:
        state["worker"].wait()
    if _sp is not None: _sp.append(time.perf_counter())
    opt.backward_step()          # one cat -> one all-reduce (N > 1) -> one fused Adam launch
    if _sp is not None: _sp.append(time.perf_counter())
    _mark("optimizer issued")
    if state is not None and state.get("pipeline") and not state.get("threaded", True) and not (at_bwd and _PREP_WORKER_ALL):
        prepare_next(model, resident, state, phase="finish")   # level sizes arrived during the step: no wait
    if _sp is not None:
        _sp.append(time.perf_counter())
        _SPIKES.append(_sp)
    return loss


def prime(model, resident, args, reducer, opt, state):
    """model.warm(): IRX_BENCH_PRIME untimed steps of the real training step BEFORE the contract's warm-up steps are
    counted, with parameters, Adam moments / step counts, BatchNorm running statistics and the RNG state restored afterwards. ON by
    default (IRX_BENCH_PRIME = 30 untimed steps, disclosed in the JSON line as `primed_steps`; 0 switches it off) — measured, round 5, alternating runs on one box, 5 warm-up + 20 timed steps: 0 primed steps 3 036 / 2 940, 40 primed
    2 996 / 3 020, 150 primed 2 822 / 2 929 scenes/s, 30 + 100 without priming 2 746-3 067 on the same box within minutes: the spread
    of the pool's boxes (+-5 %, other tenants on the host, and a downward drift under SUSTAINED load: consecutive 30-step blocks of
    one process read 5.4 -> 7.6 ms/step over ten seconds) is larger than anything the first five steps leave behind, and more
    untimed load before the timed region makes the reading worse, not better."""
    n = int(os.environ.get("IRX_BENCH_PRIME", "30"))
    if n <= 0:
        return 0
    # Round 6: the FIRST process that allocates on a fresh box runs its first ~0.6 s with 14-17 ms hiccups every few steps (host clock
    # per step on such a box: 17 14 15 22 4 16 15 6 14 5 15 63 5 3 3 3 ... then 3.3-3.5 throughout; a second process on the same box
    # does not) — first touch of device memory, not anything the step does. 30 steps are 0.12 s: the priming therefore also lasts at
    # least IRX_BENCH_PRIME_S seconds (default 1.0, at most 400 steps). Measured on fresh boxes: 30 steps only 3 386 scenes/s, 600 steps
    # 4 140, a second process 4 145.
    min_s = float(os.environ.get("IRX_BENCH_PRIME_S", "1.0"))
    bufs = [b for b in model.buffers()]
    snap = (opt.flat_p.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), list(opt.steps), [b.clone() for b in bufs],
            torch.cuda.get_rng_state(), torch.get_rng_state())
    t0, done = time.perf_counter(), 0
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        n, min_s = max(n, 200 if min_s > 0 else n), 0.0     # every rank must run the SAME number of steps (collectives inside): a fixed count
    while done < n or (time.perf_counter() - t0 < min_s and done < 400):
        step_fn(model, resident, args.workload, reducer, opt, state)
        done += 1
        if done % 25 == 0:
            torch.cuda.synchronize()          # (the host runs ahead of the GPU: the clock above must see the GPU's pace)
    n = done
    torch.cuda.synchronize()
    with torch.no_grad():
        opt.flat_p.copy_(snap[0])
        opt.exp_avg.copy_(snap[1])
        opt.exp_avg_sq.copy_(snap[2])
        opt.steps = snap[3]
        for b, v in zip(bufs, snap[4]):
            b.copy_(v)
    torch.cuda.set_rng_state(snap[5])
    torch.set_rng_state(snap[6])
    torch.cuda.synchronize()
    return n


def timeline(model, resident, args, reducer, opt, state, F_, n=12):
    """Dev (IRX_BENCH_TIMELINE=1): GPU-side clock of the REAL pipelined loop, no syncs added — main-stream events at the step's
    phase boundaries plus the library's per-layer events of both encoders (IRX_ENC_PROF, recorded on the launch streams by
    the asynchronous lanes) — printed to stderr relative to each step's start, averaged over the last n - 2 steps."""
    global _MARKS
    from instancerefer_amd.sparse import encoder_fn
    encoder_fn.EVENT_POOL.extend(encoder_fn._new_event() for _ in range(n * 2 * 6 * 13))
    torch.cuda.synchronize()
    _MARKS, F_.PROFILE, encoder_fn.PROFILE_NO_COUNT = [], [], True
    F_.PROFILE_NO_COUNT = True
    from instancerefer_amd import instancerefer as _ir
    _ir.MARK = _mark
    t0 = time.perf_counter()
    for _ in range(n):
        step_fn(model, resident, args.workload, reducer, opt, state)
    torch.cuda.synchronize()
    log("timeline steps: %.2f ms/step" % ((time.perf_counter() - t0) * 1e3 / n))
    marks, recs = _MARKS, F_.PROFILE
    _ir.MARK = None
    _MARKS, F_.PROFILE, encoder_fn.PROFILE_NO_COUNT = None, None, False
    F_.PROFILE_NO_COUNT = False
    per = len(marks) // n
    rper = len(recs) // n
    rows = {}
    for s in range(4, n):
        m = marks[s * per:(s + 1) * per]
        ev0, h0 = m[0][1], m[0][2]
        nxt = marks[(s + 1) * per] if s + 1 < n else None
        for name, ev, h in m:
            rows.setdefault(("main", name), []).append((ev0.elapsed_time(ev), (h - h0) * 1e3))
        if nxt is not None:
            rows.setdefault(("main", "next step start"), []).append((ev0.elapsed_time(nxt[1]), (nxt[2] - h0) * 1e3))
        r = recs[s * rper:(s + 1) * rper]
        # encoders in issue order: a record stream of (kind, n_out, K, cin, cout, M, start, stop, esz); a new encoder pass
        # begins at every "fwd" record whose input is a stem (cin <= 8 or > 128)
        encs, cur = [], None
        for rec in r:
            if rec[0] == "fwd" and (rec[3] <= 8 or rec[3] > 128):
                cur = []
                encs.append(cur)
            cur.append(rec)
        for ei, e in enumerate(encs):
            tag = "encoder %d (%d voxels)" % (ei, e[0][1])
            f = [x for x in e if x[0] == "fwd"]
            b = [x for x in e if x[0] != "fwd"]
            rows.setdefault((tag, "forward first kernel starts"), []).append((ev0.elapsed_time(f[0][6]), None))
            rows.setdefault((tag, "forward last conv ends"), []).append((ev0.elapsed_time(f[-1][7]), None))
            bs = min(ev0.elapsed_time(x[6]) for x in b)
            be = max(ev0.elapsed_time(x[7]) for x in b)
            rows.setdefault((tag, "backward first conv starts"), []).append((bs, None))
            rows.setdefault((tag, "backward last conv ends"), []).append((be, None))
            big = [x for x in b if x[1] >= 50000 or (x[0] == "dgrad" and x[1] >= 50000)]
            if big:
                rows.setdefault((tag, "backward reaches the >= 50 k-row levels"), []).append((min(ev0.elapsed_time(x[6]) for x in big), None))
    out = []
    for (who, name), v in rows.items():
        g = sum(a for a, _ in v) / len(v)
        h = [b for _, b in v if b is not None]
        out.append((g, who, name, sum(h) / len(h) if h else None))
    out.sort()
    log("timeline (ms after the step's start; GPU clock | host clock when issued):")
    for g, who, name, h in out:
        sys.stderr.write("  %8.3f  %-24s %-42s %s\n" % (g, who, name, "" if h is None else "host %.3f" % h))


class _Worker:
    """A persistent helper thread (no thread start per step): post() a callable, wait() for it (re-raises)."""

    def __init__(self, device):
        import queue
        import threading
        self.q, self.done, self.err = queue.Queue(), threading.Event(), None

        def loop():
            torch.cuda.set_device(device)
            while True:
                fn = self.q.get()
                try:
                    fn()
                except BaseException as e:
                    self.err = e
                self.done.set()
        threading.Thread(target=loop, name="irx-prep-worker", daemon=True).start()

    def post(self, fn):
        self.done.clear()
        self.q.put(fn)

    def wait(self):
        self.done.wait()
        if self.err is not None:
            e, self.err = self.err, None
            raise e


def usable_cores(mask_size=None):
    """Cores this process may actually use: affinity mask (or a given mask size) capped by the cgroup CPU quota."""
    try:
        n = mask_size or len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    """Host CPU model string (/proc/cpuinfo) + the logical core count of the machine (the baseline itself runs on `cores`)."""
    name = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    name = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return "%s (%d logical cores on the host)" % (name or "unknown", os.cpu_count() or 0)


def cpu_baseline_worker(points, instances, candidates, tokens, n_scenes, threads, multiview=0):
    """Runs in a child process: the oracle (CPU restatement of the reference path; kind 'port') on a
    bounded sample of the workload. Prints one JSON line."""
    torch.set_num_threads(threads)
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    from oracle.model_ref import InstanceRefer as OracleModel, oracle_data_dict
    n = max(2, n_scenes)      # BatchNorm1d heads need >= 2 samples
    model = OracleModel(7 + multiview, S.default_args())
    model.load_state_dict(S.seeded_state_dict(model, 2024))
    model.train()
    host = S.make_batch(n, seed=123, num_points=points, num_instances=instances,
                        num_candidates=candidates, tokens=tokens, multiview=multiview)
    t0 = time.perf_counter()
    dd = oracle_data_dict(host)          # scene sparse_quantize: done by the dataloader in the reference
    t_prep = time.perf_counter() - t0
    best = None
    for _ in range(2):                   # second pass = warm allocator / thread pool
        model.zero_grad()
        t0 = time.perf_counter()
        out = get_loss(model(dict(dd)), DatasetConfig())
        out["loss"].backward()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    torch_rate = n / best
    # C/OpenMP port (oracle/csrc/spconv_cpu.c): the two sparse encoders (97 % of the model's FLOPs) forward + backward
    c_rate, c_note = None, ""
    try:
        from oracle import cpu_port
        from oracle.torchsparse.utils import sparse_quantize
        sd = model.state_dict()
        ps, _ = cpu_port.pack_encoder_params(sd, "scene.net.")
        pa, _ = cpu_port.pack_encoder_params(sd, "attribute.net.")
        lid = dd["lidar"]
        cs, fs = lid.C.numpy(), lid.F.numpy()
        ca, fa = [], []
        k = 0
        for i in range(n):
            for j in range(candidates):
                pc = host["instance_points"][i][j]
                c, f = sparse_quantize(pc[:, :3], pc, quantization_size=0.02)
                ca.append(np.concatenate([c, np.full((len(c), 1), k)], 1))
                fa.append(f)
                k += 1
        ca, fa = np.concatenate(ca).astype(np.int32), np.concatenate(fa).astype(np.float32)
        gs, ga = np.ones((n, 128), np.float32), np.ones((k, 128), np.float32)
        tbest = None
        for _ in range(2):
            t0 = time.perf_counter()
            cpu_port.encoder_fwd_bwd(cs, fs, n, ps, gs, threads, fast=True)
            cpu_port.encoder_fwd_bwd(ca, fa, k, pa, ga, threads, fast=True)
            dt = time.perf_counter() - t0
            tbest = dt if tbest is None else min(tbest, dt)
        c_rate = n / tbest
        c_note = "; C/OpenMP port (oracle/csrc/spconv_cpu.c, the -ffast-math build oracle/_build/libirx_oracle_cpu_fast.so, the two sparse encoders fwd+bwd only, kernel maps included): %.2f scenes/s" % c_rate
    except Exception as e:   # the C port is optional strengthening; never lose the PyTorch-CPU number over it
        c_note = "; C/OpenMP port unavailable (%r)" % (e,)
    # the same C port on ONE thread (2 scenes: a bounded sample; scenes/s scales with the scene count)
    one_thread = None
    try:
        if c_rate is not None and n >= 2:
            ms = cs[:, 3] < 2
            ma = ca[:, 3] < 2 * candidates
            t0 = time.perf_counter()
            cpu_port.encoder_fwd_bwd(np.ascontiguousarray(cs[ms]), np.ascontiguousarray(fs[ms]), 2, ps, gs[:2], 1, fast=True)
            cpu_port.encoder_fwd_bwd(np.ascontiguousarray(ca[ma]), np.ascontiguousarray(fa[ma]), 2 * candidates, pa,
                                     ga[:2 * candidates], 1, fast=True)
            one_thread = 2 / (time.perf_counter() - t0)
    except Exception:
        one_thread = None
    value = max(torch_rate, c_rate or 0.0)
    print(json.dumps({"value": value, "unit": "scenes/s", "cores": threads, "cpu_model": cpu_model(), "kind": "port",
                      "one_thread": {"value": one_thread, "unit": "scenes/s", "cores": 1,
                                     "what": "C/OpenMP port, the two sparse encoders fwd+bwd, 1 thread, 2 scenes"},
                      "sample": "%d scenes x %d pts; the faster of: oracle/model_ref.py (CPU PyTorch gather-GEMM-scatter "
                                "restatement of the reference path, full model fwd+bwd, %d threads): %.2f scenes/s%s; scene "
                                "voxelisation (%.2fs) excluded as in the reference's dataloader"
                                % (n, points, threads, torch_rate, c_note, t_prep),
                      "torch_port": torch_rate, "c_openmp_port": c_rate, "seconds": n / value}), flush=True)


def cpu_baseline(args, workload):
    """Bounded, sandboxed: child process with a hard timeout so the baseline can never stall the bench."""
    import subprocess
    orig = _ORIG_AFFINITY                     # this process is pinned to a few cores (bind_rank_to_cores): the CPU leg is not
    threads = min(usable_cores(len(orig) if orig else None), 64)
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "try:\n    os.sched_setaffinity(0, %r)\nexcept Exception:\n    pass\n"
            "import bench; bench.cpu_baseline_worker(%d, %d, %d, %d, %d, %d, %d)"
            % (ROOT, orig or [], args.points, args.instances, args.candidates, args.tokens, args.cpu_scenes, threads, args.multiview))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=args.cpu_timeout, env=env)
    except subprocess.TimeoutExpired:
        return {"error": "cpu baseline exceeded %ds" % args.cpu_timeout, "cores": threads}
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return {"error": (r.stderr or r.stdout)[-400:], "cores": threads}


def self_launch(args):
    """`python bench.py --gpus N` without torchrun (WORLD_SIZE unset): start the N ranks ourselves — the reference is a
    single process (scripts/train.py:216-223), so this launcher is the build's own entry to the N-GPU path. The ranks run
    under torch.distributed.run (standalone rendezvous on 127.0.0.1, a free port); their output is relayed, and rank 0's
    JSON line is re-printed LAST on stdout. Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, IRX_BENCH_SELF_LAUNCHED="1")
    env.setdefault("OMP_NUM_THREADS", "4")              # torchrun would set 1 and warn; main() caps per rank anyway
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes needs it on this driver)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line_json = None
    for line in proc.stdout:
        t = line.strip()
        if t.startswith("{") and '"metric"' in t:
            line_json = t                                # held back: printed last
        else:
            sys.stdout.write(line)
            sys.stdout.flush()
    rc = proc.wait()
    if line_json is None:
        print("bench.py --gpus %d: the %d-rank launch produced no result line (exit %d)" % (args.gpus, args.gpus, rc),
              file=sys.stderr)
        return rc or 1
    n = json.loads(line_json).get("n_gpus")
    if n != args.gpus:
        print("bench.py --gpus %d: result line reports n_gpus = %r" % (args.gpus, n), file=sys.stderr)
        return 1
    print(line_json, flush=True)
    return rc


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out


_ORIG_AFFINITY = None


def bind_rank_to_cores(local_rank, world, device_index, share=False):
    """One process per GPU: give each rank its own slice of host cores — the cores of its GPU's NUMA node (PCI bus id ->
    /sys/bus/pci/devices/<bdf>/numa_node -> node cpulist) divided among the ranks whose GPUs sit on that node, or an even
    slice of the allowed cores when the topology is not readable — and cap the intra-op thread pools. 8 ranks x (Python
    + 2 library lanes + a preparation worker) otherwise wander over all cores and across sockets. Returns a description."""
    global _ORIG_AFFINITY
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return {"cores": None}
    if _ORIG_AFFINITY is None:
        _ORIG_AFFINITY = list(allowed)               # the CPU-baseline child gets the whole mask back
    # A rank = the Python thread + two library lanes + a preparation worker: a COMPACT block of cores (one L3 domain) instead
    # of the whole machine. Measured on a 2 x 64-core EPYC host with the GPU's NUMA node not readable in the container: the
    # host-bound bf16 loop runs 2019 / 2362 / 2439 / 2056 scenes/s left to the scheduler and 2456-2476 pinned to 8 (or 4) cores.
    block = int(os.environ.get("IRX_BENCH_CORES_PER_RANK", "8"))
    if len(allowed) <= max(block, 2 * world):
        torch.set_num_threads(max(1, min(4, len(allowed))))
        return {"cores": len(allowed), "numa": None}
    node_of = {}
    if not share:
        for r in range(world):
            try:
                pr = torch.cuda.get_device_properties(r)
                bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
                with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
                    node_of[r] = int(f.read())
            except (OSError, AttributeError, ValueError, RuntimeError):
                node_of = {}
                break
    mine, numa = None, None
    if node_of and node_of.get(local_rank, -1) >= 0:
        numa = node_of[local_rank]
        try:
            with open("/sys/devices/system/node/node%d/cpulist" % numa) as f:
                cpus = [c for c in _cpulist(f.read()) if c in set(allowed)]
            peers = sorted(r for r, n in node_of.items() if n == numa)
            per = min(len(cpus) // len(peers), block)
            if per >= 2:
                i = peers.index(local_rank)
                skip = block if len(cpus) >= (len(peers) + 1) * block else 0      # leave the node's first cores to the OS
                mine = cpus[skip + i * per:skip + (i + 1) * per]
        except (OSError, ValueError):
            mine = None
    if mine is None:
        per = min(len(allowed) // world, block)
        skip = block if len(allowed) >= (world + 1) * block else 0
        mine = allowed[skip + local_rank * per:skip + (local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    torch.set_num_threads(max(1, min(4, len(mine))))
    return {"cores": len(mine), "numa": numa, "first_core": mine[0]}


def log(msg):
    if os.environ.get("IRX_BENCH_VERBOSE"):
        print("[bench %.1fs] %s" % (time.perf_counter() - log.t0, msg), file=sys.stderr, flush=True)


log.t0 = time.perf_counter()


def main():
    args = parse()
    if os.environ.get("IRX_BENCH_VERBOSE"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("IRX_BENCH_DUMP_AFTER", "90")), repeat=True, file=sys.stderr)
    if os.environ.get("IRX_BENCH_AUTOGRAD_ST") == "1":    # dev A/B: backward on the calling thread (no engine-thread hand-over)
        torch.autograd.set_multithreading_enabled(False)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))                      # no torchrun around us: start the N ranks ourselves
    if args.gpus != world:                               # never report n_gpus: 1 for --gpus 8 (or the reverse)
        raise SystemExit("--gpus %d != WORLD_SIZE %d" % (args.gpus, world))
    import torch.distributed as dist
    # IRX_BENCH_SHARE_GPU=1 (test only): all ranks use cuda:0 and gloo, to exercise the multi-rank logic on a 1-GPU box
    share = os.environ.get("IRX_BENCH_SHARE_GPU") == "1"
    device = torch.device("cuda", 0 if share else local_rank)
    torch.cuda.set_device(device)
    binding = bind_rank_to_cores(local_rank, world, device.index, share) if os.environ.get("IRX_BENCH_NO_BIND") != "1" else None
    # IRX_BENCH_FORCE_DIST=1 (test only, N = 1): a one-rank RCCL group, every collective of the N > 1 path issued for real
    force_dist = world == 1 and os.environ.get("IRX_BENCH_FORCE_DIST") == "1"
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            # NOT init_process_group(..., device_id=device): the eager communicator that argument creates makes every
            # training step 3 ms slower on this stack even when no collective is issued (measured with one rank:
            # 12.2 vs 9.6 ms/step; a gloo group or the lazily created NCCL communicator cost nothing). The device is
            # bound by torch.cuda.set_device above and named explicitly in every barrier.
            dist.init_process_group("nccl")
    elif force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1)

    from instancerefer_amd import _build, _lib
    _build.build_lib()
    _lib.load()
    import instancerefer_amd as irx
    irx.set_compute_dtype({"f32": "fp32", "bf16": "bf16", "bf16op": "bf16_operands"}[args.dtype])
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.loss_helper import DatasetConfig
    from instancerefer_amd.sparse import functional as F_

    B = args.batch or (16 if args.workload == "full" else 8)
    model_workload = "attr" if args.workload == "attr" else "full"     # stress = the full model on bigger inputs
    torch.manual_seed(1234)                              # identical replicas: the same initial weights on every rank
    model = build_model(args, model_workload, device)    # (FlatAdam broadcasts rank 0's parameters / buffers anyway)
    if args.sync_bn:
        from instancerefer_amd.syncbn import convert_sync_batchnorm
        convert_sync_batchnorm(model)
    torch.manual_seed(1234 + rank)                       # dropout masks: per rank, reproducible run to run
    step_fn.cfg = DatasetConfig()
    # weak scaling: every rank owns B distinct scenes (seeds offset by rank)
    host = S.make_batch(B, seed=123 + rank * B, num_points=args.points, num_instances=args.instances,
                        num_candidates=args.candidates, tokens=args.tokens, multiview=args.multiview)
    log("host batch made")
    resident = S.to_device(host, device)
    torch.cuda.synchronize()
    log("resident on device")
    lidar = resident.pop("lidar")
    perm = torch.randperm(lidar.F.shape[0], device=device)   # un-sorted rows, as a dataloader would deliver
    resident["lidar_F"] = lidar.F[perm].contiguous()
    resident["lidar_C"] = lidar.C[perm].contiguous()
    resident["B"] = B
    n_scene_vox = int(lidar.F.shape[0])
    from instancerefer_amd.optim import FlatAdam
    reducer = None
    opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=world, module=model,
                   overlap=os.environ.get("IRX_OPT_OVERLAP", "1") != "0")   # dev A/B: early all-reduce of the encoder ranges
    opt._force_collectives = force_dist

    if world > 1 or force_dist:
        # the RCCL communicator exists now (FlatAdam broadcast / first barrier): push whatever the library printed while it
        # was created (NCCL_DEBUG=VERSION banner, C stdio buffer) out NOW, so that rank 0's JSON line is the last line
        dist.barrier(device_ids=None if share else [device.index])
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier(device_ids=None if share else [device.index])
        torch.cuda.synchronize()

    # Input-prep pipeline: launch phase before the step is issued, finish phase (level sizes) after it — the host never
    # blocks in a sync — on its own stream, so the preparation kernels also overlap the step; the same at every N (one
    # process per GPU). History: with blocking syncs in the preparation, two gloo ranks time-sharing ONE GPU (the
    # IRX_BENCH_SHARE_GPU test rig) stalled ~250 ms per sync behind a third stream, so N > 1 used the main stream; since
    # the preparation is sync-free that rig runs the side stream faster than the main one (1054-1090 vs 1019-1038
    # scenes/s for 2 ranks on one GPU), and the main-stream variant costs 5 % on a GPU of its own.
    state = {"pipeline": not args.no_pipeline, "threaded": bool(args.prep_thread) and world == 1,
             "at_backward": not args.no_prep_at_backward and not args.prep_thread}
    if os.environ.get("IRX_BENCH_PREP_MAIN") == "1":   # dev A/B: the preparation on the main stream
        state["side"] = torch.cuda.current_stream()
    if os.environ.get("IRX_BENCH_SERIAL") == "1":
        # dev (profiling): EVERYTHING on one stream, issued by one thread — under rocprofv3 every kernel's duration is then its
        # time alone on the GPU and the sum over a step is the step's serial GPU time (the budget overlap can only re-arrange)
        from instancerefer_amd import instancerefer as _ir
        from instancerefer_amd.sparse import encoder_fn as _ef
        _ir._STREAMS = _ir._LANG_THREAD = False
        _ef.ASYNC = False
        model.args.overlap_streams = False
        state["side"] = torch.cuda.current_stream()
        state["at_backward"] = False
    from instancerefer_amd.loss_helper import prepare_labels
    state["labels"] = lambda dd: prepare_labels(dd, step_fn.cfg, device) if "_attr_prepared" in dd else None
    # (N > 1 with a GPU per rank: primed too — a fixed 200 steps on every rank, see prime() — so that the per-N values of a scaling run
    #  are taken in the same state as the N = 1 one: a fresh box starts in a low-power state and its first second reads 10-20 % low.
    #  The shared-GPU test rig is not primed: its steps are 10x longer and it measures logic, not speed)
    primed = prime(model, resident, args, reducer, opt, state) if (world == 1 or not share) else 0
    for i in range(args.warmup):
        step_fn(model, resident, args.workload, reducer, opt, state)
        if i == 0:
            torch.cuda.synchronize()
            log("first warmup step done")
    # gc.freeze(): the model, the plans and the resident tensors created so far move to the permanent generation, so the
    # cyclic collector (still enabled) stops re-scanning them on every collection — host-bound bf16: 2187 -> 2319 scenes/s
    # averaged over three 150-step runs each (IRX_BENCH_GC=none / off for the A/B); no effect on the GPU-bound fp32 line
    if os.environ.get("IRX_BENCH_SWITCH_US"):          # dev A/B: GIL hand-over interval (default 5000 us)
        sys.setswitchinterval(float(os.environ["IRX_BENCH_SWITCH_US"]) * 1e-6)
    import gc
    # (round 6: the 6-10 ms pauses on every ~30th step that IRX_BENCH_SPIKES=1 showed were generation-2 collections of a reference
    #  cycle per coordinate pyramid — level -> cached encoder plan -> level, ~200 objects and ~120 device tensors per step — fixed in
    #  sparse/encoder_fn.Plan; freeze / off / default now time the same, tools/micro/gc_cycles.py counts what a step leaves behind)
    gc_mode = os.environ.get("IRX_BENCH_GC", "freeze")
    if gc_mode == "freeze":
        gc.collect()
        gc.freeze()
    elif gc_mode == "off":
        gc.collect()
        gc.disable()
    barrier()
    log("warmup done")
    if _HOSTCLK is not None:
        _HOSTCLK[0] = _HOSTCLK[1] = 0.0
        _HOSTCLK[3] = 0
        del _HOSTCLK[4:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step_fn(model, resident, args.workload, reducer, opt, state)
    barrier()
    dt = time.perf_counter() - t0
    if _HOSTCLK is not None and _HOSTCLK[3]:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "hostclock_rank%d_of_%d.txt" % (rank, world)), "w") as f:
            f.write("rank %d of %d: training thread's clock per step: forward + loss issued %.3f ms | inside backward() %.3f ms | gather + "
                    "all-reduce + Adam launch %.3f ms (gloo on a shared GPU: a blocking host copy of the 32 MB gradient buffer); wall "
                    "%.3f ms/step over %d steps; B = %d; cores %s\n"
                    % (rank, world, 1e3 * _HOSTCLK[0] / _HOSTCLK[3], 1e3 * _HOSTCLK[1] / _HOSTCLK[3], 1e3 * _HOSTCLK[2] / _HOSTCLK[3],
                       1e3 * dt / args.steps, _HOSTCLK[3], B, binding))
    if world > 1 or force_dist:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())
    if _STEP_T is not None and rank == 0:
        d = [1e3 * (b - a) for a, b in zip(_STEP_T[:-1], _STEP_T[1:])]
        sys.stderr.write("step times (ms, host clock at step start; warm-up %d then timed): %s\n" % (args.warmup, " ".join("%.1f" % v for v in d)))
    log("timed region done: %.1f ms/step" % (1000.0 * dt / args.steps))
    if _SPIKES is not None and rank == 0:
        names = ("take_prepared", "forward", "get_loss", "backward", "worker wait", "optimizer", "prepare finish")
        rows = _SPIKES[-args.steps:]
        tot = [1e3 * (r[-1] - r[0]) for r in rows]
        med = sorted(tot)[len(tot) // 2]
        sys.stderr.write("host phases (ms) of the timed steps: median step %.2f; steps over 2x the median:\n" % med)
        for i, r in enumerate(rows):
            if tot[i] > 2 * med:
                sys.stderr.write("  step %3d %6.1f ms: %s\n" % (i, tot[i], "  ".join("%s %.1f" % (n, 1e3 * (b - a)) for n, a, b in zip(names, r[:-1], r[1:]))))
                fm = _SPIKE_MARKS[len(_SPIKE_MARKS) - len(rows) + i]
                sys.stderr.write("        forward marks: %s\n" % "  ".join("%s +%.1f" % (n.replace("fwd: ", ""), 1e3 * (t - r[1])) for n, t in fm))
        sums = [sum(1e3 * (r[j + 1] - r[j]) for r in rows) / len(rows) for j in range(len(names))]
        sys.stderr.write("  mean per phase: %s\n" % "  ".join("%s %.2f" % (n, v) for n, v in zip(names, sums)))

    if rank == 0 and os.environ.get("IRX_BENCH_CPROFILE") == "1":     # dev: host profile of the real pipelined loop
        import cProfile, io, pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(16):
            step_fn(model, resident, args.workload, reducer, opt, state)
        pr.disable()
        torch.cuda.synchronize()
        for key in ("cumulative", "tottime"):
            buf = io.StringIO()
            pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(70)
            sys.stderr.write("\n".join(l[:160] for l in buf.getvalue().splitlines()) + "\n(16 steps)\n")
    if rank == 0 and os.environ.get("IRX_BENCH_TORCHPROF") == "1":     # dev: per-operator / per-autograd-node host time of the loop
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU]) as prof:
            for _ in range(8):
                step_fn(model, resident, args.workload, reducer, opt, state)
        torch.cuda.synchronize()
        sys.stderr.write(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=70, max_name_column_width=70) + "\n(8 steps)\n")
    if rank == 0 and os.environ.get("IRX_BENCH_TIMELINE") == "1":
        timeline(model, resident, args, reducer, opt, state, F_)

    # ---- the same loop in the other dtype (fp32 = the reference's / bf16 = BASELINE configs[2]-[4]'s), reported beside the headline ----
    alt = None
    other = {"f32": "bf16", "bf16": "f32"}.get(args.dtype)
    if world == 1 and other is not None and not args.no_alt_dtype:
        irx.set_compute_dtype("bf16" if other == "bf16" else "fp32")
        try:
            for _ in range(max(20, min(args.warmup, 40))):        # (a dtype switch re-sizes every arena: its own settling time)
                step_fn(model, resident, args.workload, reducer, opt, state)
            if gc_mode == "freeze":
                gc.collect()                           # the plans / arenas of the new mode join the permanent generation too
                gc.freeze()
            barrier()
            ak = max(1, min(args.steps, 100))
            t0 = time.perf_counter()
            for _ in range(ak):
                step_fn(model, resident, args.workload, reducer, opt, state)
            barrier()
            adt = time.perf_counter() - t0
            alt = {"dtype": other, "value": B * ak / adt, "unit": "scenes/s", "ms_per_step": 1000.0 * adt / ak,
                   "steps": ak, "warmup": max(20, min(args.warmup, 40)),
                   "what": ("same loop, irx_set_compute_dtype(2) = BASELINE configs[2]-[4] dtype: bf16 operands / fp32 "
                            "accumulation in the 32/64/128-channel sparse convs (fwd, dgrad, wgrad) and bf16 STORAGE of every "
                            "activation / gradient tensor inside the two encoders; BatchNorm statistics, parameters and their "
                            "gradients, stems' inputs, encoder outputs, heads fp32; parity: tests/test_bf16_gpu.py, "
                            "tests/test_fullsize_gpu.py::test_baseline_config2_full_model_bf16_vs_emulation") if other == "bf16" else
                           ("same loop in the REFERENCE's own dtype, fp32 everywhere (exact fp32 MFMA in the sparse convs): the "
                            "dtype of the north star's 1e-4 parity gate (tests/test_model_gpu.py, tests/test_fullsize_gpu.py)")}
            try:                                       # its own roofline (HBM-bound: algorithmic bytes at 2 B / element)
                F_.PROFILE = []
                with serial_issue(model):
                    for _ in range(args.profile_steps):
                        step_fn(model, resident, args.workload, reducer, opt)
                    torch.cuda.synchronize()
                recs, F_.PROFILE = F_.PROFILE, None
                alt["roofline"] = summarise_roofline(recs, other == "bf16")
            except Exception as e:
                F_.PROFILE = None
                alt["roofline"] = {"error": repr(e)}
        except Exception as e:                         # the extra leg must never take the headline down with it
            alt = {"dtype": other, "error": repr(e)}
        finally:
            irx.set_compute_dtype({"f32": "fp32", "bf16": "bf16", "bf16op": "bf16_operands"}[args.dtype])
        log("%s leg done: %s" % (other, alt.get("ms_per_step", alt.get("error")),))

    # ---- instrumented steps: per-launch events on the launch stream for the sparse-conv kernels ----
    roof = None
    roof_error = None
    if rank == 0 and args.profile_steps > 0:
        try:
            F_.PROFILE = []
            saved_world, opt.world_size = opt.world_size, 1     # rank-0-only steps: no collective (others are not in it)
            F_.SYNC_OFF = True                                   # ... nor a sync-BatchNorm fold (--sync-bn)
            with serial_issue(model):
                for _ in range(args.profile_steps):
                    step_fn(model, resident, args.workload, reducer, opt)
                torch.cuda.synchronize()
            opt.world_size = saved_world
            F_.SYNC_OFF = False
            recs = F_.PROFILE
            F_.PROFILE = None
            roof = summarise_roofline(recs, args.dtype != "f32")
        except Exception as e:                           # never lose the throughput line to the instrumented steps
            F_.PROFILE = None
            F_.SYNC_OFF = False
            roof_error = repr(e)

    if rank == 0:
        out = {
            "metric": "scenes/sec fwd+bwd (50k-pt synthetic ScanRefer)",
            "value": world * B * args.steps / dt, "unit": "scenes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"compute_dtype": {"f32": "fp32", "bf16": "bf16 operands + bf16 activation/gradient storage in the encoders", "bf16op": "bf16 operands, fp32 storage"}[args.dtype],
                       "workload": ({"full": "full InstanceRefer (lang+attribute+relation+scene) fwd+bwd+allreduce+Adam, "
                                             "BASELINE configs[2]/[3] shape",
                                     "attr": "BASELINE configs[1]: SparseConv3d + attribute_module only, fwd+bwd+Adam",
                                     "stress": "BASELINE configs[4]: dense-scene stress (200k pts, 64 instances, multiview "
                                               "C0 = 135), full InstanceRefer fwd+bwd+allreduce+Adam"}[args.workload]) +
                                   (", fp32" if args.dtype == "f32" else ", " + args.dtype),
                       "scenes_per_gpu": B, "global_batch": world * B, "points_per_scene": args.points,
                       "instances": args.instances, "candidates": args.candidates, "tokens": args.tokens,
                       "input_channels": 7 + args.multiview,
                       "scene_voxels_per_gpu": n_scene_vox, "parallelism": "dp%d" % world, "sync_bn": bool(args.sync_bn and world > 1), "loss": final_loss,
                       "input_prep": "inline" if args.no_pipeline else "side-stream prefetch of step N+1 during step N",
                       "primed_steps": primed,
                       "host_binding_rank0": binding},
            "roofline": roof if roof_error is None else {"error": roof_error},
        }
        for blk, ms_step in ((out["roofline"], out["ms_per_step"]), ((alt or {}).get("roofline"), (alt or {}).get("ms_per_step"))):
            if isinstance(blk, dict) and "bound_ms_total" in blk and ms_step:
                # whole-step figure: the time the sparse-conv kernels of ONE step would take at their rooflines (sum of the
                # per-kernel bounds of the instrumented steps / their count) over the measured step — what launch overhead,
                # small kernels, the dense heads and host stalls leave of the machine
                b = blk.pop("bound_ms_total") / max(args.profile_steps, 1)
                blk["step_frac_of_conv_roofline"] = b / ms_step
                blk["conv_roofline_ms_per_step"] = b
        if alt is not None:
            out["alt_dtype"] = alt
        if world == 1 and not args.no_cpu_baseline and isinstance(out["roofline"], dict) and "per_kernel" in out["roofline"]:
            try:                                        # the sparse path's other HBM-bound kernels, with their own lines
                out["roofline"]["support_kernels"] = measure_support_kernels(model, resident, args)
            except Exception as e:
                out["roofline"]["support_kernels"] = {"error": repr(e)}
        if world == 1 and args.workload != "attr" and not args.no_cpu_baseline:   # (quick runs skip the auxiliary legs)
            try:
                out["dense_path"] = measure_dense_path(model, resident, device)
            except Exception as e:
                out["dense_path"] = {"error": repr(e)}
        if world == 1 and args.workload == "full" and not args.no_e2e and not args.no_cpu_baseline:   # (quick runs skip both)
            out["end_to_end"] = end_to_end(args)
        if not args.no_cpu_baseline and world == 1:   # contract: the CPU leg runs on rank 0 at N = 1 only
            try:
                out["cpu_baseline"] = cpu_baseline(args, args.workload)
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline and roof_error is None and roof:   # (quick runs keep the committed figure)
            # LAST of the auxiliary legs: counter collection changes the GPU's clock management, and the end-to-end leg run right
            # behind it measured 17-22 % below the resident-input step instead of 1-2 %
            torch.cuda.synchronize()
            pmc = measure_pmc_traffic(args, args.dtype != "f32")
            traffic_for(out["roofline"], pmc, args.dtype != "f32")
            if isinstance(pmc, dict) and "error" in pmc:
                out["roofline"]["traffic_source"] = "profiles/r02_pmc_traffic*.json (live PMC pass failed: %s)" % pmc["error"]
            if alt is not None and isinstance(alt.get("roofline"), dict) and "kernel" in alt["roofline"]:
                traffic_for(alt["roofline"], measure_pmc_traffic(args, other == "bf16", other), other == "bf16")
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.barrier(device_ids=None if share else [device.index])
        dist.destroy_process_group()


def measure_pmc_traffic(args, bf16=False, dtype=None):
    """HBM bytes per launch of every k_* kernel, measured NOW on this box: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE —
    separate runs, with --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes) over a short child run of this
    same script (2 steps, serial issue so that dispatches are attributed cleanly), folded per kernel:
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 — both counters are in KB, and on gfx950 FETCH_SIZE reports half of a wide
    (16 B/lane) coalesced read stream (same section). -> {kernel name: bytes per launch} or {"error": ...}."""
    import csv
    import re
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return {"error": "rocprofv3 not found"}
    child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-pipeline", "--no-cpu-baseline",
             "--no-alt-dtype", "--profile-steps", "0", "--workload", args.workload, "--dtype", dtype or ("bf16" if bf16 else args.dtype)]
    if args.batch:
        child += ["--batch", str(args.batch)]
    sums = {}
    tmp = tempfile.mkdtemp(prefix="irx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", IRX_BENCH_NO_BIND="1")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [rp, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "p", "--"] + child
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd="/tmp", env=env)
            files = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")]
            if not files:
                return {"error": "%s pass wrote no counter_collection.csv (exit %d): %s" % (counter, r.returncode, (r.stderr or r.stdout)[-200:])}
            acc = {}
            for row in csv.DictReader(open(files[0])):
                if row.get("Counter_Name") != counter:
                    continue
                name = re.sub(r"^void ", "", row["Kernel_Name"])
                if not name.startswith("k_"):
                    continue
                m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", name)
                key = (m.group(1) + (m.group(2) or "")) if m else name
                a = acc.setdefault(key, [0.0, 0])
                a[0] += float(row["Counter_Value"])
                a[1] += 1
            for k, (tot, n) in acc.items():
                sums.setdefault(k, {})[counter] = tot / max(n, 1)
    except Exception as e:
        return {"error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {k: int((2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024) for k, v in sums.items()}


def traffic_for(roof, pmc, bf16):
    """Fill roof['traffic'] (HBM bytes per launch of the dominant kernel) from a live PMC measurement."""
    if not roof or not isinstance(pmc, dict) or "error" in pmc:
        return
    key = roof["kernel"].replace(",", ", ")
    if key.startswith(("k_spconv2<", "k_wgrad_pairs<")):           # template flags: <..., bf16 operands, bf16 storage>
        key = key[:-1] + (", true, true>" if bf16 else ", false, false>")
    hit = [k for k in pmc if k == key or k.startswith(key[:-1] + ",")]     # (k_spconv3<cin, cout, waves, K parts>, k_spconv4<cin, cout, waves, row sets>)
    if hit:
        roof["traffic"] = pmc[hit[0]]
        roof["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this bench.py on this box (2 steps, serial "
                                  "issue); bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB — the factor 2 is gfx950's half-count of wide "
                                  "(16 B per lane) read streams, which is what this kernel's row / image loads are; it would "
                                  "overstate kernels with narrow or scalar reads, for which no traffic figure is reported")


def end_to_end(args):
    """The same training step with the per-sample input pipeline INSIDE the loop (tools/e2e_train_bench.py in a process of
    its own, after this one's GPU work is done): reported beside `value`, which by contract times resident inputs."""
    import subprocess
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "e2e_train_bench.py")
    # (a fresh process on a GPU that idled while it imported torch: 8 warm-up + 40 timed steps read 8-13 % low against three
    #  back-to-back runs of the tool — 1488 vs 1624 / 1687 / 1721 scenes/s — so the leg now warms up 30 and times 100 steps)
    cmd = [sys.executable, tool, "--json", "--steps", "100", "--warmup", "30", "--scans", "16",
           "--dtype", "bf16" if args.dtype == "bf16" else "f32", "--batch", str(args.batch or 16)]
    try:
        torch.cuda.synchronize()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": "exit %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
        return json.loads(line[-1])
    except Exception as e:
        return {"error": repr(e)}


def measure_support_kernels(model, resident, args, reps=20):
    """The HBM-bound sparse kernels BESIDE the convolutions (north_star: "rocprof-reported HBM GB/s (sparse path)"; SURVEY 8(d)
    gives their algorithmic bytes): BatchNorm (+ shortcut) + ReLU forward / backward of every encoder layer, the 3^3 kernel
    maps (27-neighbour tables, octree descent) of every level, the Morton sort of the scene tensor and the candidate voxeliser,
    each timed ALONE with HIP events on the current stream on tensors of exactly this batch's sizes, through the same C-ABI
    entry points the step uses. -> {name: {calls_per_step, us_per_step, algo_mb_per_step, algo_gbs, frac_of_bound}} + totals.
    Algorithmic bytes (e = 2 bf16 storage / 4 fp32): BatchNorm forward 3 e N C (statistics read, apply read + write; + e N C for a
    shortcut operand), backward 6 e N C (sums pass: x, dy, y; apply pass: x, dy read, dx written; + 2 e N C with a shortcut:
    y read again, d res written); kernel map N K 8 probe bytes + 8 M pair bytes (K = 27; M = valid entries) + hash build 16 N;
    radix sort 2 x 12 B x N per pass (key + index read and written); voxelise P (24 + 4 C0) read + N (16 + 4 C0) written + 16 P
    hash traffic (xyz is float64 here: the reference's arrays, lib/dataset.py:224)."""
    import torch
    from instancerefer_amd import _lib
    from instancerefer_amd.sparse import encoder_fn, functional as F_
    from instancerefer_amd.sparse.tensor import SparseTensor
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    bf = 1 if args.dtype == "bf16" else 0
    e = 2.0 if bf else 4.0

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3            # us

    out = {}

    def add(name, us, byts, calls=1):
        a = out.setdefault(name, dict(calls_per_step=0, us_per_step=0.0, algo_bytes=0.0))
        a["calls_per_step"] += calls
        a["us_per_step"] += us
        a["algo_bytes"] += byts

    dd = model.prepare(fresh_batch(resident))
    torch.cuda.synchronize()
    encoders = []
    if getattr(model.args, "scene_module", None) and "lidar" in dd:
        encoders.append((model.scene.net, dd["lidar"].level()))
    prep = dd.get("_attr_prepared")
    if prep is not None and prep[0] is not None:
        encoders.append((model.attribute.net, prep[0].level()))
    P = lambda t: t.data_ptr() if t is not None else None
    s = _lib.stream_ptr()
    kmap_bytes = [0.0]
    for enc, lv0 in encoders:
        layers = encoder_fn.build_plan(enc, lv0)
        nl = len(layers)
        seen_levels = set()
        for i, L in enumerate(layers):
            n, c = int(L.n_out), int(L.cout)
            if n == 0:
                continue
            last = (i == nl - 1)
            y_bf = bf if not last else 0
            dt = torch.bfloat16 if bf else torch.float32
            x = torch.randn(n, c, device=dev).to(dt)
            y = torch.empty(n, c, device=dev, dtype=torch.bfloat16 if y_bf else torch.float32)
            res = torch.randn(n, c, device=dev).to(dt) if L.res >= 0 else None
            dy = torch.randn(n, c, device=dev).to(torch.bfloat16 if y_bf else torch.float32)
            dx = torch.empty_like(x)
            dres = torch.empty_like(x) if L.res >= 0 else None
            mean, inv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
            g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
            dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
            wsb = lib.irx_bn_workspace_bytes(n, c)
            ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
            tf = timed(lambda: lib.irx_bn_forward_ex(P(x), n, c, 1e-5, 0.1, P(g), P(b), P(res), 1, P(mean), P(inv), None, None, P(y),
                                                     P(ws), wsb, s, bf, bf, y_bf))
            tb = timed(lambda: lib.irx_bn_backward_ex(P(x), P(y), P(dy), n, c, P(mean), P(inv), P(g), None if L.res >= 0 else P(b), 1,
                                                      P(dx), P(dg), P(db), P(dres), P(ws), wsb, s, bf, y_bf, y_bf, bf, bf))
            add("BatchNorm(+shortcut)+ReLU forward (k_bn_partial<0> + k_bn_finalize<0> + k_bn_apply | k_bn_slice_fwd)", tf,
                (3 + (1 if L.res >= 0 else 0)) * e * n * c)
            add("BatchNorm(+shortcut)+ReLU backward (k_bn_partial<1> + k_bn_finalize<1> + k_bn_bwd_apply | k_bn_slice_bwd)", tb,
                (6 + (2 if L.res >= 0 else 0)) * e * n * c)
            del x, y, res, dy, dx, dres
            lv = L.lv_in
            if not L.down and id(lv) not in seen_levels and lv.n > 0:
                seen_levels.add(id(lv))
                tbl, _ = lv.nbr27()
                kmap_bytes[0] += lv.n * 27 * 8.0 + 8.0 * int((tbl[:27, :lv.n] >= 0).sum().item()) + 16.0 * lv.n
        # the 3^3 kernel maps of the whole pyramid the way the step builds them (round 6): ONE native call per pyramid — the coarsest
        # level by window search (k_kmap_win; + its hash table when it has more than 2048 rows), every finer level by octree descent
        # (k_kmap_descend). Same algorithmic bytes as rounds 4-5 priced (N K 8 probe bytes + 8 M pair bytes + 16 N).
        lvs, lv = [], lv0
        while lv is not None:
            lvs.append(lv)
            lv = lv._down.out_level if lv._down is not None else None
        from instancerefer_amd import _nodes
        mod = _nodes.load()
        if mod is not None and hasattr(mod, "kmaps_build_pyramid") and len(lvs) > 1 and all(l.n > 0 for l in lvs):
            dms = [l._down for l in lvs[:-1]]
            ka = ([l.keys for l in lvs], [l.coords for l in lvs], [l.stride for l in lvs], [d.parent for d in dms], [d.koff for d in dms],
                  [d.child for d in dms], [d.ld for d in dms], s)
            tk = timed(lambda: mod.kmaps_build_pyramid(*ka))
            add("kernel map 3^3, all levels of a pyramid in one call (k_kmap_win + %d x k_kmap_descend)" % (len(lvs) - 1), tk, kmap_bytes[0])
        kmap_bytes[0] = 0.0
    # Morton sort of the scene tensor (SparseTensor.canonical: key encode, radix sort, decode, feature gather)
    if "lidar_F" in resident:
        Fr, Cr, Bn = resident["lidar_F"], resident["lidar_C"], resident["B"]
        n = int(Cr.shape[0])
        bits = F_.morton_bits(Bn)
        passes = (bits + 7) // 8
        ts = timed(lambda: SparseTensor(Fr, Cr, 1, batch_size=Bn).canonical())
        add("scene Morton sort (k_coords_to_keys + %d x (k_rs_count + k_rs_rowscan + k_rs_scatter) + decode + row gather)" % passes, ts,
            passes * 2 * 12.0 * n + n * (16 + 8 + 16) + 2.0 * 4 * n * Fr.shape[1])
    # candidate voxeliser (quantise, hash de-duplication, select, sort, pyramid) through the attribute module's own entry
    if prep is not None and prep[0] is not None and hasattr(model.attribute, "prepare_launch"):
        host = resident.get("_host", {})
        cls_list = [int(v) for v in (host["object_cat"] if "object_cat" in host else resident["object_cat"].tolist())]
        nvox = int(prep[0].F.shape[0])
        c0 = int(prep[0].F.shape[1])
        ppi = int(resident["irx"].pts32.shape[1]) if hasattr(resident.get("irx"), "pts32") else 1024
        npts = len(prep[1]["cand"]) * ppi

        def vox():
            # what the step's preparation stage enqueues (no host sync: the voxel count and the level sizes travel to pinned memory and
            # are read by prepare_finish() a step later); rounds 4-5 timed launch + finish back to back, i.e. a host round trip per call
            d2 = fresh_batch(resident)
            model.attribute.prepare_launch(d2, cls_list)
        tv = timed(vox)
        add("candidate voxeliser + pyramid (k_quantize, k_voxel_insert / _select, k_rs_*, k_ds_*; sync-free launch)", tv,
            npts * (24 + 4.0 * c0) + nvox * (16 + 4.0 * c0) + 16.0 * npts)
    tot_us = tot_b = 0.0
    res = {}
    for k, a in out.items():
        bound_us = a["algo_bytes"] / (PEAK_HBM_GBS * 1e9) * 1e6
        res[k] = {"calls_per_step": a["calls_per_step"], "us_per_step": round(a["us_per_step"], 1),
                  "algo_mb_per_step": round(a["algo_bytes"] / 1e6, 1),
                  "algo_gbs": round(a["algo_bytes"] / (a["us_per_step"] * 1e-6) / 1e9, 1) if a["us_per_step"] else None,
                  "frac_of_bound": round(bound_us / a["us_per_step"], 4) if a["us_per_step"] else None}
        if k.startswith("BatchNorm"):
            tot_us += a["us_per_step"]
            tot_b += a["algo_bytes"]
    if tot_us:
        res["BatchNorm total"] = {"us_per_step": round(tot_us, 1), "algo_mb_per_step": round(tot_b / 1e6, 1),
                                  "algo_gbs": round(tot_b / (tot_us * 1e-6) / 1e9, 1),
                                  "frac_of_bound": round(tot_b / (PEAK_HBM_GBS * 1e9) * 1e6 / tot_us, 4)}
    res["note"] = ("each operator timed alone (HIP events, %d repetitions) on tensors of this batch's sizes through the C-ABI entry points "
                   "the step calls; frac_of_bound = algorithmic bytes / 8 TB/s over the measured time. Most launches here are far below the "
                   "size at which a kernel can reach HBM speed: three dependent launches cost ~12 us whatever the tensor" % reps)
    return res


def measure_dense_path(model, resident, device, reps=20):
    """north_star's dense path (language encoder: word-projection GEMMs, 2-layer GRU, attention pooling, classifier),
    forward + backward, timed alone with events on the current stream; FLOPs from the module's own dimensions
    (backward = 2 x forward). These are (16 x 30)-row GEMMs and a 30-step serial recurrence: the figure documents how
    little of the matrix core such shapes can use, it is not a tuning target."""
    import torch
    lang = getattr(model, "lang", None)
    if lang is None:
        return None
    feat, length = resident["lang_feat"], resident["lang_len"]
    Bn = int(feat.shape[0])
    T = int(resident.get("lang_len_max", int(length.max().item())))
    H = lang.gru.hidden_size
    nd = 2 if lang.use_bidir else 1
    lin = [m for m in lang.word_projection if isinstance(m, torch.nn.Linear)]
    fwd = sum(2.0 * Bn * T * m.in_features * m.out_features for m in lin)
    for layer in range(lang.gru.num_layers):
        inp = lang.gru.input_size if layer == 0 else H * nd
        fwd += nd * 2.0 * Bn * T * 3 * H * (inp + H)
    fwd += 2.0 * Bn * T * H * nd * 4 + 2.0 * Bn * 4 * T * lin[-1].out_features
    if lang.use_lang_classifier:
        fwd += 2.0 * Bn * lang.lang_cls[0].in_features * lang.lang_cls[0].out_features
    params = [q for q in lang.parameters() if q.requires_grad]

    def once():
        dd = lang({"lang_feat": feat, "lang_len": length, "lang_len_max": T})
        loss = (dd["lang_attr_feats"].sum() + dd["lang_rel_feats"].sum() + dd["lang_scene_feats"].sum() +
                dd["lang_scores"].sum())
        torch.autograd.grad(loss, params, allow_unused=True)

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = 3.0 * fwd / (ms * 1e-3) / 1e12
    return {"what": "LangModule forward + backward alone (word projection, %d-layer GRU(%d), attention pooling, classifier); "
                    "host-issue time included" % (lang.gru.num_layers, H),
            "sentences": Bn, "tokens": T, "gflop": 3.0 * fwd / 1e9, "ms": ms, "achieved": tf, "unit": "TFLOP/s",
            "peak": 157.3, "frac": tf / 157.3}


class serial_issue:
    """Instrumented steps only: both encoders on the main stream, issued inline (no library lanes), so that an event
    bracket times its kernel alone on the GPU — the quantity a kernel's roofline fraction is about, and what rocprofv3's
    per-kernel duration measures; in the timed loop the two encoders' kernels overlap on two streams."""

    def __init__(self, model):
        self.model = model

    def __enter__(self):
        from instancerefer_amd.sparse import encoder_fn
        self.saved = (getattr(self.model.args, "overlap_streams", True), encoder_fn.ASYNC)
        self.model.args.overlap_streams, encoder_fn.ASYNC = False, False

    def __exit__(self, *exc):
        from instancerefer_amd.sparse import encoder_fn
        self.model.args.overlap_streams, encoder_fn.ASYNC = self.saved
        return False


def summarise_roofline(recs, bf16=False):
    """recs: (kind, n_out, K, cin, cout, M, start_event, end_event), one per C-ABI conv call, events recorded on the
    launch stream. The dominant HIP kernel is k_spconv2<128,128> (forward + data-gradient of every 128->128 layer:
    the largest total in the rocprofv3 kernel stats). Algorithmic figures per launch (SURVEY §8d formula A, e = 4):
        FLOPs = 2*M*Cin*Cout ;  Bytes_A = 4*(M*Cin + N_out*Cout + K*Cin*Cout) + 8*M      (wgrad: 4*(M*(Cin+Cout) + ...))
    with M counted from the actual tables (valid columns only) and bounded by n_out*K — a pair count above that bound
    would silently inflate the roofline figure, so it is an error. achieved = sum(algorithmic flops or bytes of the
    binding resource) / sum(measured time); the events bracket the dominant kernel only (irx_profile_next_kernel)."""
    agg = {}

    try:
        from instancerefer_amd import _lib
        v3 = bool(_lib.get_knob("spconv3"))
        w3 = bool(_lib.get_knob("wgrad3"))
        v4 = int(_lib.get_knob("spconv4"))
    except Exception:
        v3 = w3 = False
        v4 = 0

    def klass(kind, cin, cout, e=4.0):
        if kind in ("fwd", "dgrad"):
            if cin in (32, 64, 128) and cout in (32, 64, 128):
                # a bf16 INPUT (bf16 storage inside the executor) runs on the third-generation kernel (csrc/irx_spconv3.hip)
                # ... the 128 -> 128 layers on the fourth (k_spconv4: LDS-DMA, row-shaped gathers; knob spconv4 = 3: every 64- /
                # 128-input-channel shape)
                if bf16 and e == 2.0 and v3 and cin * cout >= 2048:
                    g4 = v4 and cin in (64, 128) and (v4 == 3 or (cin == 128 and cout == 128))
                    return ("k_spconv4<%d,%d>" if g4 else "k_spconv3<%d,%d>") % (cin, cout)
                return "k_spconv2<%d,%d>" % (cin, cout)
            if 128 < cin <= 136 and cout == 32 and kind == "fwd":
                return "wide stem fwd (k_stem_fwd + k_spconv2<128,32>)"
            return "k_stem_fwd" if (cin <= 8 and cout == 32) else "k_spconv_fwd(generic)"
        if cin in (32, 64, 128) and cout in (32, 64, 128):
            # bf16 rows with 64 / 128 channels both ways run on k_wgrad3 (csrc/irx_pairs.hip)
            return ("k_wgrad3<%d,%d>" if (bf16 and e == 2.0 and w3 and cin >= 64 and cout >= 64) else "k_wgrad_pairs<%d,%d>") % (cin, cout)
        if 128 < cin <= 136 and cout == 32:
            return "wide stem wgrad (k_spconv2_wgrad<128,32> + k_stem_wgrad)"
        return "k_stem_wgrad" if (cin <= 8 and cout == 32) else "k_spconv_wgrad(generic)"

    tot = dict(ms=0.0, bound_ms=0.0)
    for rec in recs:
        kind, n_out, K, cin, cout, M, e0, e1 = rec[:8]
        e = float(rec[8]) if len(rec) > 8 else 4.0    # bytes per activation element (2 with bf16 storage)
        ms = e0.elapsed_time(e1)
        if not 0 <= M <= max(n_out, 1) * K * 8:       # (dgrad records carry the forward table's pair count)
            raise RuntimeError("roofline: pair count %d of a (%d rows, %d offsets) table is impossible" % (M, n_out, K))
        flops = 2.0 * M * cin * cout
        ew = 2.0 if (bf16 and cin in (32, 64, 128) and cout in (32, 64, 128)) else 4.0     # weight image element
        if kind == "wgrad":
            byts = e * M * (cin + cout) + 4.0 * K * cin * cout + 8.0 * M
        else:
            byts = e * (M * cin + n_out * cout) + ew * K * cin * cout + 8.0 * M
        kl = klass(kind, cin, cout, e)
        a = agg.setdefault(kl, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, bound_ms=0.0))
        a["ms"] += ms
        a["flops"] += flops
        a["bytes"] += byts
        a["launches"] += 1
        a["peak_tf"] = peak_tf = PEAK_BF16_TFLOPS if (bf16 and kl.startswith(("k_spconv2<", "k_spconv3<", "k_spconv4<", "k_wgrad_pairs<", "k_wgrad3<"))) else PEAK_F32_TFLOPS
        b_ms = max(byts / (PEAK_HBM_GBS * 1e9), flops / (peak_tf * 1e12)) * 1e3
        a["bound_ms"] += b_ms
        tot["ms"] += ms
        tot["bound_ms"] += b_ms
    if os.environ.get("IRX_BENCH_LAYERS"):
        lay = {}
        # one line per distinct layer shape, priced with EXACTLY the byte formula of the driver line above (element sizes from the
        # record's storage dtype: 2 B with bf16 storage — round 5's table priced every element at 4 B and printed figures above
        # the HBM peak)
        for rec in recs:
            kind, n_out, K, cin, cout, M, e0, e1 = rec[:8]
            e = float(rec[8]) if len(rec) > 8 else 4.0
            a = lay.setdefault((kind, n_out, K, cin, cout, M, e), [0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1)
        print("kind   n_out      K  cin cout        M  eB   calls  avg_us   TFLOP/s  algoGB/s  frac_hbm", file=sys.stderr)
        for (kind, n_out, K, cin, cout, M, e), (c, ms) in sorted(lay.items(), key=lambda kv: -kv[1][1]):
            us = 1e3 * ms / c
            fl = 2.0 * M * cin * cout
            ew = 2.0 if (bf16 and cin in (32, 64, 128) and cout in (32, 64, 128)) else 4.0
            if kind == "wgrad":
                by = e * M * (cin + cout) + 4.0 * K * cin * cout + 8.0 * M
            else:
                by = e * (M * cin + n_out * cout) + ew * K * cin * cout + 8.0 * M
            print("%-6s %7d %4d %4d %4d %9d %3d %6d %8.1f %8.2f %9.1f %9.3f" % (kind, n_out, K, cin, cout, M, int(e), c, us, fl / us / 1e6,
                                                                          by / us / 1e3, by / us / 1e3 / PEAK_HBM_GBS), file=sys.stderr)
    if not agg:
        return None
    dom = max(agg, key=lambda k: agg[k]["ms"])
    a = agg[dom]
    tf = a["flops"] / (a["ms"] * 1e-3) / 1e12
    gbs = a["bytes"] / (a["ms"] * 1e-3) / 1e9
    peak_tf = a["peak_tf"]
    mfma_bound = a["flops"] / (peak_tf * 1e12) >= a["bytes"] / (PEAK_HBM_GBS * 1e9)
    per_kernel = {k: {"launches": v["launches"], "avg_us": round(1e3 * v["ms"] / v["launches"], 2),
                      "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                      "algo_gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                      "frac_of_bound": round(v["bound_ms"] / v["ms"], 4)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
    traffic = None
    try:   # HBM bytes per launch from the committed PMC passes (profiles/r02_pmc_traffic*.json; see their _note)
        name = "r02_pmc_traffic_bf16.json" if bf16 else "r02_pmc_traffic.json"
        pmc = json.load(open(os.path.join(ROOT, "profiles", name)))["kernels"]
        key = dom.replace(",", ", ")
        if key.startswith(("k_spconv2<", "k_wgrad_pairs<")):       # template flags: <..., bf16 operands, bf16 storage>
            key = key[:-1] + (", true, true>" if bf16 else ", false, false>")
        if key in pmc:
            traffic = pmc[key]["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return {"kernel": dom, "bound": "mfma" if mfma_bound else "hbm", "traffic_source": "profiles/%s (committed PMC passes)" % name if traffic is not None else None,
            "achieved": tf if mfma_bound else gbs, "peak": peak_tf if mfma_bound else PEAK_HBM_GBS,
            "unit": "TFLOP/s" if mfma_bound else "GB/s",
            "frac": (tf / peak_tf) if mfma_bound else (gbs / PEAK_HBM_GBS),
            "traffic": traffic, "algorithmic_bytes_per_launch": a["bytes"] / a["launches"],
            "algorithmic_flops_per_launch": a["flops"] / a["launches"],
            "avg_launch_us": 1e3 * a["ms"] / a["launches"], "launches": a["launches"],
            "path_frac_of_roofline": tot["bound_ms"] / tot["ms"], "bound_ms_total": tot["bound_ms"], "per_kernel": per_kernel,
            "per_kernel_note": ("frac_of_bound = roofline time of the kernel's ALGORITHMIC bytes / flops (SURVEY 8(d): every gathered row "
                                "counted once per pair) over its measured time. The pair-list weight gradients (k_wgrad3 / k_wgrad_pairs) "
                                "re-read each row up to 27 x, mostly from L2 (XCD-segment work units), so their figure can approach 1 "
                                "against the HBM peak without the kernel being HBM-bound; the judged object is the dominant kernel above")}


if __name__ == "__main__":
    main()
