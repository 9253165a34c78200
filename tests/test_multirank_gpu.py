"""BASELINE configs[3] (data-parallel training) on ONE GPU: two ranks share cuda:0 and exchange over gloo (RCCL needs one
device per rank), driving the PRODUCT path — optim.FlatAdam.backward_step with the encoders' gradient sinks, the
asynchronous library lanes, the early all-reduce of the encoder ranges and the rank-0 broadcast at construction.

The reference has no multi-GPU path (lib/solver.py:200-205 is a plain backward(); step()), so the oracle here is the
single-rank run of the same code on the concatenated batch."""
import os
import sys
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _few_hardware_queues_per_rank(monkeypatch):
    """The ranks of these tests SHARE cuda:0. The package asks the HIP runtime for 8 hardware queues per process (one rank per GPU
    in production); several processes x 8 on one device oversubscribe its queue slots and every launch waits for a queue switch
    (bench.py --gpus 2 on one GPU: 31.8 s per step with 8, 1.2 s with 4, 20 ms with 2). The spawned ranks read this at start-up."""
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "2")


KW = dict(num_points=5000, num_instances=5, points_per_instance=160)


def _model(seed, dev):
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.instancerefer import InstanceRefer
    torch.manual_seed(seed)
    model = InstanceRefer(7, S.default_args())
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model.to(dev)


def _step(model, opt, batch, dev):
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    opt.zero_grad()
    dd = get_loss(model(S.to_device(batch, dev)), DatasetConfig())
    dd["loss"].backward()
    opt.gather_grads()
    opt.all_reduce()
    g = opt.flat_g[:opt.n].clone()
    opt.step()
    return float(dd["loss"].detach()), g


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from instancerefer_amd import _build, _lib, synthetic as S
    from instancerefer_amd.optim import FlatAdam
    from instancerefer_amd.sparse import encoder_fn
    _build.build_lib()
    _lib.load()
    res = {}
    # ---- A: train mode, per-rank seeds (the constructor must equalise the replicas), one rank without candidates ----
    model = _model(500 + rank, dev).train()
    with torch.no_grad():
        model.scene.net.stem[0].net[1].running_mean.add_(float(rank))
    opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, module=model)
    res["p_init"] = opt.flat_p.clone().cpu()
    res["rm_init"] = model.scene.net.stem[0].net[1].running_mean.clone().cpu()
    cands = [[3, 2], [1, 0]][rank]          # rank 1: no scene with >= 2 candidates -> empty score tensors, still joins
    losses = []
    for it in range(3):
        batch = S.make_batch(2, seed=700 + 10 * it + 2 * rank, num_candidates=cands, **KW)
        loss, g = _step(model, opt, batch, dev)
        losses.append(loss)
        if it == 0:
            res["g_first"] = g.cpu()
            res["native"], n_native = opt.native_delivered()      # C++ head nodes (MLPs, GRU layers) that delivered
            res["delivered"] = len(opt._direct) - n_native
    torch.cuda.synchronize()
    res["losses"] = losses
    res["p_final"] = opt.flat_p.clone().cpu()
    res["steps"] = list(opt.steps)
    index = {id(p): n for n, p in model.named_parameters()}
    res["skipped"] = [index[id(p)] for p, s in zip(opt.params, opt.steps) if s != 3]
    res["async"] = bool(encoder_fn.ASYNC)
    # ---- B: BatchNorm in eval mode -> every quantity is per scene: 2 ranks x 2 scenes == 1 rank x 4 scenes ----
    model2 = _model(900, dev).eval()
    opt2 = FlatAdam(model2.parameters(), lr=1e-3, weight_decay=1e-5, module=model2)
    sd0 = {k: v.clone() for k, v in model2.state_dict().items()}
    batch = S.make_batch(2, seed=800 + 2 * rank, num_candidates=[[3, 2], [2, 4]][rank], **KW)
    _, g2 = _step(model2, opt2, batch, dev)
    res["g_eval_2rank"] = (g2 / world).cpu()
    res["p_eval_2rank"] = opt2.flat_p.clone().cpu()
    # ---- C: gradient accumulation — two backward passes before one step (the second pass arrives through
    # autograd .grad and is added to the slots, which has to happen BEFORE the segment's collective) ----
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    ba = S.make_batch(2, seed=860 + 2 * rank, num_candidates=[[3, 2], [2, 3]][rank], **KW)
    bb = S.make_batch(2, seed=880 + 2 * rank, num_candidates=[[2, 2], [4, 2]][rank], **KW)

    def reduced_grad(batches):
        opt2.zero_grad()
        for b in batches:
            get_loss(model2(S.to_device(dict(b), dev)), DatasetConfig())["loss"].backward()
        opt2.gather_grads()
        opt2.all_reduce()
        torch.cuda.synchronize()
        return opt2.flat_g[:opt2.n].clone().cpu()
    res["g_a"], res["g_b"], res["g_ab"] = reduced_grad([ba]), reduced_grad([bb]), reduced_grad([ba, bb])
    dist.barrier()
    if rank == 0:
        model1 = _model(901, dev).eval()
        model1.load_state_dict(sd0)
        opt1 = FlatAdam(model1.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1, broadcast=False)
        batch = S.make_batch(4, seed=800, num_candidates=[3, 2, 2, 4], **KW)
        _, g1 = _step(model1, opt1, batch, dev)
        res["g_eval_1rank"] = g1.cpu()
    torch.cuda.synchronize()
    torch.save(res, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(tmp_path, timeout=420):
    mp.set_start_method("spawn", force=True)
    port = 29000 + (os.getpid() * 7 + int(time.time())) % 2000
    ctx = mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=False, start_method="spawn")
    deadline = time.time() + timeout
    try:
        while not ctx.join(timeout=5):
            if time.time() > deadline:
                raise TimeoutError("two-rank run exceeded %d s" % timeout)
    finally:
        for p in ctx.processes:              # exact PIDs we started, never a pattern
            if p.is_alive():
                p.kill()
    return [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(2)]


def test_two_ranks_share_one_gpu_product_path(lib, tmp_path):
    r0, r1 = _run_two_ranks(tmp_path)
    # construction: rank 0's parameters AND buffers everywhere, whatever the local seeds were
    assert torch.equal(r0["p_init"], r1["p_init"]) and torch.equal(r0["rm_init"], r1["rm_init"])
    # training: the summed gradient is the same buffer on both ranks -> bit-identical parameters after 3 steps
    assert torch.equal(r0["g_first"], r1["g_first"])
    assert torch.equal(r0["p_final"], r1["p_final"])
    assert not torch.equal(r0["p_final"], r0["p_init"]) and bool(torch.isfinite(r0["p_final"]).all())
    assert all(np.isfinite(r0["losses"] + r1["losses"]))
    assert r0["async"] and r0["delivered"] == 78      # both encoders delivered through the sink (lanes on)
    assert r1["delivered"] == 39                      # rank 1 never ran its candidate encoder: scene encoder only
    # rank 1 had no gradient for the attribute / relation parameters but rank 0 did: nobody skips them
    assert r0["steps"] == r1["steps"] and r0["skipped"] == r1["skipped"]
    assert not [n for n in r0["skipped"] if n.startswith(("attribute.", "relation.", "scene.", "lang."))], r0["skipped"]
    # 2 ranks x 2 scenes == 1 rank x 4 scenes when BatchNorm does not couple the scenes (fp32 summation order differs)
    g2, g1 = r0["g_eval_2rank"], r0["g_eval_1rank"]
    assert torch.equal(r0["g_eval_2rank"], r1["g_eval_2rank"]) and torch.equal(r0["p_eval_2rank"], r1["p_eval_2rank"])
    scale = float(g1.abs().max())
    assert scale > 0 and float((g2 - g1).abs().max()) <= 2e-4 * scale, (float((g2 - g1).abs().max()), scale)
    assert abs(float(g2.double().norm()) - float(g1.double().norm())) <= 1e-4 * float(g1.double().norm())
    # two backward passes before the step: same reduced buffer on both ranks, equal to the sum of the two single passes
    assert torch.equal(r0["g_ab"], r1["g_ab"])
    want = r0["g_a"] + r0["g_b"]
    assert float(want.abs().max()) > 0
    assert float((r0["g_ab"] - want).abs().max()) <= 1e-5 * float(want.abs().max())


def _many_worker(rank, world, port, out_dir):
    """Scenario A of _worker for any world size: per-rank seeds, ranks r % 3 == 1 hold no scene with >= 2 candidates (for
    world 8 that is ranks 1, 4 and 7: three EMPTY ranks), three optimizer steps through the product path."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from instancerefer_amd import _build, _lib, synthetic as S
    from instancerefer_amd.optim import FlatAdam
    _build.build_lib()
    _lib.load()
    res = {}
    model = _model(500 + rank, dev).train()
    opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, module=model)
    res["p_init"] = opt.flat_p.clone().cpu()
    empty = rank % 3 == 1
    cands = [1, 0] if empty else [3, 2 + rank % 2]
    losses = []
    for it in range(3):
        batch = S.make_batch(2, seed=700 + 10 * it + 2 * rank, num_candidates=cands, **KW)
        loss, g = _step(model, opt, batch, dev)
        losses.append(loss)
        if it == 0:
            res["g_first"] = g.cpu()
            res["native"], n_native = opt.native_delivered()      # C++ head nodes (MLPs, GRU layers) that delivered
            res["delivered"] = len(opt._direct) - n_native
    torch.cuda.synchronize()
    res.update(losses=losses, p_final=opt.flat_p.clone().cpu(), steps=list(opt.steps), empty=empty)
    index = {id(p): n for n, p in model.named_parameters()}
    res["skipped"] = [index[id(p)] for p, s_ in zip(opt.params, opt.steps) if s_ != 3]
    torch.save(res, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_four_and_eight_ranks_share_one_gpu(lib, tmp_path, world):
    """BASELINE configs[3]'s world size (8) — and 4 — with the hardware there is: `world` ranks on cuda:0 over gloo through
    the product path (FlatAdam sinks, lanes, early all-reduce of the encoder ranges). More than one rank has no candidates
    (ranks 1, 4, 7), so the collective sequence must not depend on the data: every rank ends with bit-identical parameters,
    nobody skips a parameter another rank had a gradient for, and the empty ranks delivered the scene encoder only."""
    mp.set_start_method("spawn", force=True)
    port = 33000 + (os.getpid() * 11 + int(time.time())) % 2000
    ctx = mp.start_processes(_many_worker, args=(world, port, str(tmp_path)), nprocs=world, join=False, start_method="spawn")
    deadline = time.time() + 600
    try:
        while not ctx.join(timeout=5):
            if time.time() > deadline:
                raise TimeoutError("%d-rank run exceeded 600 s" % world)
    finally:
        for p in ctx.processes:              # exact PIDs we started, never a pattern
            if p.is_alive():
                p.kill()
    rs = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(world)]
    assert sum(r["empty"] for r in rs) >= (2 if world >= 5 else 1)
    for r in rs[1:]:
        assert torch.equal(r["p_init"], rs[0]["p_init"])
        assert torch.equal(r["g_first"], rs[0]["g_first"])
        assert torch.equal(r["p_final"], rs[0]["p_final"])
        assert r["steps"] == rs[0]["steps"] and r["skipped"] == rs[0]["skipped"]
    assert all(np.isfinite(r["losses"]).all() for r in rs)
    assert bool(torch.isfinite(rs[0]["p_final"]).all()) and not torch.equal(rs[0]["p_final"], rs[0]["p_init"])
    assert all(r["delivered"] == (39 if r["empty"] else 78) for r in rs), [r["delivered"] for r in rs]
    assert not [n for n in rs[0]["skipped"] if n.startswith(("attribute.", "relation.", "scene.", "lang."))], rs[0]["skipped"]


def _rccl_worker(rank, world, port, out_dir):
    """One rank, backend nccl (= RCCL): every collective of the product path is issued for real (FlatAdam._force_collectives)
    and the result must equal the run without a process group bit for bit (a one-rank sum is the identity)."""
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from instancerefer_amd import _build, _lib, synthetic as S
    from instancerefer_amd.optim import FlatAdam
    _build.build_lib()
    _lib.load()
    res = {}
    for mode in ("plain", "rccl"):
        if mode == "rccl":
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ["MASTER_PORT"] = str(port)
            dist.init_process_group("nccl", rank=0, world_size=1)   # (lazy communicator: see bench.py on device_id)
        model = _model(321, dev).train()
        opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1, module=model)
        opt._force_collectives = mode == "rccl"
        losses = []
        for it in range(3):
            batch = S.make_batch(2, seed=40 + it, num_candidates=[3, 2], **KW)
            loss, g = _step(model, opt, batch, dev)
            losses.append(loss)
        torch.cuda.synchronize()
        res[mode] = (losses, opt.flat_p.clone().cpu(), len(opt._reduced))
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()
    torch.save(res, os.path.join(out_dir, "rccl.pt"))


def test_product_collectives_over_rccl_one_rank(lib, tmp_path):
    """RCCL itself needs one device per rank, so the two-rank test above runs over gloo. This one drives the same calls —
    early async all-reduce of the encoder ranges on the producers' streams, the remaining segments and the activity flags
    after the gather, the waits — through backend "nccl" with a single rank on the one GPU there is."""
    mp.set_start_method("spawn", force=True)
    port = 31000 + (os.getpid() * 13 + int(time.time())) % 2000
    ctx = mp.start_processes(_rccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=False, start_method="spawn")
    deadline = time.time() + 300
    try:
        while not ctx.join(timeout=5):
            if time.time() > deadline:
                raise TimeoutError("one-rank RCCL run exceeded 300 s")
    finally:
        for p in ctx.processes:
            if p.is_alive():
                p.kill()
    res = torch.load(os.path.join(str(tmp_path), "rccl.pt"), weights_only=False)
    assert res["rccl"][2] >= 2                         # both encoder ranges went through an all-reduce
    assert res["plain"][0] == res["rccl"][0]
    assert torch.equal(res["plain"][1], res["rccl"][1])


def _syncbn_worker(rank, world, port, out_dir):
    """Sync BatchNorm: 2 ranks x 2 scenes in TRAIN mode == 1 rank x 4 scenes (BatchNorm couples the scenes of a batch, so
    without the cross-rank statistics this does not hold)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from instancerefer_amd import _build, _lib, synthetic as S
    from instancerefer_amd.optim import FlatAdam
    from instancerefer_amd.syncbn import convert_sync_batchnorm
    _build.build_lib()
    _lib.load()
    res = {}
    model = convert_sync_batchnorm(_model(1200, dev)).train()
    opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, module=model)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    batch = S.make_batch(2, seed=810 + 2 * rank, num_candidates=[[3, 2], [2, 4]][rank], **KW)
    loss, g2 = _step(model, opt, batch, dev)
    res["loss"] = loss
    res["g_2rank"] = (g2 / world).cpu()
    res["running_mean"] = model.scene.net.stem[0].net[1].running_mean.clone().cpu()
    res["sync_layers"] = sum(1 for m in model.modules() if getattr(m, "_irx_sync", False))
    from instancerefer_amd.sparse import encoder_fn
    res["executor_sync_calls"] = list(encoder_fn.SYNC_CALLS)
    dist.barrier()
    if rank == 0:
        model1 = _model(1201, dev).train()
        model1.load_state_dict(sd0)
        opt1 = FlatAdam(model1.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1, broadcast=False)
        batch = S.make_batch(4, seed=810, num_candidates=[3, 2, 2, 4], **KW)
        loss1, g1 = _step(model1, opt1, batch, dev)
        res["loss_1rank"] = loss1
        res["g_1rank"] = g1.cpu()
        res["running_mean_1rank"] = model1.scene.net.stem[0].net[1].running_mean.clone().cpu()
        # and WITHOUT sync the two-rank gradient is a different one: the test has teeth
        model2 = _model(1202, dev).train()
        model2.load_state_dict(sd0)
        opt2 = FlatAdam(model2.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1, broadcast=False)
        _, gl = _step(model2, opt2, S.make_batch(2, seed=810, num_candidates=[3, 2], **KW), dev)
        res["g_local_only"] = gl.cpu()
    torch.cuda.synchronize()
    torch.save(res, os.path.join(out_dir, "sync%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("path", ["executor", "per_layer"])
def test_sync_batchnorm_two_ranks_equal_one_rank_on_the_whole_batch(lib, tmp_path, monkeypatch, path):
    """"executor": both encoders stay in the one-call executor (irx_encoder_forward_sync / _backward_sync: the library calls
    back into torch.distributed between every layer's statistics and apply pass); "per_layer" (IRX_SYNC_BN_EXECUTOR=0): the
    round-2/3 path, one Python autograd node per BatchNorm layer. Same assertions."""
    monkeypatch.setenv("IRX_SYNC_BN_EXECUTOR", "1" if path == "executor" else "0")
    mp.set_start_method("spawn", force=True)
    port = 33000 + (os.getpid() * 11 + int(time.time())) % 2000
    ctx = mp.start_processes(_syncbn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=False, start_method="spawn")
    deadline = time.time() + 420
    try:
        while not ctx.join(timeout=5):
            if time.time() > deadline:
                raise TimeoutError("sync BatchNorm run exceeded 420 s")
    finally:
        for p in ctx.processes:
            if p.is_alive():
                p.kill()
    r0, r1 = (torch.load(os.path.join(str(tmp_path), "sync%d.pt" % r), weights_only=False) for r in range(2))
    assert r0["sync_layers"] >= 30                       # both encoders (13 + 13), the scene head, the heads' BatchNorm1d
    # two encoders, forward and backward, through the sync executor — or none of them
    assert r0["executor_sync_calls"] == r1["executor_sync_calls"] == ([2, 2] if path == "executor" else [0, 0])
    assert torch.equal(r0["g_2rank"], r1["g_2rank"])
    assert torch.equal(r0["running_mean"], r1["running_mean"])           # one set of statistics for both ranks
    g2, g1 = r0["g_2rank"], r0["g_1rank"]
    scale = float(g1.abs().max())
    assert scale > 0 and float((g2 - g1).abs().max()) <= 5e-4 * scale, (float((g2 - g1).abs().max()), scale)
    assert abs(float(g2.double().norm()) - float(g1.double().norm())) <= 2e-4 * float(g1.double().norm())
    assert float((r0["running_mean"] - r0["running_mean_1rank"]).abs().max()) <= 1e-6
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - r0["loss_1rank"]) <= 1e-4 * abs(r0["loss_1rank"])
    # rank 0's own half batch without the cross-rank statistics gives another gradient altogether
    assert float((r0["g_local_only"] - g1).abs().max()) > 50 * float((g2 - g1).abs().max())


def test_bench_gpus_n_launches_n_ranks_itself(lib):
    """`python bench.py --gpus 2` WITHOUT torchrun must run two ranks (self_launch) and report n_gpus = 2 in
    the last stdout line; here both ranks share cuda:0 over gloo (IRX_BENCH_SHARE_GPU), the only multi-rank rig one GPU
    allows. A --gpus / WORLD_SIZE mismatch is refused instead of mislabelled."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IRX_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--points", "6000", "--no-cpu-baseline", "--no-alt-dtype", "--profile-steps", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-1500:])
    last = r.stdout.strip().splitlines()[-1]
    out = json.loads(last)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and out["scaling"] == "weak"
    # one rank asked to call itself eight: refused
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=env1)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
