"""End-to-end GPU parity of the drop-in InstanceRefer (HIP kernels through the C-ABI) against
tests/golden/model.npz — outputs of the REFERENCE's models/instancerefer.py + lib/loss_helper.py run on the
same seeded scenes and weights (tests/golden/make_golden.py). Tolerance: 1e-4 absolute on every forward
tensor (north star), gradients 1e-3 relative to the gradient norm."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN_CFG, WEIGHT_SEED

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FWD_KEYS = ("lang_scores", "lang_cls_feats", "lang_attr_feats", "lang_rel_feats", "lang_scene_feats", "atten_attr",
            "atten_rel", "atten_scene", "obj_feats", "attribute_scores", "relation_scores", "scene_scores",
            "seg_scores", "vis_atten", "loss", "ref_loss", "lang_loss", "seg_loss", "seg_acc")


def _build(mode):
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.instancerefer import InstanceRefer
    dev = torch.device("cuda")
    model = InstanceRefer(7, S.default_args())
    model.load_state_dict(S.seeded_state_dict(model, WEIGHT_SEED))
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.to(dev).train(mode == "train")
    dd = S.to_device(S.make_batch(**dict(GOLDEN_CFG)), dev)
    return model, dd


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_forward_matches_reference(lib, mode):
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    gold = np.load(os.path.join(G, "model.npz"))
    model, dd = _build(mode)
    with torch.set_grad_enabled(mode == "train"):
        dd = get_loss(model(dd), DatasetConfig())
    assert list(dd["num_filtered_objs"]) == gold[mode + "/num_filtered_objs"].tolist()
    pob = np.concatenate([p.reshape(-1, 7) for p in dd["pred_obb_batch"]], 0)
    assert np.array_equal(pob, gold[mode + "/pred_obb_batch"])
    lab = np.concatenate([c.cpu().numpy() if len(c) else np.zeros(0) for c in dd["cluster_label"]])
    assert np.array_equal(lab, gold[mode + "/cluster_label"])
    worst = {}
    for k in FWD_KEYS:
        got = dd[k].detach().float().cpu().numpy()
        exp = gold["%s/%s" % (mode, k)]
        assert got.shape == exp.shape, (k, got.shape, exp.shape)
        worst[k] = float(np.abs(got - exp).max()) if got.size else 0.0
    bad = {k: v for k, v in worst.items() if not v <= 1e-4}
    assert not bad, bad


def test_backward_matches_reference(lib):
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    gold = np.load(os.path.join(G, "model.npz"))
    model, dd = _build("train")
    dd = get_loss(model(dd), DatasetConfig())
    dd["loss"].backward()
    params = dict(model.named_parameters())
    bad = {}
    total = float(np.sqrt(sum(float(gold[k]) ** 2 for k in gold.files if k.startswith("grad_norm/"))))
    for k in gold.files:
        if k.startswith("grad_norm/"):
            name = k[len("grad_norm/"):]
            g = params[name].grad
            assert g is not None, name
            got = float(g.double().norm())
            exp = float(gold[k])
            if abs(got - exp) > 1e-3 * max(exp, 1e-3 * total):
                bad[name] = (got, exp)
        elif k.startswith("grad/"):
            name = k[len("grad/"):]
            got = params[name].grad.cpu().numpy()
            exp = gold[k]
            # 5e-3: one ReLU sitting within fp32 round-off of its kink (|a| ~ 1e-6) may fire on one side and not
            # on the other; a single such flip in the BEV head moves these sums by ~2e-3 (measured)
            if np.abs(got - exp).max() > 5e-3 * max(np.abs(exp).max(), 1e-6):
                bad[name] = float(np.abs(got - exp).max())
    assert not bad, bad
    sd = model.state_dict()
    for k in gold.files:
        if k.startswith("running/"):
            assert np.abs(sd[k[len("running/"):]].cpu().numpy() - gold[k]).max() <= 1e-5, k


REL_KEYS = ("atten_attr", "atten_rel", "atten_scene", "vis_atten", "attribute_scores", "relation_scores", "scene_scores")


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_small_forward_tensors_also_meet_a_relative_bar(lib, mode):
    """1e-4 absolute (the north star's bar) is loose for the tensors whose entries are small — attention weights (max 8e-2),
    vis_atten (max 9e-3), the three matching-score vectors (median 4e-3): beside it, 1e-4 RELATIVE to each tensor's largest
    entry, against the reference's own output (tests/golden/model.npz = models/instancerefer.py:37-70 run in the build
    container)."""
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    gold = np.load(os.path.join(G, "model.npz"))
    model, dd = _build(mode)
    with torch.set_grad_enabled(mode == "train"):
        dd = get_loss(model(dd), DatasetConfig())
    worst = {}
    for k in REL_KEYS:
        got, exp = dd[k].detach().float().cpu().numpy(), gold["%s/%s" % (mode, k)]
        worst[k] = float(np.abs(got - exp).max()) / float(np.abs(exp).max())
    print("relative errors (%s):" % mode, {k: "%.1e" % v for k, v in worst.items()})
    assert all(v <= 1e-4 for v in worst.values()), worst


def test_every_parameter_gradient_elementwise_vs_oracle(lib):
    """ALL 160 parameter gradients of the whole model, element by element (not norms: a permuted or transposed gradient keeps
    its norm), against oracle/model_ref.py (pinned to the reference's models/instancerefer.py + lib/loss_helper.py output by
    tests/test_oracle_cpu.py) on the golden batch, with the encoders' ReLUs kept away from their kinks
    (helpers.kink_free_state_dict). Bar per tensor: 1e-3 of its largest entry (floor: 1e-6 of the largest entry of any
    gradient). The forward tensors of this variant meet the 1e-4 bars too."""
    from helpers import elementwise_grad_report, kink_free_state_dict
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.instancerefer import InstanceRefer
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    from oracle.model_ref import InstanceRefer as OracleModel, oracle_data_dict
    dev = torch.device("cuda")
    model = InstanceRefer(7, S.default_args())
    sd = kink_free_state_dict(S.seeded_state_dict(model, WEIGHT_SEED))
    model.load_state_dict(sd)
    oracle = OracleModel(7, S.default_args())
    oracle.load_state_dict(sd)
    for m in list(model.modules()) + list(oracle.modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.to(dev).train()
    oracle.train()
    host = S.make_batch(**dict(GOLDEN_CFG))
    dd = get_loss(model(S.to_device(dict(host), dev)), DatasetConfig())
    od = get_loss(oracle(oracle_data_dict(dict(host))), DatasetConfig())
    for k in FWD_KEYS:
        got, exp = dd[k].detach().float().cpu(), od[k].detach().float()
        assert got.shape == exp.shape, k
        if got.numel():
            assert float((got - exp).abs().max()) <= 1e-4 * max(1.0, float(exp.abs().max())), k
    dd["loss"].backward()
    od["loss"].backward()
    bad, worst, lines = elementwise_grad_report(dict(model.named_parameters()), dict(oracle.named_parameters()))
    os.makedirs('gpurun_out', exist_ok=True)
    open('gpurun_out/elementwise_golden.txt', 'w').write('\n'.join(lines) + '\n')
    print("element-wise gradients: worst error / bar = %.3f over %d tensors" % (worst, len(list(oracle.parameters()))))
    assert not bad, bad


@pytest.mark.parametrize("backend", ["cpp", "py", "aten"])
def test_model_with_the_fused_head_mlps_matches_reference(lib, monkeypatch, backend):
    """dense.mlp2 routes the seven head MLPs through irx_mlp2_fwd / _bwd — as C++ autograd nodes (csrc/torch_nodes.cpp, the
    default when built), as the Python autograd.Function, or not at all (the ATen modules): the same 1e-4 forward bar against
    the reference's output (tests/golden/model.npz) and the same gradient-norm bar on every path."""
    from instancerefer_amd import dense, heads
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    monkeypatch.setattr(heads, "FUSED", False)          # the per-operator heads (one node per head: tests/test_heads_gpu.py)
    monkeypatch.setattr(dense, "FUSED_MLP2", backend != "aten")
    monkeypatch.setattr(dense, "MLP2_BACKEND", backend)
    gold = np.load(os.path.join(G, "model.npz"))
    model, dd = _build("train")
    dd = get_loss(model(dd), DatasetConfig())
    assert type(dd["attribute_scores"].grad_fn).__name__ == "CosineRowsFnBackward"
    for k in FWD_KEYS:
        got, exp = dd[k].detach().float().cpu().numpy(), gold["train/%s" % k]
        assert got.shape == exp.shape and (not got.size or float(np.abs(got - exp).max()) <= 1e-4), k
    dd["loss"].backward()
    params = dict(model.named_parameters())
    total = float(np.sqrt(sum(float(gold[k]) ** 2 for k in gold.files if k.startswith("grad_norm/"))))
    bad = {}
    for k in gold.files:
        if k.startswith("grad_norm/"):
            name = k[len("grad_norm/"):]
            got, exp = float(params[name].grad.double().norm()), float(gold[k])
            if abs(got - exp) > 1e-3 * max(exp, 1e-3 * total):
                bad[name] = (got, exp)
    assert not bad, bad


def test_encoder_executor_equals_per_layer_path(lib):
    """The one-node encoder executor issues the same kernels in the same order as the per-layer modules: outputs,
    input gradient and every parameter gradient must be bit-identical."""
    from helpers import device_batch, surface_cloud
    from instancerefer_amd.basic_blocks import SparseConvEncoder
    from instancerefer_amd.sparse import encoder_fn
    rng = np.random.default_rng(5)
    clouds = [surface_cloud(rng, 3000, rng.uniform(0, 3, 3), rng.uniform(0.8, 2.0, 3)) for _ in range(3)]
    torch.manual_seed(1)
    enc = SparseConvEncoder(7).cuda().train()
    res = {}
    # "fused_up": the executor with the parent-tiled stride-2 data-gradient (k_updgrad) forced on at this small size — in
    # production it takes over from 40 k parent rows; same weight image, same reduction order, one pair per output row
    from instancerefer_amd import _lib
    updgrad_min = _lib.get_knob("updgrad_min")
    for mode in ("fused", "layers", "fused_up"):
        _lib.set_knob("updgrad_min", 0 if mode == "fused_up" else 1000000000)
        enc.zero_grad()
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.reset_running_stats()
        st = device_batch(clouds, 0.05)
        x = st.F.clone().requires_grad_(True)
        st = st.with_feats(x)
        if mode == "layers":
            saved, encoder_fn.can_fuse = encoder_fn.can_fuse, (lambda e: False)
        out = enc(st)
        if mode == "layers":
            encoder_fn.can_fuse = saved
        g = torch.linspace(-1, 1, out.F.numel(), device="cuda").view_as(out.F)
        out.F.backward(g)
        res[mode] = (out.F.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in enc.named_parameters()},
                     {n: b.clone() for n, b in enc.named_buffers()})
    _lib.set_knob("updgrad_min", updgrad_min)
    for other in ("layers", "fused_up"):
        assert torch.equal(res["fused"][0], res[other][0])
        assert torch.equal(res["fused"][1], res[other][1]), other
        for n in res["fused"][2]:
            assert torch.equal(res["fused"][2][n], res[other][2][n]), (other, n)
        for n in res["fused"][3]:
            assert torch.equal(res["fused"][3][n], res[other][3][n]), (other, n)


def _oracle_and_product(cfg, seed, c0=7):
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.instancerefer import InstanceRefer
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    from oracle.model_ref import InstanceRefer as OracleModel, oracle_data_dict
    dev = torch.device("cuda")
    model = InstanceRefer(c0, S.default_args())
    sd = S.seeded_state_dict(model, seed)
    model.load_state_dict(sd)
    oracle = OracleModel(c0, S.default_args())
    oracle.load_state_dict(sd)
    for m in list(model.modules()) + list(oracle.modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.to(dev).train()
    oracle.train()
    dd = get_loss(model(S.to_device(S.make_batch(**dict(cfg)), dev)), DatasetConfig())
    od = get_loss(oracle(oracle_data_dict(S.make_batch(**dict(cfg)))), DatasetConfig())
    return model, oracle, dd, od


@pytest.mark.parametrize("variant", ["centred", "corner"])
def test_full_model_vs_cpu_oracle_negative_coords_and_ragged(lib, variant):
    """Seeded scenes the golden fixture does not cover, HIP path vs the CPU oracle (oracle/model_ref.py, itself pinned
    to the reference fixture): room centred on the origin (negative voxel coordinates, part of the scene outside the
    BEV crop window), ragged candidate counts incl. a scene with 1 and one with 0 same-class candidates, ragged
    utterance lengths. Forward 1e-4 absolute; gradient norms 2e-3."""
    cfg = dict(batch_size=4, seed=900, num_points=5000, num_instances=5, num_candidates=[3, 1, 0, 2],
               tokens=[17, 30, 5, 1], points_per_instance=200, variant=variant)
    from instancerefer_amd import synthetic as S
    # candidate count 0: make_scene gives instance classes target+1+j for j >= c, so c = 0 leaves no match
    model, oracle, dd, od = _oracle_and_product(cfg, 77)
    assert list(dd["num_filtered_objs"]) == list(od["num_filtered_objs"]) == [3, 1, 0, 2]
    for k in ("lang_scores", "obj_feats", "attribute_scores", "relation_scores", "scene_scores", "seg_scores",
              "vis_atten", "loss", "ref_loss", "lang_loss", "seg_loss"):
        err = float((dd[k].detach().cpu() - od[k].detach()).abs().max())
        assert err <= 1e-4, (k, err)
    dd["loss"].backward()
    od["loss"].backward()
    gp = dict(model.named_parameters())
    total = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in oracle.parameters() if p.grad is not None)))
    for n, p in oracle.named_parameters():
        if p.grad is None:
            continue
        exp = float(p.grad.double().norm())
        got = float(gp[n].grad.double().norm())
        assert abs(got - exp) <= 2e-3 * max(exp, 1e-3 * total), (n, got, exp)


def test_full_model_multiview_c135_vs_cpu_oracle(lib):
    """BASELINE configs[4]'s input: use_multiview adds 128 ENet channels (reference scripts/train.py:74-75,
    lib/dataset.py:112-118) -> C0 = 135 for both sparse stems, the relation node features (135 + 18) and the edge MLPs.
    HIP path vs the CPU oracle at that width: forward 1e-4 absolute, gradient norms 2e-3."""
    cfg = dict(batch_size=2, seed=950, num_points=4000, num_instances=5, num_candidates=[3, 2], tokens=[20, 11],
               points_per_instance=200, multiview=128)
    model, oracle, dd, od = _oracle_and_product(cfg, 78, c0=135)
    assert model.scene.net.stem[0].net[0].kernel.shape == (27, 135, 32)
    assert list(dd["num_filtered_objs"]) == list(od["num_filtered_objs"]) == [3, 2]
    for k in ("lang_scores", "obj_feats", "attribute_scores", "relation_scores", "scene_scores", "seg_scores",
              "vis_atten", "loss", "ref_loss", "lang_loss", "seg_loss"):
        err = float((dd[k].detach().cpu() - od[k].detach()).abs().max())
        assert err <= 1e-4, (k, err)
    dd["loss"].backward()
    od["loss"].backward()
    gp = dict(model.named_parameters())
    total = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in oracle.parameters() if p.grad is not None)))
    for n, p in oracle.named_parameters():
        if p.grad is None:
            continue
        exp = float(p.grad.double().norm())
        got = float(gp[n].grad.double().norm())
        assert abs(got - exp) <= 2e-3 * max(exp, 1e-3 * total), (n, got, exp)
    # the two 135 -> 32 stems element by element (executor: 128 leading channels through the pair lists in k_wgrad_pairs,
    # the 7-channel tail through the stem kernel, merged): a misplaced channel or offset would keep the norm
    for n in ("scene.net.stem.0.net.0.kernel", "attribute.net.stem.0.net.0.kernel"):
        exp, got = dict(oracle.named_parameters())[n].grad, gp[n].grad.cpu()
        assert exp.shape == got.shape == (27, 135, 32)
        assert float((got - exp).abs().max()) <= 1e-2 * float(exp.abs().max()), n
        assert float((got[:, :128] - exp[:, :128]).norm()) <= 5e-3 * float(exp[:, :128].norm()), n
        assert float((got[:, 128:] - exp[:, 128:]).norm()) <= 5e-3 * float(exp[:, 128:].norm()), n


def test_use_gt_lang_false_takes_the_argmax_branch(lib):
    """`use_gt_lang: False` (reference models/attribute_module.py:93-95, relation_module.py:86-88): the candidates are the
    instances of the class the language classifier PREDICTS, so nothing can be prepared ahead of the language module.
    The instances are relabelled so that the predicted class of each utterance has 3 / 2 / 1 members; HIP path vs the
    CPU oracle: same selection, forward 1e-4, gradient norms 2e-3."""
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.instancerefer import InstanceRefer
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    from oracle.model_ref import InstanceRefer as OracleModel, oracle_data_dict
    dev = torch.device("cuda")
    args = S.default_args(use_gt_lang=False)
    model = InstanceRefer(7, args)
    sd = S.seeded_state_dict(model, 91)
    model.load_state_dict(sd)
    oracle = OracleModel(7, S.default_args(use_gt_lang=False))
    oracle.load_state_dict(sd)
    for m in list(model.modules()) + list(oracle.modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.to(dev).train()
    oracle.train()
    cfg = dict(batch_size=3, seed=1200, num_points=4000, num_instances=5, num_candidates=[3, 2, 1], tokens=[12, 30, 7],
               points_per_instance=200)
    oracle.eval()
    with torch.no_grad():
        pred = oracle.lang(oracle_data_dict(S.make_batch(**dict(cfg))))["lang_scores"].argmax(1).tolist()
    oracle.train()
    assert pred != [4, 4, 4], "the classifier must disagree with the ground-truth class somewhere"

    def batch():
        host = S.make_batch(**dict(cfg))
        for i, c in enumerate(cfg["num_candidates"]):
            host["instance_class"][i] = [pred[i] if j < c else (pred[i] + 1 + j) % 18 for j in range(cfg["num_instances"])]
        return host
    dd = get_loss(model(S.to_device(batch(), dev)), DatasetConfig())
    od = get_loss(oracle(oracle_data_dict(batch())), DatasetConfig())
    assert list(dd["num_filtered_objs"]) == list(od["num_filtered_objs"]) == [3, 2, 1]
    for k in ("lang_scores", "obj_feats", "attribute_scores", "relation_scores", "scene_scores", "seg_scores", "loss",
              "ref_loss"):
        err = float((dd[k].detach().cpu() - od[k].detach()).abs().max())
        assert err <= 1e-4, (k, err)
    dd["loss"].backward()
    od["loss"].backward()
    gp = dict(model.named_parameters())
    total = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in oracle.parameters() if p.grad is not None)))
    for n, p in oracle.named_parameters():
        if p.grad is None:
            continue
        exp, got = float(p.grad.double().norm()), float(gp[n].grad.double().norm())
        assert abs(got - exp) <= 2e-3 * max(exp, 1e-3 * total), (n, got, exp)


def test_all_scenes_below_two_candidates_is_graceful(lib):
    """A shard where no scene has >= 2 same-class candidates (the reference crashes in torch.cat([])): the drop-in
    returns empty score tensors, a finite loss from the language / scene-area terms, and gradients for every rank
    to enter the all-reduce with."""
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.instancerefer import InstanceRefer
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    dev = torch.device("cuda")
    model = InstanceRefer(7, S.default_args()).to(dev).train()
    dd = S.to_device(S.make_batch(2, seed=5, num_points=3000, num_instances=4, num_candidates=[1, 0],
                                  points_per_instance=128), dev)
    dd = get_loss(model(dd), DatasetConfig())
    assert dd["attribute_scores"].numel() == 0 and dd["relation_scores"].numel() == 0 and dd["scene_scores"].numel() == 0
    assert torch.isfinite(dd["loss"]).all() and float(dd["ref_loss"]) == 0.0
    dd["loss"].backward()
    assert model.lang.lang_cls[0].weight.grad is not None and model.scene.cls[3].weight.grad is not None


def test_solver_trains_and_writes_reference_style_checkpoints(lib, tmp_path):
    """A few optimisation steps on synthetic batches: the loss goes down on a repeated batch, model_last.pth / model.pth /
    checkpoint.tar are written with the reference's key layout and reload into a fresh model."""
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.instancerefer import InstanceRefer
    from instancerefer_amd.loss_helper import DatasetConfig
    from instancerefer_amd.solver import Solver, SyntheticLoader
    torch.manual_seed(0)
    model = InstanceRefer(7, S.default_args())
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    kw = dict(num_points=4000, num_instances=5, num_candidates=3, points_per_instance=128)

    class Repeat(SyntheticLoader):           # the same batch every iteration: the loss must fall
        def __iter__(self):
            for _ in range(self.batches):
                yield next(iter(SyntheticLoader(1, self.batch_size, seed=self.seed, **self.kw)))

    solver = Solver(model, DatasetConfig(), {"train": Repeat(6, 3, seed=40, **kw), "val": SyntheticLoader(1, 3, seed=99, **kw)},
                    lr=1e-3, out_dir=str(tmp_path), verbose=1)
    solver(1)
    losses = [r["loss"] for r in solver.log["train"]]
    assert len(losses) == 6 and all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert solver.log["val"] and 0.0 <= solver.log["val"][0]["iou_rate_0.25"] <= 1.0
    for f in ("model_last.pth", "model.pth", "checkpoint.tar", "log.txt", "best.txt"):
        assert os.path.exists(os.path.join(str(tmp_path), f)), f
    sd = torch.load(os.path.join(str(tmp_path), "model_last.pth"), map_location="cpu")
    assert "attribute.net.stem.0.net.0.kernel" in sd and "scene.to_bev.1.kernel" in sd and "lang.gru.weight_ih_l0" in sd
    fresh = InstanceRefer(7, S.default_args())
    fresh.load_state_dict(sd)
    ck = torch.load(os.path.join(str(tmp_path), "checkpoint.tar"), map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict"}
    # the reference resumes with torch.optim.Adam(model.parameters()).load_state_dict(...) (scripts/train.py:114-119)
    ta = torch.optim.Adam(fresh.parameters(), lr=1.0)
    ta.load_state_dict(ck["optimizer_state_dict"])
    assert ta.param_groups[0]["lr"] == 1e-3 and len(ta.state) == len(list(fresh.parameters()))
    assert all(float(s["step"]) == 6.0 for s in ta.state.values())
    assert os.path.exists(os.path.join(str(tmp_path), "scalars.jsonl"))
    # and this Solver resumes from it: same moments, same step counts, next epoch
    again = InstanceRefer(7, S.default_args())
    s2 = Solver(again, DatasetConfig(), {"train": Repeat(2, 3, seed=40, **kw)}, lr=1e-3, out_dir=str(tmp_path / "resumed"),
                verbose=1, use_checkpoint=os.path.join(str(tmp_path), "checkpoint.tar"))
    assert s2.start_epoch == 1 and s2.optimizer.steps == solver.optimizer.steps
    assert torch.equal(s2.optimizer.exp_avg, solver.optimizer.exp_avg) and torch.equal(s2.optimizer.flat_p, solver.optimizer.flat_p)
    for m in again.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    s2(2)
    assert len(s2.log["train"]) == 2 and s2.log["train"][0]["loss"] < losses[0]
    # resuming past a decay milestone continues the schedule (the saved rate is already decayed)
    s3 = Solver(InstanceRefer(7, S.default_args()), DatasetConfig(), {"train": Repeat(1, 3, seed=40, **kw)}, lr=1e-3,
                lr_decay_step=(1, 3), out_dir=str(tmp_path / "r3"), verbose=1,
                use_checkpoint=os.path.join(str(tmp_path / "resumed"), "checkpoint.tar"))
    assert s3.start_epoch == 2 and s3.base_lr == 1e-3
    s3.train_epoch(2)
    assert abs(s3.optimizer.lr - 1e-4) <= 1e-12
    s3.finish(3)
    s4 = Solver(InstanceRefer(7, S.default_args()), DatasetConfig(), {"train": Repeat(1, 3, seed=40, **kw)}, lr=1e-3,
                lr_decay_step=(1, 3), out_dir=str(tmp_path / "r4"), verbose=1,
                use_checkpoint=os.path.join(str(tmp_path / "r3"), "checkpoint.tar"))
    s4.train_epoch(3)
    assert s4.base_lr == 1e-3 and abs(s4.optimizer.lr - 1e-5) <= 1e-13


def test_pipelined_training_steps_equal_inline_steps(lib):
    """bench.py's input pipeline (prepare() of batch N+1 on its own HIP stream behind step N, inline or on a helper
    thread, record_stream hand-over, scene encoder on a second stream) must not change a single bit of the training trajectory compared
    with fully inline steps: same seeds -> identical loss sequence and identical parameters after 4 steps."""
    import argparse
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.loss_helper import DatasetConfig, prepare_labels
    from instancerefer_amd.optim import FlatAdam
    dev = torch.device("cuda")
    bench.step_fn.cfg = DatasetConfig()
    out = {}
    for mode in ("inline", "pipelined", "pipelined-thread", "pipelined-at-backward"):
        torch.manual_seed(99)
        model = bench.build_model(argparse.Namespace(), "full", dev)
        resident = S.to_device(S.make_batch(4, seed=11, num_points=6000, num_instances=6, num_candidates=3,
                                            points_per_instance=256), dev)
        lidar = resident.pop("lidar")
        resident["lidar_F"], resident["lidar_C"], resident["B"] = lidar.F, lidar.C, 4
        opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
        state = {"pipeline": mode != "inline", "threaded": mode == "pipelined-thread",
                 "at_backward": mode == "pipelined-at-backward"}
        if mode != "inline":
            state["labels"] = lambda dd: prepare_labels(dd, bench.step_fn.cfg, dev) if "_attr_prepared" in dd else None
        losses = [float(bench.step_fn(model, resident, "full", None, opt, state)) for _ in range(4)]
        torch.cuda.synchronize()
        th = state.pop("thread", None)
        if th is not None:
            th.join()
        out[mode] = (losses, torch.cat([p.detach().flatten() for p in model.parameters()]).clone())
    for mode in ("pipelined", "pipelined-thread", "pipelined-at-backward"):
        assert out["inline"][0] == out[mode][0], (mode, out["inline"][0], out[mode][0])
        assert torch.equal(out["inline"][1], out[mode][1]), mode


def test_gradient_sink_equals_autograd_accumulation(lib):
    """The encoders' executor writes its parameter gradients straight into FlatAdam's flat gradient buffer (sink
    protocol, incl. the cross-stream event for the scene encoder); with the sink disabled the same gradients travel
    through AccumulateGrad + the gather copy. Same seeds -> bit-identical losses and parameters after 3 steps."""
    import argparse
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.loss_helper import DatasetConfig
    from instancerefer_amd.optim import FlatAdam
    dev = torch.device("cuda")
    bench.step_fn.cfg = DatasetConfig()
    out = {}
    for mode in ("sink", "autograd"):
        torch.manual_seed(7)
        model = bench.build_model(argparse.Namespace(), "full", dev)
        resident = S.to_device(S.make_batch(4, seed=21, num_points=6000, num_instances=6, num_candidates=3,
                                            points_per_instance=256), dev)
        lidar = resident.pop("lidar")
        resident["lidar_F"], resident["lidar_C"], resident["B"] = lidar.F, lidar.C, 4
        opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
        if mode == "autograd":
            opt.sink_slots = lambda key, params: None
            opt.native_sink = lambda key, params: None
        losses = [float(bench.step_fn(model, resident, "full", None, opt, None).detach()) for _ in range(3)]
        native, n_native = opt.native_delivered()         # the heads' C++ nodes (csrc/torch_nodes.cpp): 7 MLPs + the 2 GRU layers
        delivered = len(opt._direct) - n_native
        torch.cuda.synchronize()
        out[mode] = (losses, opt.flat_p.clone(), delivered, native)
    from instancerefer_amd import _nodes, dense
    assert out["sink"][2] == 78 and out["autograd"][2] == 0          # 2 encoders x 13 layers x (kernel, gamma, beta)
    from instancerefer_amd import heads
    # C++ producers: round 5 = 7 head MLPs + 2 GRU layers; with one node per head (heads.py) = scene head + attribute/scene-score head
    # + the relation head's 2 MLPs + 2 GRU layers
    # ... then the relation head as one node (its 2 MLPs inside) and the language attention pooling: 4 head nodes + 2 GRU layers
    # ... and the two heads' language-side MLPs as nodes of their own (heads.PreLang): 4 head nodes + 2 MLPs + 2 GRU layers
    expect = 0 if (_nodes.load() is None or dense.FUSED_MLP2 is False) else (8 if heads._mod() is not None else 9)
    assert out["sink"][3] == expect and out["autograd"][3] == 0, out["sink"][3]
    assert out["sink"][0] == out["autograd"][0], (out["sink"][0], out["autograd"][0])
    assert torch.equal(out["sink"][1], out["autograd"][1])


@pytest.mark.parametrize("mode", ["bf16_operands", "bf16"])
def test_bf16_modes_track_the_fp32_reference(lib, mode):
    """BASELINE configs[2]-[4] dtype on the golden batch — "bf16_operands": bf16 operands / fp32 accumulation in the MFMA
    sparse convs, every tensor fp32; "bf16": the same plus bf16 STORAGE of the activations / gradients inside the two
    encoders (statistics, parameters, heads fp32): same discrete decisions (candidates, labels), matching
    scores and features within bf16 round-off (2e-2 of max(1, |reference|max); measured: scores 2-4e-4, pooled
    features 4e-3), loss within 2 %, finite
    gradients whose norm is within 5 % of the fp32 reference's. The fp32 mode stays the 1e-4 parity gate; the bf16 arithmetic
    itself is pinned against the emulating oracle in tests/test_bf16_gpu.py (layer by layer <= 3e-4)."""
    import instancerefer_amd as irx
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    gold = np.load(os.path.join(G, "model.npz"))
    model, dd = _build("train")
    irx.set_compute_dtype(mode)
    try:
        assert irx.get_compute_dtype() == mode
        dd = get_loss(model(dd), DatasetConfig())
        dd["loss"].backward()
        torch.cuda.synchronize()
    finally:
        irx.set_compute_dtype("fp32")
    assert list(dd["num_filtered_objs"]) == gold["train/num_filtered_objs"].tolist()
    lab = np.concatenate([c.cpu().numpy() if len(c) else np.zeros(0) for c in dd["cluster_label"]])
    assert np.array_equal(lab, gold["train/cluster_label"])
    worst = {}
    for k in ("attribute_scores", "relation_scores", "scene_scores", "seg_scores", "obj_feats", "vis_atten"):
        exp = gold["train/" + k]
        worst[k] = float(np.abs(dd[k].detach().float().cpu().numpy() - exp).max()) / max(1.0, float(np.abs(exp).max()))
    assert all(v <= 2e-2 for v in worst.values()), worst
    assert worst["attribute_scores"] > 1e-6, "bf16 mode did not change the arithmetic"
    assert abs(float(dd["loss"]) - float(gold["train/loss"])) <= 2e-2 * abs(float(gold["train/loss"]))
    total = float(np.sqrt(sum(float(gold[k]) ** 2 for k in gold.files if k.startswith("grad_norm/"))))
    got = float(np.sqrt(sum(float(p.grad.double().norm()) ** 2 for p in model.parameters() if p.grad is not None)))
    assert np.isfinite(got) and abs(got - total) <= 5e-2 * total, (got, total)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_asynchronous_encoder_issue_changes_nothing(lib, dtype):
    """Encoder passes issued by library threads (irx_encoder_submit / irx_encoder_wait; in bf16 mode the candidate
    encoder is also issued ahead of the language module) vs the same passes issued inline by the Python thread: same
    kernels on the same streams -> bit-identical losses and parameters after 3 training steps."""
    import argparse
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import instancerefer_amd as irx
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.loss_helper import DatasetConfig
    from instancerefer_amd.optim import FlatAdam
    from instancerefer_amd.sparse import encoder_fn
    dev = torch.device("cuda")
    bench.step_fn.cfg = DatasetConfig()
    out = {}
    saved = encoder_fn.ASYNC
    irx.set_compute_dtype(dtype)
    try:
        for mode in (True, False):
            encoder_fn.ASYNC = mode
            torch.manual_seed(11)
            model = bench.build_model(argparse.Namespace(), "full", dev)
            resident = S.to_device(S.make_batch(4, seed=33, num_points=6000, num_instances=6, num_candidates=3,
                                                points_per_instance=256), dev)
            lidar = resident.pop("lidar")
            resident["lidar_F"], resident["lidar_C"], resident["B"] = lidar.F, lidar.C, 4
            opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
            losses = [float(bench.step_fn(model, resident, "full", None, opt, None).detach()) for _ in range(3)]
            torch.cuda.synchronize()
            out[mode] = (losses, opt.flat_p.clone())
    finally:
        encoder_fn.ASYNC = saved
        irx.set_compute_dtype("fp32")
    assert out[True][0] == out[False][0], (out[True][0], out[False][0])
    assert torch.equal(out[True][1], out[False][1])


def test_training_trajectory_tracks_the_oracle(lib, tmp_path):
    """The only accuracy proxy available without ScanRefer (north star: Acc@0.25/0.5 within +-0.2 of the reference): 50
    fp32 optimisation steps (reference lib/solver.py:200-205: backward(); step() with torch.optim.Adam(lr=1e-3,
    weight_decay=1e-5), scripts/train.py:121) on 50 different synthetic batches, product (FlatAdam: one fused launch over
    the flat buffer) vs oracle/model_ref.py + a real torch.optim.Adam.

    Free-running trajectories cannot be compared beyond a few steps: Adam's update lr * m / (sqrt(v) + 1e-8) is sign-like,
    so a parameter whose true gradient is a near-cancellation (|g| ~ fp32 noise of its terms) moves by +-lr in a direction
    the noise decides — measured here: loss deviation 1e-7, 5e-5, 8e-4, 2e-2 at steps 0..3, with ANY second fp32
    implementation. So the trajectory is checked state by state ("teacher forced"): at each of the 50 steps the oracle is
    given the product's current parameters, buffers and Adam moments, both run forward / backward / step on the same batch:
      * loss, every step: <= 1e-4 relative;  gradient (all parameters, flat): cosine >= 1 - 1e-6 (two ReLU-kink steps may reach
        1 - 2e-5), norm within 1e-4;
      * parameters after the step: || p_product - p_oracle || <= 3e-2 || update || (measured 1e-2) (the noise-decided elements are few);
      * BatchNorm running statistics after the step: <= 1e-5;
    and the free-running first two steps agree within 1e-3, the third within 1e-2; the model is learning (mean loss of the last 10 steps below
    the first 10)."""
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.instancerefer import InstanceRefer
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    from instancerefer_amd.optim import FlatAdam
    from oracle.model_ref import InstanceRefer as OracleModel, oracle_data_dict
    steps, bs = 50, 3
    kw = dict(num_points=4000, num_instances=5, num_candidates=3, points_per_instance=128)
    dev = torch.device("cuda")
    nthreads = torch.get_num_threads()
    torch.set_num_threads(min(8, nthreads))          # the CPU oracle's small ops crawl with one thread per core of a big host

    def fresh(cls):
        m = cls(7, S.default_args())
        m.load_state_dict(S.seeded_state_dict(m, 61))
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        return m.train()
    try:
        model, oracle, free = fresh(InstanceRefer).to(dev), fresh(OracleModel), fresh(OracleModel)
        opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
        o_opt = torch.optim.Adam(oracle.parameters(), lr=1e-3, weight_decay=1e-5)
        f_opt = torch.optim.Adam(free.parameters(), lr=1e-3, weight_decay=1e-5)
        names = [n for n, _ in model.named_parameters()]
        worst = dict(loss=0.0, cos=0.0, gnorm=0.0, step=0.0, buf=0.0)
        losses, free_dev, cos_all = [], [], []
        for b in range(steps):
            host = lambda: S.make_batch(bs, seed=500 + b * bs, **dict(kw))
            # teacher forcing: the oracle starts this step from the product's state (parameters, buffers, Adam moments)
            oracle.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
            if b > 0:
                o_opt.load_state_dict(opt.state_dict())
            before = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}
            opt.zero_grad()
            dd = get_loss(model(S.to_device(host(), dev)), DatasetConfig())
            dd["loss"].backward()
            opt.gather_grads()
            flat_g = opt.flat_g[:opt.n].clone()
            opt.all_reduce()
            opt.step()
            o_opt.zero_grad()
            od = get_loss(oracle(oracle_data_dict(host())), DatasetConfig())
            od["loss"].backward()
            lp, lo = float(dd["loss"].detach()), float(od["loss"].detach())
            losses.append(lp)
            worst["loss"] = max(worst["loss"], abs(lp - lo) / abs(lo))
            gp = torch.cat([flat_g[off:off + p.numel()].cpu().double() for p, off in zip(opt.params, opt.offsets)])
            go = torch.cat([(q.grad if q.grad is not None else torch.zeros_like(q)).reshape(-1).double()
                            for q in (dict(oracle.named_parameters())[n] for n in names)])
            cdev = 1.0 - float(gp @ go / (gp.norm() * go.norm()))
            if cdev > 5e-7:                             # diagnostics: which parameters carry a step's deviation
                offs, per = 0, []
                for n_, p_ in zip(names, opt.params):
                    k_ = p_.numel()
                    per.append((float((gp[offs:offs + k_] - go[offs:offs + k_]).norm()), n_))
                    offs += k_
                print("step %d: 1 - cos = %.2e; largest gradient differences:" % (b, cdev), sorted(per, reverse=True)[:4],
                      "|g| = %.3e" % float(go.norm()))
            worst["cos"] = max(worst["cos"], cdev)
            cos_all.append(cdev)
            worst["gnorm"] = max(worst["gnorm"], abs(float(gp.norm() / go.norm()) - 1.0))
            o_opt.step()
            torch.cuda.synchronize()
            num = den = 0.0
            op = dict(oracle.named_parameters())
            for n, p in model.named_parameters():
                pc = p.detach().cpu().double()
                num += float(((pc - op[n].detach().double()) ** 2).sum())
                den += float(((pc - before[n].double()) ** 2).sum())
            worst["step"] = max(worst["step"], (num / den) ** 0.5)
            ob = dict(oracle.named_buffers())
            for n, buf in model.named_buffers():
                if buf.dtype.is_floating_point:
                    worst["buf"] = max(worst["buf"], float((buf.detach().cpu() - ob[n]).abs().max()) / max(1.0, float(ob[n].abs().max())))
            if b < 3:                                   # free-running oracle: never re-synchronised
                f_opt.zero_grad()
                fd = get_loss(free(oracle_data_dict(host())), DatasetConfig())
                fd["loss"].backward()
                f_opt.step()
                free_dev.append(abs(lp - float(fd["loss"].detach())) / abs(lp))
    finally:
        torch.set_num_threads(nthreads)
    print("trajectory (teacher forced, %d steps):" % steps, {k: "%.1e" % v for k, v in worst.items()},
          "free-running first steps:", ["%.1e" % v for v in free_dev], "loss %.3f -> %.3f" % (np.mean(losses[:10]), np.mean(losses[-10:])))
    # gradient direction: 1 - cos <= 1e-6 at every step but (at most) two, which may reach 2e-5: once in a while a unit of a deep
    # encoder level sits within fp32 summation noise of its ReLU kink and the two implementations take different sides (measured:
    # step 19 of this run, 2.4e-6, all of it in attribute.net.stage3's kernels; the other 49 steps stay below 5e-7)
    cs = sorted(cos_all)
    assert worst["loss"] <= 1e-4 and cs[-3] <= 1e-6 and cs[-1] <= 2e-5 and worst["gnorm"] <= 1e-4, (worst, cs[-3:])
    assert worst["step"] <= 3e-2 and worst["buf"] <= 1e-5, worst
    # (free-running steps amplify fp32 summation-order differences through Adam's sign-like update, docstring: 1e-7, 5e-5,
    #  8e-4 at steps 0..2 with the heads through ATen, 2e-7, 5e-5, 2e-3 with the fused head MLPs' own summation order)
    assert max(free_dev[:2]) <= 1e-3 and max(free_dev) <= 1e-2, free_dev
    assert np.mean(losses[-10:]) < np.mean(losses[:10]), losses


def _three_steps(dev, dtype="fp32", no_dropout=False):
    import argparse
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.loss_helper import DatasetConfig
    from instancerefer_amd.optim import FlatAdam
    bench.step_fn.cfg = DatasetConfig()
    torch.manual_seed(11)
    model = bench.build_model(argparse.Namespace(), "full", dev)
    if no_dropout:                       # (the order in which the modules draw their dropout seeds follows the issue order)
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    resident = S.to_device(S.make_batch(4, seed=33, num_points=6000, num_instances=6, num_candidates=3,
                                        points_per_instance=256), dev)
    lidar = resident.pop("lidar")
    resident["lidar_F"], resident["lidar_C"], resident["B"] = lidar.F, lidar.C, 4
    opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
    losses = [float(bench.step_fn(model, resident, "full", None, opt, None).detach()) for _ in range(3)]
    torch.cuda.synchronize()
    return model, resident, opt, losses


@pytest.mark.parametrize("knob", ["lang_thread", "scene_defer", "streams", "streams_inline_lang", "bwd_gate"])
def test_helper_thread_and_deferred_scene_node_change_nothing(lib, monkeypatch, knob):
    """(1) The language module issued by the helper thread vs inline (IRX_LANG_THREAD=0): same kernels on the same stream.
    (2) The scene encoder's autograd node created at the head of SceneModule.forward (its pass issued earlier:
    encoder_fn.Launched) vs created at issue time: only the ORDER in which the backward reaches the encoders changes.
    (3) The three-stream forward (InstanceRefer._forward_streams: scene encoder + scene head | language + relation | candidate
    encoder + attribute head + scores) vs the one-stream layout, with the language module on the helper thread or inline.
    (4) The backward gate of the three-stream layout on vs off.
    Bit-identical losses and parameters after 3 training steps either way."""
    from instancerefer_amd import instancerefer as IR, scene_module as SM
    dev = torch.device("cuda")
    out = {}
    for on in (True, False):
        if knob == "lang_thread":
            monkeypatch.setattr(IR, "_LANG_THREAD", on)
        elif knob == "scene_defer":
            monkeypatch.setattr(SM, "_DEFER", on)
        elif knob == "bwd_gate":
            # (4) the backward gate between the two encoder passes (irx_encoder_gate_next: the scene encoder's backward waits,
            # on its stream, for the head of the candidate encoder's): an ordering of launches, nothing else
            monkeypatch.setattr(IR, "_STREAMS", True)
            monkeypatch.setattr(IR, "_STREAMS_ENV", "1")
            monkeypatch.setattr(IR, "_BWD_GATE", on)
        else:
            monkeypatch.setattr(IR, "_STREAMS", on)
            monkeypatch.setattr(IR, "_STREAMS_ENV", "1" if on else "0")     # (the policy alone keeps fp32 on one stream pair)
            monkeypatch.setattr(IR, "_LANG_THREAD", knob == "streams")
        _, _, opt, losses = _three_steps(dev, no_dropout=knob.startswith("streams") or knob == "bwd_gate")
        out[on] = (losses, opt.flat_p.clone())
    assert out[True][0] == out[False][0], (out[True][0], out[False][0])
    assert torch.equal(out[True][1], out[False][1])


def test_helper_thread_lifecycle(lib, monkeypatch):
    """(a) an exception inside the language module (on the helper thread) and one inside the encoders' issue (main
    thread, helper job in flight) both surface in forward() and the NEXT forward is unaffected — no stale result is consumed;
    (b) copy.deepcopy(model) and pickling work after a training forward; (c) the worker thread holds no reference to the model:
    dropping the model stops it."""
    import copy
    import gc
    import io
    import weakref
    from instancerefer_amd import instancerefer as IR
    dev = torch.device("cuda")
    monkeypatch.setattr(IR, "_LANG_THREAD", True)
    model, resident, opt, losses = _three_steps(dev)
    import bench
    worker = model.__dict__["_lang_worker"]
    assert worker.thread.is_alive()
    ref = float(bench.step_fn(model, resident, "full", None, opt, None).detach())
    state = opt.flat_p.clone()

    class Boom(RuntimeError):
        pass
    # (a1) inside the language module
    orig_lang = type(model.lang).forward

    def bad_lang(self, dd):
        raise Boom("lang")
    monkeypatch.setattr(type(model.lang), "forward", bad_lang)
    with pytest.raises(Boom):
        model(bench.fresh_batch(resident))
    monkeypatch.setattr(type(model.lang), "forward", orig_lang)
    # (a2) inside the encoders' issue, while the helper job is in flight
    orig_encode = type(model.scene).encode

    def bad_encode(self, dd):
        raise Boom("encoders")
    monkeypatch.setattr(type(model.scene), "encode", bad_encode)
    with pytest.raises(Boom):
        model(bench.fresh_batch(resident))
    monkeypatch.setattr(type(model.scene), "encode", orig_encode)
    # (a3) an interrupted join: the result of that job stays in the queue; the next forward must not consume it
    stale = worker.post((model.lang, None), dict(bench.fresh_batch(resident), lang_feat=resident["lang_feat"] * 0))
    torch.cuda.synchronize()
    assert torch.equal(opt.flat_p, state)
    nxt = float(bench.step_fn(model, resident, "full", None, opt, None).detach())
    assert np.isfinite(nxt) and abs(nxt - ref) < 0.5 * abs(ref) + 1.0 and stale < worker.seq
    # (b) copies
    twin = copy.deepcopy(model)
    assert "_lang_worker" not in twin.__dict__ and "_enc_streams" not in twin.__dict__
    buf = io.BytesIO()
    torch.save(model, buf)
    assert buf.tell() > 1 << 20
    # (c) the thread ends with the model
    th = worker.thread
    wr = weakref.ref(model)
    del model, opt, twin, worker
    gc.collect()
    assert wr() is None
    th.join(timeout=5.0)
    assert not th.is_alive()


def test_a_training_step_leaves_nothing_for_the_cyclic_collector(lib):
    """Reference counting alone frees a step: the coordinate pyramid, its tables, the encoder plans cached on it and the autograd graph.
    (Round 6: level -> cached plan -> level was a cycle per pyramid — ~200 objects and ~120 device tensors per step that stayed until a
    generation-2 collection, a 6-10 ms pause every ~30 steps of a training loop. encoder_fn.Plan holds its finest level weakly.)"""
    import gc
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    from instancerefer_amd import synthetic as S
    model, _ = _build("train")
    cfg = DatasetConfig()

    def step():
        dd = S.to_device(S.make_batch(**dict(GOLDEN_CFG)), torch.device("cuda"))
        get_loss(model(dd), cfg)["loss"].backward()
        model.zero_grad(set_to_none=True)

    step()
    torch.cuda.synchronize()
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        gc.set_debug(gc.DEBUG_SAVEALL)
        gc.collect()
        gc.set_debug(0)
        kinds = sorted({type(o).__name__ for o in gc.garbage})
        tensors = sum(isinstance(o, torch.Tensor) for o in gc.garbage)
        levels = [o for o in gc.garbage if type(o).__name__ in ("Level", "DownMap", "Plan", "_Layer")]
        del gc.garbage[:]
    finally:
        if was:
            gc.enable()
    # (the regression this guards left ~120 tensors and ~45 pyramid objects PER STEP; a handful of unrelated small cycles would not matter)
    assert not levels and tensors < 30, "three steps left %d tensors / %d pyramid objects to the cyclic collector (%s)" % (tensors, len(levels), kinds)
