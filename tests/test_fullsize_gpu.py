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


@pytest.mark.parametrize("which,weights", [("scene", "kinkfree"), ("candidates", "kinkfree"), ("scene", "seeded")])
def test_encoder_fwd_bwd_equals_c_port_at_full_size(lib, which, weights):
    """BASELINE.json's FULL sizes against the C/OpenMP port (oracle/csrc/spconv_cpu.c — itself pinned to the Python oracle
    in tests/test_oracle_cpu.py; restates torchsparse's gather-GEMM-scatter conv behind reference
    models/basic_blocks.py:59-95): the scene encoder on 16 scenes x 50 k points (~490 k voxels at 5 cm) and the candidate
    encoder on 64 candidates x 1024 points at 2 cm; train-mode BatchNorm over the whole batch, global max-pool, loss =
    <pooled, g>. Pooled features <= 1e-4 absolute in every case.
    Gradients: the whole backward signal enters through the 16 x 128 arg-max entries of the pooling, so it is SPIKY, and
    every one of the ~60 M ReLU decisions is a hard kink: an activation within fp32 round-off of zero fires in one
    implementation and not in the other (measured with the seeded weights: the deepest stage agrees to 2e-6, one flip
    at stage 3 then moves every shallower gradient by ~2e-3 — tools/fullsize_diag.py). Two checks therefore:
      kinkfree — BatchNorm shifts of +6 put every pre-activation six standard deviations above the kink (the masks are
                 still applied, just never ambiguous): each of the 39 parameter gradients <= 1e-3 of its own max-norm
                 (floor 1e-3 of the largest gradient), i.e. the judge's bar, at full size;
      seeded   — realistic weights: the kink-limited bound, 2e-2 of the max-norm and 1e-2 in relative L2."""
    from oracle import cpu_port
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.basic_blocks import SparseConvEncoder
    from instancerefer_amd.sparse import nn as spnn
    from instancerefer_amd.sparse.utils import voxelize
    dev = torch.device("cuda")
    if which == "scene":
        dd = S.make_batch(16, seed=321)
        pts = [torch.from_numpy(p) for p in dd["scene_points"]]
        voxel, nb = 0.05, 16
    else:
        dd = S.make_batch(16, seed=321, num_points=20000)
        pts = [torch.from_numpy(p) for ps in dd["instance_points"] for p in ps[:4]]
        voxel, nb = 0.02, 64
    allp = torch.cat(pts).to(dev)
    batch = torch.cat([torch.full((p.shape[0],), i, dtype=torch.int32) for i, p in enumerate(pts)]).to(dev)
    st = voxelize(allp[:, :3].contiguous(), allp.float(), batch, [voxel] * 3, nb)
    assert st.F.shape[0] > (300_000 if which == "scene" else 40_000)
    enc = SparseConvEncoder(7)
    sd = S.seeded_state_dict(enc, 4242)
    if weights == "kinkfree":
        for k in sd:
            if k.endswith("net.1.bias") or k.endswith("net.4.bias"):
                sd[k] = sd[k] + 6.0
    enc.load_state_dict(sd)
    enc = enc.to(dev).train()
    g = torch.from_numpy(np.random.default_rng(5).standard_normal((nb, 128)).astype(np.float32))
    pooled = spnn.GlobalMaxPooling()(enc(st))
    (pooled * g.to(dev)).sum().backward()
    torch.cuda.synchronize()
    params, order = cpu_port.pack_encoder_params({k: v.detach().cpu() for k, v in enc.state_dict().items()}, "")
    closs, cpooled, cgrads = cpu_port.encoder_fwd_bwd(st.C.cpu().numpy(), st.F.detach().cpu().numpy(), nb, params, g.numpy(),
                                                      wgrad_double=True)
    err = float(np.abs(pooled.detach().cpu().numpy() - cpooled).max())
    assert err <= 1e-4, ("pooled", err)
    named = dict(enc.named_parameters())
    refs, off = {}, 0
    for conv, bn in order:
        for name in (conv + ".kernel", bn + ".weight", bn + ".bias"):
            n = named[name].numel()
            refs[name] = cgrads[off:off + n]
            off += n
    assert off == cgrads.size and len(refs) == 39
    top = max(float(np.abs(r).max()) for r in refs.values())
    tol_max, tol_l2 = (1e-3, 1e-3) if weights == "kinkfree" else (2e-2, 1e-2)
    bad = {}
    for name, ref in refs.items():
        got = named[name].grad.detach().cpu().numpy().reshape(-1)
        e = float(np.abs(got - ref).max())
        l2 = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-3 * top))
        if e > tol_max * max(float(np.abs(ref).max()), 1e-3 * top) or l2 > tol_l2:
            bad[name] = (e, float(np.abs(ref).max()), l2)
    assert not bad, bad


# ---- BASELINE configs[4]: dense-scene stress — 200 k points, 64 instances, 16 candidates, multiview C0 = 135, bf16 ----------
STRESS = dict(batch_size=2, seed=4321, num_points=200000, num_instances=64, num_candidates=16, multiview=128)


@pytest.fixture(scope="module")
def stress_batch():
    from instancerefer_amd import synthetic as S
    return S.make_batch(**dict(STRESS))


def test_stress_scene_pyramid_tables_and_encoder_vs_c_port(lib, stress_batch):
    """configs[4]'s scene tensor (2 x 200 k points -> > 150 k voxels, 135 channels; reference scripts/train.py:74-75,
    lib/dataset.py:112-118): pyramid sortedness / parent consistency, neighbour-table symmetry, pair-list counts (exact),
    then the C0 = 135 scene encoder forward + backward in fp32 against the C/OpenMP port (pooled features <= 1e-4; with
    kink-free BatchNorm shifts every parameter gradient <= 1e-3 of its max-norm, incl. the 27 x 135 x 32 stem kernel)."""
    from oracle import cpu_port
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.basic_blocks import SparseConvEncoder
    from instancerefer_amd.sparse import nn as spnn
    from instancerefer_amd.sparse.utils import voxelize
    dev = torch.device("cuda")
    pts = [torch.from_numpy(p) for p in stress_batch["scene_points"]]
    assert pts[0].shape == (200000, 135)
    allp = torch.cat(pts).to(dev)
    batch = torch.cat([torch.full((p.shape[0],), i, dtype=torch.int32) for i, p in enumerate(pts)]).to(dev)
    st = voxelize(allp[:, :3].contiguous(), allp.float(), batch, [0.05] * 3, 2)
    lv = st.level()
    lv.build_pyramid(4)
    assert lv.n > 150_000 and st.F.shape[1] == 135
    stride, l = 1, lv
    for _ in range(5):
        assert bool((l.keys[1:] > l.keys[:-1]).all())
        assert bool((l.coords[:, :3] % stride == 0).all())
        tbl, _ = l.nbr27()
        t = tbl[:, :l.n]
        ar = torch.arange(l.n, device=dev, dtype=torch.int32)
        assert torch.equal(t[13], ar)
        for k in (1, 9):
            v = t[k] >= 0
            assert torch.equal(t[26 - k].index_select(0, t[k][v].long()), ar[v])
        _, _, counts, _ = l.pairs27()
        assert torch.equal(counts.long(), (t >= 0).sum(1))
        if stride < 16:
            dm = l.down()
            exp = torch.div(l.coords[:, :3], 2 * stride, rounding_mode="floor") * (2 * stride)
            got = dm.out_level.coords.index_select(0, dm.parent.long())
            assert torch.equal(got[:, :3], exp) and torch.equal(got[:, 3], l.coords[:, 3])
            l = dm.out_level
        stride *= 2
    enc = SparseConvEncoder(135)
    sd = S.seeded_state_dict(enc, 4242)
    for k in sd:
        if k.endswith("net.1.bias") or k.endswith("net.4.bias"):
            sd[k] = sd[k] + 6.0
    enc.load_state_dict(sd)
    enc = enc.to(dev).train()
    g = torch.from_numpy(np.random.default_rng(5).standard_normal((2, 128)).astype(np.float32))
    pooled = spnn.GlobalMaxPooling()(enc(st))
    (pooled * g.to(dev)).sum().backward()
    torch.cuda.synchronize()
    params, order = cpu_port.pack_encoder_params({k: v.detach().cpu() for k, v in enc.state_dict().items()}, "")
    _, cpooled, cgrads = cpu_port.encoder_fwd_bwd(st.C.cpu().numpy(), st.F.detach().cpu().numpy(), 2, params, g.numpy(),
                                                  wgrad_double=True)
    assert float(np.abs(pooled.detach().cpu().numpy() - cpooled).max()) <= 1e-4
    named = dict(enc.named_parameters())
    refs, off = {}, 0
    for conv, bn in order:
        for name in (conv + ".kernel", bn + ".weight", bn + ".bias"):
            n = named[name].numel()
            refs[name] = cgrads[off:off + n]
            off += n
    assert off == cgrads.size and refs["stem.0.net.0.kernel"].size == 27 * 135 * 32
    top = max(float(np.abs(r).max()) for r in refs.values())
    bad = {}
    for name, ref in refs.items():
        got = named[name].grad.detach().cpu().numpy().reshape(-1)
        e = float(np.abs(got - ref).max())
        if e > 1e-3 * max(float(np.abs(ref).max()), 1e-3 * top):
            bad[name] = (e, float(np.abs(ref).max()))
    assert not bad, bad


def test_stress_full_model_bf16_within_the_reordering_distance(lib, stress_batch):
    """configs[4] at its own dtype and size (B = 2): the FULL model in the bf16 mode BASELINE names (bf16 operands + bf16
    storage in both encoders: wide stem with bf16 output, pair-list weight-gradient reading x with a row stride and a
    separately typed dy, 2 x 16 candidates of 64 instances, relation graph on 153-feature nodes) against
    oracle/model_ref.py with the same rounding points (oracle/torchsparse/nn/emulate.py): identical discrete decisions,
    every score tensor within 2x the emulation's own reordering distance (tests/test_bf16_gpu.py explains the bar)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_bf16_gpu as TB
    (model, dd), runs = TB._model_runs(dict(STRESS), 99, 135, "bf16")
    assert list(dd["num_filtered_objs"]) == [16, 16]
    TB.check_model_against_emulation(model, dd, runs, "stress 200k x 64, C0 = 135, bf16")


def test_baseline_config2_full_model_bf16_vs_emulation(lib):
    """BASELINE configs[2] in ITS OWN dtype and size: the full model, 16 scenes x 50 k points, 8 instances, 4 candidates,
    30 tokens, bf16 operands + bf16 storage in both encoders (every 32/64/128-channel convolution of the executor on the
    third-generation kernel, csrc/irx_spconv3.hip), training mode, against oracle/model_ref.py with the same rounding points
    (oracle/torchsparse/nn/emulate.py): identical discrete decisions (candidates, cluster labels), every score tensor within
    2x the emulation's own reordering distance (fp32 vs float64 sums; tests/test_bf16_gpu.py explains why that is the bar a
    chain of rounded layers admits), total gradient norm within 5 %. Reference dtype: fp32, models/basic_blocks.py:59-95."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_bf16_gpu as TB
    from instancerefer_amd import _lib
    assert _lib.get_knob("spconv3") == 1
    (model, dd), runs = TB._model_runs(dict(batch_size=16, seed=123), 2024, 7, "bf16")
    assert list(dd["num_filtered_objs"]) == [4] * 16 and dd["attribute_scores"].shape == (64,)
    TB.check_model_against_emulation(model, dd, runs, "configs[2]: 16 x 50k, bf16")


@pytest.mark.parametrize("which", ["full", "attr_only", "full_elementwise"])
def test_whole_model_at_baseline_size_vs_oracle(lib, which):
    """BASELINE configs[2] / configs[1] shapes through the WHOLE model, not only the encoders: 16 scenes x 50 k points, 8
    instances, 4 candidates each (64 candidates), 30-token utterances, fp32, training mode — language module, candidate
    encoder + max-pool, relation graph, BEV + scene head, the three matching heads, get_loss — vs oracle/model_ref.py
    (pinned to the reference's own models/instancerefer.py:37-70 output by tests/test_oracle_cpu.py). "attr_only" is
    configs[1]'s model: relation_module = scene_module = None (reference models/instancerefer.py:24-34,56-68).
    Forward tensors <= 1e-4 absolute (the north star's bar); loss terms <= 1e-4; gradient norms 2e-3 per parameter
    (floor: 1e-3 of the total), total 1e-3. "full_elementwise": the full model with the encoders' ReLUs kept away from their
    kinks (helpers.kink_free_state_dict) and EVERY parameter gradient compared element by element: 1e-3 of the tensor's
    largest entry (floor 1e-6 of the largest entry of any gradient)."""
    from helpers import elementwise_grad_report, kink_free_state_dict
    elementwise = which == "full_elementwise"
    if elementwise:
        which = "full"
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.instancerefer import InstanceRefer
    from instancerefer_amd.loss_helper import DatasetConfig, compute_lang_classification_loss, get_loss
    from oracle.model_ref import InstanceRefer as OracleModel, oracle_data_dict
    dev = torch.device("cuda")
    args = S.default_args()
    if which == "attr_only":
        args.relation_module = None
        args.scene_module = None
    model = InstanceRefer(7, args)
    sd = S.seeded_state_dict(model, 2024)
    if elementwise:
        sd = kink_free_state_dict(sd)
    model.load_state_dict(sd)
    oracle = OracleModel(7, args)
    oracle.load_state_dict(sd)
    assert hasattr(oracle, "relation") == (which == "full") and hasattr(model, "relation") == (which == "full")
    for m in list(model.modules()) + list(oracle.modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.to(dev).train()
    oracle.train()
    host = S.make_batch(16, seed=123)                    # bench.py's batch: 50 k points, 8 instances, 4 candidates
    dd = model(S.to_device(dict(host), dev))
    od = oracle(oracle_data_dict(dict(host)))
    keys = ["lang_scores", "obj_feats", "attribute_scores"]
    if which == "full":
        dd, od = get_loss(dd, DatasetConfig()), get_loss(od, DatasetConfig())
        keys += ["relation_scores", "scene_scores", "seg_scores", "vis_atten", "loss", "ref_loss", "lang_loss", "seg_loss"]
        ld, lo = dd["loss"], od["loss"]
    else:
        # configs[1] has no relation / scene scores for get_loss's sum: language CE + a dense functional of the scores
        ld = compute_lang_classification_loss(dd) + (dd["attribute_scores"] * dd["attribute_scores"]).sum()
        lo = compute_lang_classification_loss(od) + (od["attribute_scores"] * od["attribute_scores"]).sum()
    assert list(dd["num_filtered_objs"]) == list(od["num_filtered_objs"]) == [4] * 16
    assert dd["attribute_scores"].shape == (64,)
    worst = {k: float((dd[k].detach().cpu() - od[k].detach()).abs().max()) for k in keys}
    assert all(v <= 1e-4 for v in worst.values()), worst
    ld.backward()
    lo.backward()
    gp = dict(model.named_parameters())
    if elementwise:
        bad, worst_ratio, lines = elementwise_grad_report(gp, dict(oracle.named_parameters()))
        import os
        os.makedirs('gpurun_out', exist_ok=True)
        open('gpurun_out/elementwise_fullsize.txt', 'w').write('\n'.join(lines) + '\n')
        print("element-wise gradients at 16 x 50 k: worst error / bar = %.3f" % worst_ratio)
        assert not bad, bad
        return
    tot_o = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in oracle.parameters() if p.grad is not None)))
    tot_d = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
    assert abs(tot_d - tot_o) <= 1e-3 * tot_o, (tot_d, tot_o)
    bad = {}
    for n, p in oracle.named_parameters():
        if p.grad is None:
            assert gp[n].grad is None or float(gp[n].grad.abs().max()) == 0.0, n
            continue
        exp, got = float(p.grad.double().norm()), float(gp[n].grad.double().norm())
        if abs(got - exp) > 2e-3 * max(exp, 1e-3 * tot_o):
            bad[n] = (got, exp)
    assert not bad, bad
