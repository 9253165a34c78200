"""One autograd node per head (instancerefer_amd/heads.py, csrc/heads_nodes.cpp) against the operator-by-operator heads
(attribute_module.py / scene_module.py / dense.TotalLossFn): the same C-ABI calls in the same order, so every forward tensor, the
loss and EVERY parameter gradient must be bit-identical with dropout off — with the gradients returned to autograd and with the
gradients written into optim.FlatAdam's slots (native sink). Reference semantics: models/attribute_module.py:105-126,
models/scene_module.py:61-106, lib/loss_helper.py:196-269."""
import numpy as np
import pytest
import torch

from helpers import GOLDEN_CFG, WEIGHT_SEED

pytestmark = pytest.mark.gpu

KEYS = ("obj_feats", "attribute_scores", "relation_scores", "scene_scores", "seg_scores", "vis_atten", "lang_scores", "loss",
        "ref_loss", "lang_loss", "seg_loss", "seg_acc")


def _model(dropout):
    from instancerefer_amd import synthetic as S
    from instancerefer_amd.instancerefer import InstanceRefer
    dev = torch.device("cuda")
    model = InstanceRefer(7, S.default_args())
    model.load_state_dict(S.seeded_state_dict(model, WEIGHT_SEED))
    if not dropout:
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    return model.to(dev).train()


def _step(model, fused, monkeypatch, opt=None, cfg=GOLDEN_CFG):
    from instancerefer_amd import heads, synthetic as S
    from instancerefer_amd.loss_helper import DatasetConfig, get_loss
    monkeypatch.setattr(heads, "FUSED", fused)
    before = dict(heads.CALLS)
    dd = S.to_device(S.make_batch(**dict(cfg)), torch.device("cuda"))
    if opt is not None:
        opt.zero_grad()
    dd = get_loss(model(dd), DatasetConfig())
    dd["loss"].backward()
    if opt is not None:
        opt.gather_grads()
    torch.cuda.synchronize()
    taken = {k: heads.CALLS[k] - before[k] for k in before}
    return dd, taken


@pytest.mark.parametrize("streams", ["1", "0"])
def test_head_nodes_equal_the_per_operator_heads(lib, monkeypatch, streams):
    """outputs, loss and every parameter gradient (through autograd's AccumulateGrad), three-stream and one-stream forward"""
    from instancerefer_amd import instancerefer as IR
    monkeypatch.setattr(IR, "_STREAMS_ENV", streams)
    monkeypatch.setattr(IR, "_STREAMS", streams != "0")
    ma, mb = _model(False), _model(False)
    da, taken = _step(ma, True, monkeypatch)
    assert taken == {"scene_head": 1, "attr_scene": 1, "total_loss": 1, "relation_head": 1, "lang_pool": 1}, taken
    db, taken = _step(mb, False, monkeypatch)
    assert not any(taken.values()), taken
    for k in KEYS:
        assert torch.equal(da[k].detach(), db[k].detach()), (k, float((da[k].detach() - db[k].detach()).abs().max()))
    nb = dict(mb.named_parameters())
    n_checked = 0
    for n, p in ma.named_parameters():
        q = nb[n]
        assert (p.grad is None) == (q.grad is None), n
        if p.grad is not None:
            assert torch.equal(p.grad, q.grad), (n, float((p.grad - q.grad).abs().max()))
            n_checked += 1
    assert n_checked > 150
    for (n, a), (_, b) in zip(ma.named_buffers(), mb.named_buffers()):       # running statistics and batch counters
        assert torch.equal(a, b), n


def test_head_nodes_deliver_into_the_optimizer_slots(lib, monkeypatch):
    """with optim.FlatAdam the nodes write their parameter gradients straight into the flat gradient buffer: equal, slot by slot, to the
    gradients the per-operator heads hand to autograd; a second backward before zero_grad() falls back to accumulation"""
    from instancerefer_amd.optim import FlatAdam
    ma, mb = _model(False), _model(False)
    opt = FlatAdam(ma.parameters(), lr=1e-3, weight_decay=0.0, world_size=1, module=ma)
    da, taken = _step(ma, True, monkeypatch, opt)
    assert taken["scene_head"] == 1 and taken["attr_scene"] == 1
    prod, nparams = opt.native_delivered()
    assert prod >= 4 and nparams >= 21 + 18 + 20 + 8          # scene head, attribute / scene-score head, relation head, language pooling
    db, _ = _step(mb, False, monkeypatch)
    assert torch.equal(da["loss"].detach(), db["loss"].detach())
    nb = dict(mb.named_parameters())
    index = {id(p): i for i, p in enumerate(opt.params)}
    for n, p in ma.named_parameters():
        g = nb[n].grad
        slot = opt._slots[index[id(p)]]
        if g is None:
            assert float(slot.abs().max()) == 0.0, n
        else:
            assert torch.equal(slot, g), (n, float((slot - g).abs().max()))


def test_head_nodes_with_dropout_train_and_respect_the_seed(lib, monkeypatch):
    """dropout on: the masks come from the library's counter-based hash keyed by torch's generator — same seed, same step; another
    seed, another step; the loss stays finite and every parameter receives a gradient"""
    from instancerefer_amd import heads
    losses = []
    for seed in (5, 5, 6):
        torch.manual_seed(seed)
        m = _model(True)
        dd, taken = _step(m, True, monkeypatch)
        assert taken["scene_head"] == 1 and taken["attr_scene"] == 1
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
        losses.append(float(dd["loss"].detach()))
    assert losses[0] == losses[1] and losses[0] != losses[2] and np.isfinite(losses).all()


def test_dropout_flat_is_a_scaled_bernoulli_mask(lib):
    """irx_dropout_flat: kept elements scaled by 1 / (1 - p), the keep rate matches p, the same seed reproduces the mask (the backward
    is the same call on the gradient), p = 0 is the identity, odd sizes and unaligned tails are covered"""
    from instancerefer_amd import _lib
    dev = torch.device("cuda")
    for n in (1, 7, 4096 + 3, 231 * 128 * 16):
        x = torch.rand(n, device=dev) + 0.5
        y, y2, g = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        _lib.call("irx_dropout_flat", _lib.ptr(x), n, 0.15, 1234567, _lib.ptr(y), _lib.stream_ptr())
        _lib.call("irx_dropout_flat", _lib.ptr(x), n, 0.15, 1234567, _lib.ptr(y2), _lib.stream_ptr())
        assert torch.equal(y, y2)
        kept = y != 0
        assert torch.allclose(y[kept], x[kept] / 0.85, rtol=1e-6)
        if n > 1000:
            assert abs(float(kept.float().mean()) - 0.85) < 0.02
            _lib.call("irx_dropout_flat", _lib.ptr(x), n, 0.15, 7654321, _lib.ptr(y2), _lib.stream_ptr())
            assert not torch.equal(y, y2)
        ones = torch.ones_like(x)
        _lib.call("irx_dropout_flat", _lib.ptr(ones), n, 0.15, 1234567, _lib.ptr(g), _lib.stream_ptr())
        assert torch.equal(g != 0, kept)
        _lib.call("irx_dropout_flat", _lib.ptr(x), n, 0.0, 99, _lib.ptr(y), _lib.stream_ptr())
        assert torch.equal(y, x)
    # an unaligned view (element offset 1) takes the scalar tail path everywhere
    base = torch.rand(1001, device=dev) + 0.5
    x = base[1:]
    y = torch.empty(1004, device=dev)[1:1001]
    _lib.call("irx_dropout_flat", x.data_ptr(), 1000, 0.5, 42, y.data_ptr(), _lib.stream_ptr())
    ref = torch.empty(1000, device=dev)
    xc = x.clone()
    _lib.call("irx_dropout_flat", _lib.ptr(xc), 1000, 0.5, 42, _lib.ptr(ref), _lib.stream_ptr())
    assert torch.equal(y, ref)


def test_encoder_backward_with_weight_gradients_on_a_lent_stream(lib, monkeypatch):
    """IRX_ENC_DC2 / IRX_ENC_WSTREAM (include/irx.h): the scene encoder's backward issues its weight gradients on a second stream,
    two gradient scratches alternating by layer — same kernels, same operands: every parameter gradient bit-identical to the
    one-stream order (the option is off by default: measured slower, instancerefer.py _WGRAD_LANG)"""
    from instancerefer_amd import instancerefer as IR
    monkeypatch.setattr(IR, "_STREAMS_ENV", "1")
    monkeypatch.setattr(IR, "_STREAMS", True)
    grads = {}
    for lent in (True, False):
        monkeypatch.setattr(IR, "_WGRAD_LANG", lent)
        m = _model(False)
        dd, _ = _step(m, True, monkeypatch)
        assert (m.scene.net.__dict__.get("_irx_wgrad_stream") is not None) == lent
        grads[lent] = ({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}, dd["loss"].detach().clone())
    assert torch.equal(grads[True][1], grads[False][1])
    assert grads[True][0].keys() == grads[False][0].keys() and len(grads[True][0]) > 150
    for n, g in grads[True][0].items():
        assert torch.equal(g, grads[False][0][n]), n


def test_two_optimizers_keep_their_native_sinks(lib, monkeypatch):
    """the sink's liveness is the optimizer's own (a cell of its record tensor), not a process-wide generation — a second
    model + optimizer in the process must not switch the first one's C++ nodes to the ordinary-gradient path; re-homing the SAME
    parameters in a new optimizer retires the old one's sink."""
    from instancerefer_amd.optim import FlatAdam
    ma, mb = _model(False), _model(False)
    oa = FlatAdam(ma.parameters(), lr=1e-3, weight_decay=0.0, world_size=1, module=ma)
    ob = FlatAdam(mb.parameters(), lr=1e-3, weight_decay=0.0, world_size=1, module=mb)      # constructed AFTER oa
    _step(ma, True, monkeypatch, oa)
    _step(mb, True, monkeypatch, ob)
    pa, pb = oa.native_delivered(), ob.native_delivered()
    assert pa[0] >= 4 and pb[0] >= 4 and pa == pb, (pa, pb)
    oc = FlatAdam(ma.parameters(), lr=1e-3, weight_decay=0.0, world_size=1, module=ma)      # takes ma's parameters over
    assert oa._native_rec[256, 0] == 0 and oc._native_rec[256, 0] == 1 and ob._native_rec[256, 0] == 1
    _step(ma, True, monkeypatch, oc)
    assert oc.native_delivered()[0] >= 4
