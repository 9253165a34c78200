"""The drop-in boundary as the reference's scripts see it: with `instancerefer_amd/compat` in front of the
reference checkout on sys.path, `models.*` / `lib.loss_helper` / `lib.eval_helper` / `torchsparse` / `torch_geometric`
resolve to the irx packages and everything else still comes from the reference. No compute here (no GPU): imports,
class construction, state-dict keys, and the fail-loudly rule."""
import importlib
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "instancerefer_amd", "compat")
REF = "/root/reference"
_NAMES = ("models", "lib", "torchsparse", "torch_geometric")


@pytest.fixture
def compat_path():
    """sys.path as `PYTHONPATH=<repo>:<repo>/instancerefer_amd/compat` + the reference's own appends would leave it."""
    saved_path = list(sys.path)
    saved_mods = {k: v for k, v in sys.modules.items() if k.split(".")[0] in _NAMES}
    for k in saved_mods:
        del sys.modules[k]
    sys.path.insert(0, COMPAT)
    if os.path.isdir(REF):
        sys.path.append(REF)                     # scripts/train.py:11 appends the checkout root
    yield
    for k in [k for k in sys.modules if k.split(".")[0] in _NAMES]:
        del sys.modules[k]
    sys.modules.update(saved_mods)
    sys.path[:] = saved_path


def test_reference_import_lines_resolve_to_the_drop_in(compat_path):
    import instancerefer_amd.instancerefer as prod
    from models.instancerefer import InstanceRefer                    # scripts/train.py:18, scripts/eval.py:18
    from lib.loss_helper import get_loss                              # lib/solver.py:17, scripts/eval.py:15
    from lib.eval_helper import get_eval                              # lib/solver.py:18, scripts/eval.py:16
    import torchsparse.nn as spnn                                     # models/basic_blocks.py:4
    from torchsparse import SparseTensor                              # models/basic_blocks.py:6
    from torchsparse.utils import sparse_collate_fn, sparse_collate_tensors, sparse_quantize   # lib/dataset.py:17
    from torch_geometric.nn import MessagePassing, knn                # models/basic_blocks.py:7
    import instancerefer_amd.loss_helper as pl
    import instancerefer_amd.eval_helper as pe
    from instancerefer_amd.sparse import nn as psnn, tensor as pst
    from instancerefer_amd.graph import nn as pg
    assert InstanceRefer is prod.InstanceRefer and get_loss is pl.get_loss and get_eval is pe.get_eval
    assert spnn.Conv3d is psnn.Conv3d and spnn.GlobalMaxPooling is psnn.GlobalMaxPooling and SparseTensor is pst.SparseTensor
    assert MessagePassing is pg.MessagePassing and knn is pg.knn and callable(sparse_quantize) and callable(sparse_collate_fn)
    for name in ("lang_module", "attribute_module", "relation_module", "scene_module", "basic_blocks"):
        m = importlib.import_module("models." + name)
        assert m.__file__.startswith(COMPAT), m.__file__
    from models.basic_blocks import SparseConvEncoder, DynamicEdgeConv, ToDenseBEVConvolution, SparseCrop  # noqa: F401
    from instancerefer_amd import synthetic as S
    model = InstanceRefer(input_feature_dim=7, args=S.default_args())    # scripts/train.py:77-80
    keys = set(model.state_dict())
    assert {"attribute.net.stem.0.net.0.kernel", "scene.to_bev.1.kernel", "relation.gcn.mlp.0.weight",
            "lang.gru.weight_ih_l0"} <= keys


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
def test_reference_modules_run_on_the_shims(compat_path):
    """The reference's OWN models/basic_blocks.py, executed against the alias packages: every class constructs and
    the parameters carry the reference's state-dict layout. Modules the reference keeps (lib.scheduler_helper) still
    resolve through the merged namespace package."""
    spec = importlib.util.spec_from_file_location("ref_basic_blocks", os.path.join(REF, "models", "basic_blocks.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from instancerefer_amd.graph.nn import MessagePassing
    from instancerefer_amd.sparse import nn as psnn
    blocks = [ref.BasicConvolutionBlock(7, 32, 3), ref.ResidualBlock(32, 32, 3), ref.ResidualBlock(32, 64, 3),
              ref.SparseConvEncoder(7), ref.BEVEncoder(135), ref.DynamicEdgeConv(25, 128, k=8, num_classes=18),
              ref.SparseCrop(np.array([0, 0, 0]), np.array([240, 400, 80]))]
    enc = blocks[3]
    assert isinstance(enc.stem[0].net[0], psnn.Conv3d) and isinstance(blocks[5], MessagePassing)
    assert tuple(enc.stage2[0].net[0].kernel.shape) == (8, 64, 128) and tuple(blocks[4].stem[0].net[0].kernel.shape) == (27, 135, 32)
    assert tuple(blocks[2].downsample[0].kernel.shape) == (32, 64)                 # kernel_size 1: (Cin, Cout)
    import instancerefer_amd.basic_blocks as prod
    mine = prod.SparseConvEncoder(7)
    assert list(enc.state_dict()) == list(mine.state_dict())
    assert list(blocks[5].state_dict()) == list(prod.DynamicEdgeConv(25, 128, k=8, num_classes=18).state_dict())
    sched = importlib.import_module("lib.scheduler_helper")                       # stays the reference's
    assert sched.__file__.startswith(REF) and hasattr(sched, "BNMomentumScheduler")
    assert importlib.import_module("lib.loss_helper").__file__.startswith(COMPAT)


def test_graph_surface_fails_loudly_without_a_hip_device():
    from instancerefer_amd.graph import nn as gnn
    x = torch.rand(6, 3)
    with pytest.raises(RuntimeError, match="HIP device"):
        gnn.knn(x, x[:2], 2)
    mp_ = gnn.MessagePassing(aggr='max')
    with pytest.raises(RuntimeError, match="HIP device"):
        mp_.propagate(torch.tensor([[0, 1], [0, 0]]), x=(torch.rand(2, 4), torch.rand(1, 4)))
    with pytest.raises(NotImplementedError):
        gnn.MessagePassing(aggr='min')
