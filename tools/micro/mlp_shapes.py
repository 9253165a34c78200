import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench, argparse
from instancerefer_amd import dense, synthetic as S
orig = dense.mlp2
seen = {}
def spy(seq, x):
    mods = list(seq)
    key = (tuple(x.shape), mods[0].in_features, mods[0].out_features, mods[-1].out_features, type(mods[1]).__name__, len(mods))
    seen[key] = seen.get(key, 0) + 1
    return orig(seq, x)
dense.mlp2 = spy
import instancerefer_amd.attribute_module as am, instancerefer_amd.relation_module as rm, instancerefer_amd.scene_module as sm
for m in (am, rm, sm):
    if hasattr(m, 'mlp2'):
        m.mlp2 = spy
dev = torch.device('cuda')
from instancerefer_amd.loss_helper import DatasetConfig
bench.step_fn.cfg = DatasetConfig()
model = bench.build_model(argparse.Namespace(), "full", dev)
resident = S.to_device(S.make_batch(16, seed=21), dev)
lidar = resident.pop("lidar")
resident["lidar_F"], resident["lidar_C"], resident["B"] = lidar.F, lidar.C, 16
from instancerefer_amd.optim import FlatAdam
opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
bench.step_fn(model, resident, "full", None, opt, None)
torch.cuda.synchronize()
for k, v in seen.items():
    print(k, v)
