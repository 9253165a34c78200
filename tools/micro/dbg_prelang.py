import sys, os, argparse, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from instancerefer_amd import synthetic as S, heads
from instancerefer_amd.loss_helper import DatasetConfig
from instancerefer_amd.optim import FlatAdam
dev = torch.device("cuda")
bench.step_fn.cfg = DatasetConfig()
res = {}
for pre in (True, False):
    heads.PRE_LANG = pre
    torch.manual_seed(11)
    model = bench.build_model(argparse.Namespace(), "full", dev)
    if os.environ.get("NODROP"):
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout): m.p = 0.0
    resident = S.to_device(S.make_batch(4, seed=33, num_points=6000, num_instances=6, num_candidates=3, points_per_instance=256), dev)
    lidar = resident.pop("lidar")
    resident["lidar_F"], resident["lidar_C"], resident["B"] = lidar.F, lidar.C, 4
    opt = FlatAdam(model.parameters(), lr=1e-3, weight_decay=1e-5, world_size=1)
    # one step without the optimizer update: gather gradients
    from instancerefer_amd.loss_helper import get_loss
    dd = bench.fresh_batch(resident)
    opt.zero_grad()
    loss = get_loss(model(dd), bench.step_fn.cfg)["loss"]
    loss.backward()
    opt.gather_grads()
    torch.cuda.synchronize()
    res[pre] = (float(loss), {n: opt._slots[i].clone() for i, (n, p) in enumerate(model.named_parameters())})
print("loss", res[True][0], res[False][0])
for n in res[True][1]:
    a, b = res[True][1][n], res[False][1][n]
    if not torch.equal(a, b):
        print("%-50s max|d| %.3e  max|g| %.3e" % (n, float((a - b).abs().max()), float(b.abs().max())))
