"""Dev: GRU layer gradients, C++ node (irx_gru_wgrad) vs Python node (ATen GEMMs) vs torch.nn.GRU on the CPU in float64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
from instancerefer_amd import dense
torch.manual_seed(3)
gru = torch.nn.GRU(256, 128, num_layers=2, batch_first=True, bidirectional=True)
lens = torch.tensor([30, 7, 41, 1, 18, 30, 25, 12])
x = torch.randn(8, 41, 256)
g = torch.randn(8, 41, 256)
g64 = gru.double()
xr = x.double().clone().requires_grad_(True)
yr, _ = pad_packed_sequence(g64(pack_padded_sequence(xr, lens, batch_first=True, enforce_sorted=False))[0], batch_first=True)
yr.backward(g.double())
ref = {n: p.grad.clone() for n, p in g64.named_parameters()}
res = {}
for be in ("py", "auto"):
    dense.GRU_BACKEND = be
    gd = torch.nn.GRU(256, 128, num_layers=2, batch_first=True, bidirectional=True)
    gd.load_state_dict({k: v.float() for k, v in g64.state_dict().items()})
    gd = gd.cuda()
    xd = x.clone().cuda().requires_grad_(True)
    yd = dense.gru_packed(gd, xd, lens.cuda(), 41)
    yd.backward(g.cuda())
    res[be] = {n: p.grad.double().cpu() for n, p in gd.named_parameters()}
    res[be]["x"] = xd.grad.double().cpu()
ref["x"] = xr.grad
for n in ref:
    e = {be: float((res[be][n] - ref[n]).norm() / ref[n].norm()) for be in res}
    print("%-28s rel.err py %.2e  cpp %.2e" % (n, e["py"], e["auto"]))
