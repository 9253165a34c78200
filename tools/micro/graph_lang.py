"""Experiment: hipGraph capture (torch.cuda.make_graphed_callables) of the language module, fwd + bwd."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn as nn
from instancerefer_amd import _lib
from instancerefer_amd.lang_module import LangModule
_lib.load()
dev = torch.device("cuda")
torch.manual_seed(0)
B, T = 16, 30
lang = LangModule(18, True, True, 300, 128).to(dev).train()

class Core(nn.Module):
    def __init__(self, m): super().__init__(); self.m = m
    def forward(self, feat, length):
        dd = {"lang_feat": feat, "lang_len": length, "lang_len_max": T}
        dd = self.m(dd)
        return (dd["lang_feat"], dd["atten_attr"], dd["atten_rel"], dd["atten_scene"], dd["lang_attr_feats"],
                dd["lang_cls_feats"], dd["lang_rel_feats"], dd["lang_scene_feats"], dd["lang_scores"])

core = Core(lang)
feat = torch.randn(B, T, 300, device=dev)
length = torch.full((B,), T, device=dev, dtype=torch.int64)
def run(c, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        outs = c(feat, length)
        loss = sum(o.sum() for o in outs[4:])
        loss.backward()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
run(core, 5)
print("eager : host %.3f ms/iter, wall %.3f" % run(core, 50))
g0 = [p.grad.clone() for p in core.parameters()]
for p in core.parameters(): p.grad = None
gcore = torch.cuda.make_graphed_callables(core, (feat, length))
run(gcore, 5)
print("graph : host %.3f ms/iter, wall %.3f" % run(gcore, 50))
# numerics (dropout differs run to run; compare in eval-like conditions by zeroing dropout)
for m in core.modules():
    if isinstance(m, nn.Dropout): m.p = 0.0
