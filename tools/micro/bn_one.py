"""Dev: one BatchNorm forward + backward size in a loop (for rocprofv3 --kernel-trace --stats): python tools/micro/bn_one.py n c bf"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from instancerefer_amd import _lib
lib = _lib.load(); dev = torch.device('cuda')
n, c, bf = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dt = torch.bfloat16 if bf else torch.float32
x = torch.randn(n, c, device=dev).to(dt); y = torch.empty_like(x); dy = torch.randn(n, c, device=dev).to(dt); dx = torch.empty_like(x)
mean = torch.zeros(c, device=dev); inv = torch.ones(c, device=dev); g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
dg = torch.empty(c, device=dev); db = torch.empty(c, device=dev)
wsb = lib.irx_bn_workspace_bytes(n, c); ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
s = _lib.stream_ptr(); P = lambda t: t.data_ptr()
for _ in range(50):
    lib.irx_bn_forward_ex(P(x), n, c, 1e-5, 0.1, P(g), P(b), None, 1, P(mean), P(inv), None, None, P(y), P(ws), wsb, s, bf, bf, bf)
torch.cuda.synchronize()
