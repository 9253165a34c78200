"""Dev: the radix sort alone (irx_sort_pairs_u64) on Morton-like keys: us per call for n keys / end_bit bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from instancerefer_amd.sparse import functional as F_
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(1)
for n, bits in ((800000, 52), (489000, 52), (61000, 55), (800000, 8), (60000, 52)):
    keys = torch.randint(0, 2 ** min(bits, 62), (n,), generator=g, dtype=torch.int64).to(dev)
    ref = torch.sort(keys, stable=True)
    out, order = F_.sort_keys(keys, bits)
    assert torch.equal(out, ref[0]) and torch.equal(order.long(), ref[1]), "sort mismatch"
    for _ in range(3):
        F_.sort_keys(keys, bits)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        F_.sort_keys(keys, bits)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    passes = (bits + 7) // 8
    print("n %7d bits %2d: %7.1f us  (%d passes, %.1f us / pass, %.0f GB/s of 2 x 12 B x n per pass)" % (n, bits, us, passes, us / passes, passes * 24.0 * n / us / 1e3))
