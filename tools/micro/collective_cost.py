"""Dev: where does the one-rank RCCL path of the training step spend its extra time? Host time of every phase of
FlatAdam.backward_step with collectives forced on, and the per-call host / GPU cost of an async all-reduce."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.ones(8_000_000, device=dev)
for n in (8_000_000, 1_000_000, 1024):
    v = x[:n]
    for _ in range(5): dist.all_reduce(v, async_op=True).wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): dist.all_reduce(v, async_op=True).wait()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("all_reduce %9d floats: host issue %.1f us per call, drained total %.1f us per call" % (n, (t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
side = torch.cuda.Stream()
for _ in range(5):
    with torch.cuda.stream(side): w = dist.all_reduce(x, async_op=True)
    w.wait()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    with torch.cuda.stream(side): w = dist.all_reduce(x, async_op=True)
    w.wait()
t1 = time.perf_counter(); torch.cuda.synchronize()
print("issued on a side stream: host %.1f us per call" % ((t1 - t0) / 50 * 1e6))
dist.barrier(); dist.destroy_process_group()
