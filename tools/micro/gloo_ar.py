import os, time, torch, torch.distributed as dist
dist.init_process_group("gloo")
torch.cuda.set_device(0)
x = torch.randn(8_000_000, device="cuda")
for _ in range(2): dist.all_reduce(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): dist.all_reduce(x)
torch.cuda.synchronize()
if dist.get_rank() == 0: print("gloo all_reduce 32MB cuda tensor: %.1f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
