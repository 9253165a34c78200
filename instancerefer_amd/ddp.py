"""Data parallelism for the hot path: one process per GPU, scenes sharded across ranks, ONE gradient
all-reduce per step over RCCL/xGMI (torch.distributed backend "nccl" is RCCL on ROCm; "gloo" on CPU tests).

The reference is single-GPU; this is the only exchange step the path needs: all
parameter gradients live as views into one flat fp32 buffer (~8.0 M params = 32 MB), so the all-reduce is
a single large collective — the right shape for xGMI's point-to-point links — with no per-tensor launches
and no copy in or out. BatchNorm statistics stay per-rank (standard DDP semantics; not SyncBN).
Ranks whose shard has no scene with >= 2 candidates still enter the collective with zero gradients.
"""
import torch
import torch.distributed as dist


class FlatGradAllReduce:
    def __init__(self, params, world_size=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        self.views = []
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            self.views.append(v)
            off += p.numel()
        self.world_size = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)

    def zero_grad(self):
        """Zero the flat buffer and (re)attach the views as .grad so autograd accumulates in place."""
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce(self):
        """Average gradients over ranks (sum / world). No-op for a single process."""
        if self.world_size > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(self.world_size)


def shard_range(n_items, rank, world_size):
    """Contiguous shard [lo, hi) of n_items for `rank` (DistributedSampler-like, equal sizes required
    for loss parity with a single-process global batch: each rank divides by its local batch size)."""
    per = n_items // world_size
    return rank * per, (rank + 1) * per
