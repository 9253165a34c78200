"""Flat-buffer training state: every parameter is a view into ONE fp32 buffer, gradients are gathered into a
second flat buffer with a single multi-tensor copy, all-reduced once over RCCL/xGMI (ddp.py) and applied by
ONE fused Adam launch (csrc/irx_optim.hip). Semantics = torch.optim.Adam(lr, betas, eps, weight_decay), the
optimizer of the reference (scripts/train.py:121)."""
import torch
import torch.distributed as dist

from . import _lib


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, world_size=None):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam needs parameters on a HIP device")
        # every parameter starts on a 64-byte boundary of the flat buffer (the conv kernels need 16-byte aligned
        # weight pointers for their 16 B/lane loads); the padding elements stay zero in all four buffers
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 15) // 16 * 16
        n = off
        self.n = n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        # gradient slots: views of flat_g with the parameters' shapes (same offsets as the parameters in flat_p)
        self._slots = [self.flat_g[off:off + p.numel()].view_as(p) for p, off in zip(self.params, self.offsets)]
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):   # re-home every parameter inside the flat buffer
                view = self.flat_p[off:off + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.world_size = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)

    def zero_grad(self):
        for p in self.params:
            p.grad = None        # autograd then hands over freshly computed gradients without an add kernel

    def gather_grads(self):
        """All .grad tensors -> their slots in flat_g with one multi-tensor copy (the padding between slots stays zero;
        a missing grad counts as zero)."""
        grads = [p.grad for p in self.params]
        if any(g is None for g in grads):
            grads = [g if g is not None else torch.zeros_like(p) for g, p in zip(grads, self.params)]
        torch._foreach_copy_(self._slots, grads)

    def all_reduce(self):
        if self.world_size > 1:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM)

    def step(self):
        self.step_count += 1
        _lib.call("irx_adam_step", _lib.ptr(self.flat_p), _lib.ptr(self.flat_g), _lib.ptr(self.exp_avg),
                  _lib.ptr(self.exp_avg_sq), self.n, float(self.lr), float(self.betas[0]), float(self.betas[1]),
                  float(self.eps), float(self.weight_decay), self.step_count, 1.0 / self.world_size,
                  _lib.stream_ptr())

    def backward_step(self):
        """After loss.backward(): gather -> all-reduce -> Adam."""
        self.gather_grads()
        self.all_reduce()
        self.step()
