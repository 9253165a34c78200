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
        # Gradient sinks: a producer that computes ALL gradients of a parameter group in one native call (the sparse
        # encoders' executor) may write them straight into the group's slots of flat_g and hand autograd `None` for those
        # parameters: no AccumulateGrad node, no .grad tensor, no copy at gather time for ~half of the parameters.
        self._index = {id(p): i for i, p in enumerate(self.params)}
        for p, slot in zip(self.params, self._slots):
            p._irx_sink = (self, slot)
        self._direct = set()            # parameter indices whose slot already holds this step's gradient
        self._direct_groups = set()     # producer keys that delivered since the last zero_grad()
        self._gather_cache = {}
        self._pending = []              # events of sink deliveries not yet waited for
        self._todo_last = list(range(len(self.params)))
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.world_size = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)

    def zero_grad(self):
        # autograd then hands over freshly computed gradients without an add kernel. Parameters whose gradient went
        # through the sink never held a .grad tensor (gather_grads clears the rare exception), so they are skipped.
        if self._direct:
            for i in self._todo_last:
                self.params[i].grad = None
        else:
            for p in self.params:
                p.grad = None
        self._direct.clear()
        self._direct_groups.clear()

    # ---- gradient-sink protocol (see __init__) ----
    def sink_slots(self, key, params):
        """-> list of slot tensors for `params` if the producer `key` may deliver directly now (every parameter is ours
        and the producer has not delivered since the last zero_grad()), else None."""
        if key in self._direct_groups:
            return None                  # second backward before the step: fall back to ordinary accumulation
        try:
            return [self._slots[self._index[id(p)]] for p in params]
        except KeyError:
            return None

    def sink_delivered(self, key, params, lane=None):
        """Called by the producer right after it enqueued the kernels that write the slots (on ITS current stream, which
        for the scene encoder is not the optimizer's): an event makes gather_grads() wait for them. (Autograd only
        synchronises the streams of AccumulateGrad nodes at the end of backward(), and these parameters have none now.)"""
        self._direct_groups.add(key)
        self._direct.update(self._index[id(p)] for p in params)
        if lane is not None:
            # the producer's launches are still being issued by a library thread (encoder_fn.lane_wait): the event is
            # recorded on its stream by gather_grads(), after the lane went idle
            self._pending.append((lane, torch.cuda.current_stream()))
            return
        ev = torch.cuda.Event()
        ev.record()
        self._pending.append(ev)

    def gather_grads(self):
        """All .grad tensors -> their slots in flat_g with one multi-tensor copy (the padding between slots stays zero;
        a missing grad counts as zero); slots already filled through the sink protocol are left alone."""
        if self._pending:
            cur = torch.cuda.current_stream()
            for ev in self._pending:
                if isinstance(ev, tuple):
                    from .sparse.encoder_fn import lane_wait
                    lane_wait(ev[0])
                    if ev[1] != cur:
                        cur.wait_stream(ev[1])
                else:
                    cur.wait_event(ev)
            self._pending.clear()
        key = frozenset(self._direct_groups)
        todo = self._gather_cache.get(key)
        if todo is None:
            todo = [i for i in range(len(self.params)) if i not in self._direct]
            self._gather_cache[key] = todo
        self._todo_last = todo
        slots, grads = [], []
        for i in todo:
            g = self.params[i].grad
            slots.append(self._slots[i])
            grads.append(g if g is not None else torch.zeros_like(self.params[i]))
        if slots:
            torch._foreach_copy_(slots, grads)
        for i in self._direct:           # a producer delivered AND autograd accumulated (second backward): add it
            g = self.params[i].grad
            if g is not None:
                self._slots[i].add_(g)
                self.params[i].grad = None

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
