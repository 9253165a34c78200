"""Flat-buffer training state — optimizer AND gradient reducer of the data-parallel path.

Every parameter is a view into ONE fp32 buffer; gradients are gathered into a second flat buffer (the sparse encoders
write theirs straight into it: "gradient sink"), all-reduced over RCCL/xGMI (torch.distributed backend "nccl" is RCCL
on ROCm; "gloo" in the CPU tests) and applied by ONE fused Adam launch (csrc/irx_optim.hip). Semantics =
torch.optim.Adam(lr, betas, eps, weight_decay), the optimizer of the reference (scripts/train.py:121), including its
rule that a parameter without a gradient is left alone (no weight decay, no moment decay, no step count).

The reference is single-GPU (lib/solver.py:200-205 is a plain backward(); step()), so the multi-rank behaviour is this
build's own: one process per GPU, scenes sharded by rank (`shard_range`), parameters and buffers broadcast from rank 0
at construction, gradients summed over ranks (ranks whose shard produced no gradient for a parameter contribute
zeros) and divided by the world size inside the Adam kernel. flat_g is cut into a STATIC list of segments — one per
asynchronously issued sparse encoder of `module` (80 % of the 32 MB), plus what lies between / behind them — and every
rank all-reduces every segment in the same order each step; an encoder segment whose gradients arrived through the sink
is reduced as soon as its backward pass is enqueued, on the encoder's own stream, while the rest of the backward still
runs (`overlap=True`); on a rank where that encoder did not run this step the same collective is issued after the
gather instead, so the sequence of collectives never depends on the data.

Buffer management, gather, all-reduce, broadcast and the state dict work on any device (the world-size-2 gloo test
runs them on CPU tensors); `step()` is the HIP kernel and raises without a HIP device — there is no CPU optimizer."""
import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def shard_range(n_items, rank, world_size):
    """Contiguous shard [lo, hi) of n_items for `rank` (DistributedSampler-like, equal sizes required
    for loss parity with a single-process global batch: each rank divides by its local batch size)."""
    per = n_items // world_size
    return rank * per, (rank + 1) * per


def broadcast_buffers(module, src=0):
    """BatchNorm running statistics (and any other buffer) of rank `src` to every rank, one flat collective."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    bufs = [b for b in module.buffers() if b.numel()]
    if not bufs:
        return
    by_dtype = {}
    for b in bufs:
        by_dtype.setdefault(b.dtype, []).append(b)
    for group in by_dtype.values():
        flat = torch.cat([b.detach().reshape(-1) for b in group])
        dist.broadcast(flat, src)
        off = 0
        with torch.no_grad():
            for b in group:
                b.copy_(flat[off:off + b.numel()].view_as(b))
                off += b.numel()


# Liveness of an optimizer's native gradient sink: a cell in the optimizer's own record tensor (row 256), 1 while this optimizer owns
# its parameters' slots, 0 once another FlatAdam re-homed them. C++ nodes capture the cell's address at forward time (the record tensor
# travels with the node, so the address stays valid) and deliver into the slots at backward time only while it still reads 1 — a node
# whose optimizer was replaced between its forward and its backward returns ordinary gradient tensors. (Round 5 used ONE process-wide
# generation counter: constructing a second optimizer — another model, a bench leg, a test — silently switched the first one's nodes
# to the slow path for the rest of the run: ADVICE r5.)
_SINK_LOCK = __import__("threading").Lock()


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, world_size=None, module=None,
                 broadcast=True, overlap=True):
        """params: iterable of parameters. module (optional): its buffers are broadcast from rank 0 together with the
        parameters when world_size > 1 (broadcast=False: the caller guarantees identical replicas)."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.device = dev
        # every parameter starts on a 64-byte boundary of the flat buffer (the conv kernels need 16-byte aligned
        # weight pointers for their 16 B/lane loads); the padding elements stay zero in all four buffers
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 15) // 16 * 16
        n = off
        self.n = n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        # flat_g carries one activity flag per parameter behind the gradients, so that the flags travel inside the
        # gradient all-reduce (no extra collective): flag > 0 after the reduce <=> some rank produced a gradient
        self.n_total = n + (len(self.params) + 15) // 16 * 16
        self.flat_g = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self._flags = self.flat_g[n:n + len(self.params)]
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
            prev = getattr(p, "_irx_sink", None)
            if prev is not None and prev[0] is not self:
                prev[0]._native_rec[256, 0] = 0              # that optimizer's slots are no longer this parameter's home: retire its sink
            p._irx_sink = (self, slot)
        # ... and the C++ autograd nodes (csrc/torch_nodes.cpp): they get the slot ADDRESSES at forward time and raise one host
        # flag per producer when their backward wrote them (no interpreter on that path); gather_grads() folds the flags in
        # one record per producer: [delivered flag, HIP stream the backward ran on]; a torch tensor so that the nodes can keep it
        # (and flat_g) alive for as long as their graph exists
        self._native_rec_t = torch.zeros((257, 2), dtype=torch.int64)          # rows 0..255: producers; row 256: the liveness cell
        self._native_rec = self._native_rec_t.numpy()
        with _SINK_LOCK:
            self._native_rec[256, 0] = 1
            self._gen = 1
        self._native = {}               # producer key -> (record index, parameter indices, (slot addresses + record), keep-alive)
        self._direct = set()            # parameter indices whose slot already holds this step's gradient
        self._direct_groups = set()     # producer keys that delivered since the last zero_grad()
        self._gather_cache = {}
        self._pending = []              # sink deliveries not yet waited for: (event | (lane, stream), parameter indices)
        self._todo_last = list(range(len(self.params)))
        self._works = []                # async collectives in flight
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.initial_lr = lr                          # undecayed rate (lr schedules change self.lr only; state_dict keeps both)
        self.steps = [0] * len(self.params)           # torch.optim.Adam keeps `step` per parameter
        self._inactive = frozenset()                  # parameters without a gradient this step
        self._runs_cache = {}
        self.world_size = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.overlap = bool(overlap)
        # static segments of flat_g: [(lo, hi)] of the early groups (parameter runs of module's lane-issued encoders, in
        # module order), then the gaps; identical on every rank because it only depends on the model structure
        self._groups = []               # (first parameter index, lo, hi), in lane order (lane 0 = the scene encoder,
        lanes = []                      # which runs on every rank every step and sits on the critical path: first)
        if module is not None and self.overlap:
            for m in module.modules():
                if m.__dict__.get('_irx_lane') is not None:
                    idx = sorted(self._index[id(p)] for p in m.parameters() if id(p) in self._index)
                    if idx and idx == list(range(idx[0], idx[-1] + 1)):
                        hi = self.offsets[idx[-1] + 1] if idx[-1] + 1 < len(self.offsets) else n
                        lanes.append((m.__dict__['_irx_lane'], idx[0], self.offsets[idx[0]], hi))
        self._groups = [g[1:] for g in sorted(lanes)]
        self._gaps, lo = [], 0
        for _, a, b in sorted(self._groups, key=lambda g: g[1]):
            if a > lo:
                self._gaps.append((lo, a))
            lo = b
        self._gaps.append((lo, self.n_total))            # ... and the activity flags behind the gradients
        self._reduced = set()           # groups already all-reduced this step
        if broadcast and self.world_size > 1 and dist.is_initialized():
            dist.broadcast(self.flat_p, 0)            # replicas start from rank 0's weights whatever the local seeds
            if module is not None:
                broadcast_buffers(module, 0)

    @property
    def step_count(self):
        return max(self.steps) if self.steps else 0

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
        self._reduced.clear()
        if self._native:
            self._native_rec[:len(self._native)] = 0

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

    def native_sink(self, key, params):
        """-> (sink vector, keep-alive tensors) for a C++ node (csrc/torch_nodes.cpp) that computes all gradients of `params`,
        or None when a parameter is not ours. Sink vector = the slot addresses followed by [record address, generation, address
        of the generation counter]; the node writes the slots in its backward, sets record[0] = 1 and record[1] = the stream it
        ran on — if the generation still matches (this optimizer has not been replaced since the forward) and the record is
        still 0 (a second backward before zero_grad() returns ordinary gradients, which gather_grads() adds). The keep-alive
        list (flat_g, the record tensor) travels with the node, so its addresses stay valid whatever happens to this object.
        gather_grads() makes its stream wait for a delivering stream that is not its own. Thread-safe (the language module's
        helper thread calls it)."""
        ent = self._native.get(key)
        if ent is None:
            with _SINK_LOCK:
                ent = self._native.get(key)
                if ent is None:
                    try:
                        idx = [self._index[id(p)] for p in params]
                    except KeyError:
                        return None
                    j = len(self._native)
                    if j >= 256:
                        return None
                    vec = [self._slots[i].data_ptr() for i in idx] + [self._native_rec.ctypes.data + 16 * j, self._gen,
                                                                      self._native_rec.ctypes.data + 16 * 256]
                    ent = (j, idx, vec, [self.flat_g, self._native_rec_t])
                    self._native[key] = ent
        return ent[2], ent[3]

    def native_delivered(self):
        """-> (producers, parameters) whose C++ nodes have delivered since the last zero_grad()"""
        done = [ent for ent in self._native.values() if self._native_rec[ent[0], 0]]
        return len(done), sum(len(ent[1]) for ent in done)

    def sink_delivered_inline(self, key, params):
        """sink_delivered for a producer that enqueued its kernels on the CURRENT stream and whose stream is the one
        gather_grads() / the optimizer will run on (the heads' fused MLP nodes: autograd replays them on the main stream): the
        stream order already guarantees the slots are written before they are read — no event, nothing pending."""
        self._direct_groups.add(key)
        self._direct.update(self._index[id(p)] for p in params)

    def sink_delivered(self, key, params, lane=None):
        """Called by the producer right after it enqueued the kernels that write the slots (on ITS current stream, which
        for the scene encoder is not the optimizer's): an event makes gather_grads() wait for them. (Autograd only
        synchronises the streams of AccumulateGrad nodes at the end of backward(), and these parameters have none now.)"""
        self._direct_groups.add(key)
        idx = [self._index[id(p)] for p in params]
        self._direct.update(idx)
        if lane is not None:
            # the producer's launches are still being issued by a library thread (encoder_fn.lane_wait): the event is
            # recorded on its stream by gather_grads(), after the lane went idle
            self._pending.append(((lane, torch.cuda.current_stream()), idx))
            return
        ev = torch.cuda.Event()
        ev.record()
        self._pending.append(((ev, torch.cuda.current_stream()), idx))

    def _group_of(self, idx):
        """Position in self._groups of the static segment whose first parameter is min(idx), or None."""
        first = min(idx)
        for g, (i0, _, _) in enumerate(self._groups):
            if i0 == first:
                return g
        return None

    def _multi_rank(self):
        # _force_collectives (tests): a one-rank process group still issues every collective, so that the RCCL stream
        # ordering of the product path can be exercised on a single GPU
        return dist.is_initialized() and (self.world_size > 1 or getattr(self, "_force_collectives", False))

    def gather_grads(self):
        """All .grad tensors -> their slots in flat_g with one multi-tensor copy (the padding between slots stays zero;
        a missing grad leaves a zero slot and marks the parameter inactive for this step); slots already filled through
        the sink protocol are left alone. In overlap mode with world_size > 1, each delivered group that is one
        contiguous range of flat_g is all-reduced right here, on the producer's stream, before the copy of the remaining
        gradients is even enqueued."""
        if self._pending:
            cur = torch.cuda.current_stream()
            if self._groups:             # segment order, so that a later segment can go early behind an earlier one
                self._pending.sort(key=lambda e: (lambda g: len(self._groups) if g is None else g)(self._group_of(e[1])))
            for (ev, stream), idx in self._pending:
                if not isinstance(ev, torch.cuda.Event):
                    from .sparse.encoder_fn import lane_wait
                    lane_wait(ev)
                g = self._group_of(idx) if self._multi_rank() else None
                # early reduction only in segment order: an earlier segment that did not deliver on THIS rank is reduced
                # after the gather, and the collectives of all ranks must line up
                # (and never for a group that autograd ALSO accumulated into — a second backward before the step,
                # gradient accumulation: its .grad is added to the slots below, which must happen before the collective)
                if g is not None and g not in self._reduced and all(h in self._reduced for h in range(g)) \
                        and not any(self.params[i].grad is not None for i in idx):
                    _, lo, hi = self._groups[g]
                    with torch.cuda.stream(stream):          # ordered behind the producer's kernels, nothing else
                        if isinstance(ev, torch.cuda.Event):
                            stream.wait_event(ev)
                        w = dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM, async_op=True)
                    self._works.append(w)
                    self._reduced.add(g)
                if isinstance(ev, torch.cuda.Event):
                    cur.wait_event(ev)
                elif stream != cur:
                    cur.wait_stream(stream)
            self._pending.clear()
        if self._native:
            rec = self._native_rec
            cur_ptr = None
            for k, (j, idx, _, _) in self._native.items():
                if rec[j, 0] and k not in self._direct_groups:
                    self._direct_groups.add(k)
                    self._direct.update(idx)
                    if self.device.type == "cuda":
                        # the node ran on the stream autograd replayed it on; backward() has returned, so every launch is enqueued
                        # there: if that is not the stream the gather / optimizer run on, order them behind it
                        if cur_ptr is None:
                            cur = torch.cuda.current_stream(self.device)
                            cur_ptr = int(cur.cuda_stream)
                        sp = int(rec[j, 1])
                        if sp != cur_ptr:
                            cur.wait_stream(torch.cuda.ExternalStream(sp, device=self.device))
        key = frozenset(self._direct_groups)
        todo = self._gather_cache.get(key)
        if todo is None:
            todo = [i for i in range(len(self.params)) if i not in self._direct]
            self._gather_cache[key] = todo
        self._todo_last = todo
        slots, grads, zero, inactive = [], [], [], []
        for i in todo:
            g = self.params[i].grad
            if g is None:
                zero.append(self._slots[i])
                inactive.append(i)
            else:
                slots.append(self._slots[i])
                grads.append(g)
        if slots:
            torch._foreach_copy_(slots, grads)
        if zero:
            torch._foreach_zero_(zero)
        self._inactive = frozenset(inactive)
        for i in self._direct:           # a producer delivered AND autograd accumulated (second backward): add it
            g = self.params[i].grad
            if g is not None:
                self._slots[i].add_(g)
                self.params[i].grad = None

    def all_reduce(self):
        """Sum flat_g over the ranks (the division by world_size happens in the Adam kernel): the static segments in order,
        skipping those reduced early; the activity flags travel with the gradients so that every rank skips the same
        parameters (a parameter is inactive only when NO rank produced a gradient for it)."""
        if not self._multi_rank():
            return
        self._flags.fill_(1.0)
        if self._inactive:
            self._flags[torch.as_tensor(sorted(self._inactive), device=self.device)] = 0.0
        rest = [(a, b) for g, (_, a, b) in enumerate(self._groups) if g not in self._reduced] + self._gaps
        for a, b in rest:
            self._works.append(dist.all_reduce(self.flat_g[a:b], op=dist.ReduceOp.SUM, async_op=True))
        for w in self._works:
            w.wait()                     # NCCL: the current stream waits for the collective; gloo: the host does
        self._works = []
        self._reduced.update(range(len(self._groups)))
        if self._inactive:
            # A rank with gradients for everything knows every flag is > 0 and reads nothing back (the common case);
            # only a rank that skipped something has to learn whether another rank did not (one small D2H, rare)
            self._inactive = frozenset(i for i, f in enumerate(self._flags.tolist()) if f == 0.0)

    def _runs(self):
        """Contiguous runs of the flat buffer whose parameters are active and share one step count:
        [(lo, hi, step)], one fused launch each (one run in all but the steps after a skipped gradient)."""
        for i in range(len(self.params)):
            if i not in self._inactive:
                self.steps[i] += 1
        key = (self._inactive, tuple(self.steps)) if (self._inactive or len(set(self.steps)) > 1) else None
        if key is None:
            return [(0, self.n, self.steps[0])]
        runs = self._runs_cache.get(key)
        if runs is None:
            runs = []
            for i in range(len(self.params)):
                if i in self._inactive:
                    continue
                lo = self.offsets[i]
                hi = self.offsets[i + 1] if i + 1 < len(self.offsets) else self.n
                if runs and runs[-1][1] == lo and runs[-1][2] == self.steps[i]:
                    runs[-1] = (runs[-1][0], hi, self.steps[i])
                else:
                    runs.append((lo, hi, self.steps[i]))
            if len(self._runs_cache) > 64:
                self._runs_cache.clear()
            self._runs_cache[key] = runs
        return runs

    def step(self):
        for lo, hi, step in self._runs():
            _lib.call("irx_adam_step", _lib.ptr(self.flat_p[lo:hi]), _lib.ptr(self.flat_g[lo:hi]),
                      _lib.ptr(self.exp_avg[lo:hi]), _lib.ptr(self.exp_avg_sq[lo:hi]), hi - lo, float(self.lr),
                      float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay), step,
                      1.0 / self.world_size, _lib.stream_ptr())

    def backward_step(self):
        """After loss.backward(): gather -> all-reduce -> Adam."""
        self.gather_grads()
        self.all_reduce()
        self.step()

    # ---- torch.optim.Adam-compatible state (reference scripts/train.py:114-119 calls optimizer.load_state_dict) ----
    def state_dict(self):
        """The layout torch.optim.Adam.state_dict() produces for the same parameter list: per-parameter `step`,
        `exp_avg`, `exp_avg_sq` (own storage, parameter-shaped) and ONE param group."""
        state = {}
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            if self.steps[i] == 0:
                continue                  # torch creates a parameter's state at its first step
            state[i] = {"step": torch.tensor(float(self.steps[i])),
                        "exp_avg": self.exp_avg[off:off + p.numel()].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + p.numel()].view_as(p).clone()}
        # `initial_lr` is what torch's lr schedulers add to a param group (the reference's MultiStepLR does,
        # lib/solver.py:119-124): the undecayed rate, so a resume can continue the schedule instead of decaying twice
        group = {"lr": self.lr, "initial_lr": float(self.initial_lr if self.initial_lr is not None else self.lr), "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": False, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts what torch.optim.Adam.state_dict() (any torch since 1.6: `step` int or tensor) or state_dict() above
        wrote for the same parameter list."""
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if len(ids) != len(self.params):
            raise ValueError("optimizer state has %d parameters, this model %d" % (len(ids), len(self.params)))
        g0 = groups[0]
        self.lr, self.betas = float(g0["lr"]), tuple(float(b) for b in g0["betas"])
        self.initial_lr = float(g0["initial_lr"]) if "initial_lr" in g0 else None
        self.eps, self.weight_decay = float(g0["eps"]), float(g0["weight_decay"])
        if g0.get("amsgrad"):
            raise ValueError("amsgrad state is not supported")
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.steps = [0] * len(self.params)
        with torch.no_grad():
            for pos, pid in enumerate(ids):
                st = sd["state"].get(pid)
                if st is None:
                    continue
                p, off = self.params[pos], self.offsets[pos]
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError("optimizer state %d has shape %s, parameter %s" % (pid, tuple(st["exp_avg"].shape), tuple(p.shape)))
                self.steps[pos] = int(float(st["step"]))
                self.exp_avg[off:off + p.numel()].view_as(p).copy_(st["exp_avg"])
                self.exp_avg_sq[off:off + p.numel()].view_as(p).copy_(st["exp_avg_sq"])
