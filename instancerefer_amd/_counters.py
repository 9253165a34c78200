"""BatchNorm `num_batches_tracked` increments of one forward, applied in ONE multi-tensor launch.

Every train-mode BatchNorm bumps its int64 counter buffer per forward (nn.BatchNorm semantics; reference models use nn.BatchNorm1d /
2d and spnn.BatchNorm). Issued where they occur that is one tiny launch + ~30 us of host time per group — two encoders, two head nodes,
the relation head: 5 per step on a host-bound loop. InstanceRefer.forward opens a collector; producers hand their counter tensors to
bump(); close() applies them with a single torch._foreach_add_. Without an open collector bump() applies at once (modules used on
their own keep nn.BatchNorm's behaviour)."""
import threading

import torch

_lock = threading.Lock()
_open = None          # list collecting tensors, or None


def open_collector():
    global _open
    with _lock:
        _open = []


def bump(counters):
    """counters: iterable of int64 buffer tensors to increment by one"""
    counters = [c for c in counters if c is not None]
    if not counters:
        return
    with _lock:
        if _open is not None:
            _open.extend(counters)
            return
    with torch.no_grad():
        torch._foreach_add_(counters, 1)


def close():
    global _open
    with _lock:
        pending, _open = _open, None
    if pending:
        with torch.no_grad():
            torch._foreach_add_(pending, 1)
