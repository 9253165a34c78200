"""Sync BatchNorm for the data-parallel path (optional): BatchNorm statistics over the rows of ALL ranks.

The reference trains on one GPU (lib/solver.py:200-205), so there is nothing to mirror; the contract is
torch.nn.SyncBatchNorm's: `convert_sync_batchnorm(model)` once, before the optimizer is built, every rank runs every
BatchNorm layer in every step. All BatchNorm layers of the model go through the irx kernels then (the sparse encoders'
layers, the scene head's BatchNorm2d rows, and the heads' BatchNorm1d): sparse/functional.BatchNormActFn folds this rank's
float64 sums over the default process group (one small all-reduce per layer and direction), include/irx.h "Sync BatchNorm".
The two sparse encoders stay in the one-call executor (round 4: irx_encoder_forward_sync / _backward_sync — the library calls back
into torch.distributed between every layer's statistics and apply pass, sparse/encoder_fn.py; the pass then runs inline on the
calling thread instead of a library lane; IRX_SYNC_BN_EXECUTOR=0 restores the layer-by-layer path of rounds 2-3).
Every rank must run every BatchNorm layer in every step (a shard with no scene of >= 2 candidates skips the candidate
encoder and would leave the other ranks waiting in its collectives): Solver / bench.py use shards that all hold candidates.
A converted layer that cannot take the synchronised path raises in multi-rank training instead of using per-rank statistics.
"""
import torch
import torch.nn as nn


class SyncRowsBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d on (N, C) device tensors through the irx statistics / apply kernels (same parameters, buffers and
    state-dict keys); anything else falls back to nn.BatchNorm1d."""

    def forward(self, x):
        if x.dim() == 2 and x.is_cuda:
            from .basic_blocks import batchnorm_rows
            return batchnorm_rows(self, x)
        _refuse_unsynced(self)
        return super().forward(x)


def _multi_rank():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _refuse_unsynced(m):
    """torch.nn.SyncBatchNorm's contract: a converted layer never silently normalises with per-rank statistics."""
    if m.training and _multi_rank():
        raise RuntimeError("%s was converted by convert_sync_batchnorm but this input cannot take the synchronised path "
                           "(needs a 2-D device tensor)" % type(m).__name__)


def _guard_hook(m, args):
    # layers the irx kernels reach directly (sparse.nn.BatchNorm, batchnorm_rows callers) never run this torch forward
    if getattr(m, "_irx_sync", False) and not getattr(m, "_irx_sync_native", False):
        _refuse_unsynced(m)


def convert_sync_batchnorm(module):
    """Marks every BatchNorm layer of `module` for cross-rank statistics (in place; returns `module`). Without an
    initialised process group of more than one rank the layers behave exactly as before."""
    for m in module.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            m._irx_sync = True
            if type(m) is nn.BatchNorm1d:
                m.__class__ = SyncRowsBatchNorm1d
            elif type(m).forward is nn.modules.batchnorm._BatchNorm.forward and not getattr(m, "_irx_guarded", False):
                # e.g. BatchNorm2d: the model's own code feeds these layers to batchnorm_rows (scene head) and never calls
                # their torch forward; if someone does, in training with more than one rank, fail instead of desynchronising
                m.register_forward_pre_hook(_guard_hook)
                m._irx_guarded = True
    return module
