"""Sync BatchNorm for the data-parallel path (optional): BatchNorm statistics over the rows of ALL ranks.

The reference trains on one GPU (lib/solver.py:200-205), so there is nothing to mirror; the contract is
torch.nn.SyncBatchNorm's: `convert_sync_batchnorm(model)` once, before the optimizer is built, every rank runs every
BatchNorm layer in every step. All BatchNorm layers of the model go through the irx kernels then (the sparse encoders'
layers, the scene head's BatchNorm2d rows, and the heads' BatchNorm1d): sparse/functional.BatchNormActFn folds this rank's
float64 sums over the default process group (one small all-reduce per layer and direction), include/irx.h "Sync BatchNorm".
The encoders run layer by layer in this mode (sparse/encoder_fn.can_fuse): the one-call executor has no collective inside.
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
        return super().forward(x)


def convert_sync_batchnorm(module):
    """Marks every BatchNorm layer of `module` for cross-rank statistics (in place; returns `module`). Without an
    initialised process group of more than one rank the layers behave exactly as before."""
    for m in module.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            m._irx_sync = True
            if type(m) is nn.BatchNorm1d:
                m.__class__ = SyncRowsBatchNorm1d
    return module
