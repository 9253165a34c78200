import inspect

import numpy as np
import torch
import torch.nn as nn


def knn(x, y, k, batch_x=None, batch_y=None):
    """torch_cluster.knn: for each row of y the k nearest rows of x with the same batch id (Euclidean).
    Returns [2, E]: row 0 = y (query) index, row 1 = x (support) index; fewer than k edges when the batch
    item has fewer than k support rows. Distances in float32 as sum of squared differences; ties resolve
    to the lower support index (upstream leaves tie order unspecified)."""
    xs = x.detach().cpu().numpy().astype(np.float32)
    ys = y.detach().cpu().numpy().astype(np.float32)
    bx = np.zeros(len(xs), np.int64) if batch_x is None else batch_x.cpu().numpy()
    by = np.zeros(len(ys), np.int64) if batch_y is None else batch_y.cpu().numpy()
    rows, cols = [], []
    for qi in range(len(ys)):
        cand = np.nonzero(bx == by[qi])[0]
        if cand.size == 0:
            continue
        d = xs[cand] - ys[qi]
        d = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + (d[:, 2] * d[:, 2]).astype(np.float32) \
            if xs.shape[1] == 3 else (d * d).sum(1)
        order = np.lexsort((cand, d))[:k]
        rows += [qi] * len(order)
        cols += cand[order].tolist()
    return torch.tensor([rows, cols], dtype=torch.long)


class MessagePassing(nn.Module):
    """propagate(edge_index, x=(x_src, x_dst), pos=(p_src, p_dst)) with flow source_to_target:
    *_j = arg[0][edge_index[0]], *_i = arg[1][edge_index[1]]; messages are max-aggregated per target
    row (rows without edges get 0, torch_scatter's fill)."""

    def __init__(self, aggr='add', flow='source_to_target'):
        super().__init__()
        assert flow == 'source_to_target'
        self.aggr = aggr
        self._msg_args = [p for p in inspect.signature(self.message).parameters]

    def message(self, x_j):
        return x_j

    def propagate(self, edge_index, size=None, **kwargs):
        j, i = edge_index[0], edge_index[1]
        n_dst = None
        args = {}
        for name in self._msg_args:
            base, which = name.rsplit('_', 1)
            data = kwargs[base]
            if isinstance(data, (tuple, list)):
                src, dst = data
            else:
                src = dst = data
            if n_dst is None:
                n_dst = dst.shape[0]
            args[name] = src.index_select(0, j) if which == 'j' else dst.index_select(0, i)
        msg = self.message(**args)
        out = torch.zeros(n_dst, msg.shape[1], dtype=msg.dtype)
        idx = i.unsqueeze(1).expand_as(msg)
        if self.aggr == 'max':
            return out.scatter_reduce(0, idx, msg, reduce='amax', include_self=False)
        if self.aggr == 'add':
            return out.scatter_add(0, idx, msg)
        raise NotImplementedError(self.aggr)
