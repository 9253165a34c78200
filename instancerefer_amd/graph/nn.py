"""`knn` and `MessagePassing` with the call signatures of torch_geometric.nn (reference models/basic_blocks.py:7):

    row, col = knn(x, y, k, batch_x, batch_y)                       # basic_blocks.py:120
    self.propagate(edge_index, x=(xs, xq), pos=(ps, pq))            # basic_blocks.py:125, aggr='max'

Both run on the HIP library (no CPU path: tensors must live on a HIP device). This is the general, edge-list shaped
surface for reference-style modules; the drop-in DynamicEdgeConv (instancerefer_amd/basic_blocks.py) calls
irx_knn_batched directly on a fixed (query, k) grid and never materialises the data-dependent edge list."""
import inspect

import torch
import torch.nn as nn

from ..sparse import functional as F_


def _offsets(batch, nb):
    """Start row of every batch item in a sorted batch-index vector -> int32 [nb + 1] (device, no sync)."""
    edges = torch.arange(nb + 1, device=batch.device, dtype=batch.dtype)
    return torch.searchsorted(batch.contiguous(), edges).to(torch.int32)


def knn(x, y, k, batch_x=None, batch_y=None, cosine=False, num_workers=1, batch_size=None):
    """torch_cluster.knn: for every row of `y` (queries) its k nearest rows of `x` (support) among the rows with the
    same batch index, Euclidean. -> LongTensor [2, E]: row 0 = query index, row 1 = support index, queries in order,
    neighbours by ascending distance; a batch item with fewer than k support rows yields fewer edges.
    `batch_x` / `batch_y` must be sorted ascending (upstream requires the same)."""
    if cosine:
        raise NotImplementedError("irx knn: cosine distance is not on the InstanceRefer path")
    if x.shape[1] != 3 or y.shape[1] != 3:
        raise NotImplementedError("irx knn: 3-D points only (the instance graph lives on box centres)")
    nq = y.shape[0]
    if nq == 0 or x.shape[0] == 0:
        return torch.zeros((2, 0), dtype=torch.long, device=x.device)
    if batch_x is None:
        batch_x = torch.zeros(x.shape[0], dtype=torch.long, device=x.device)
    if batch_y is None:
        batch_y = torch.zeros(nq, dtype=torch.long, device=x.device)
    # batch_size (torch_cluster >= 1.6.1 has the same keyword) saves the host round trip of deducing it; the [2, E] result has
    # a data-dependent length either way (fewer than k neighbours in small scenes), which is one sync this compat surface
    # cannot avoid — the drop-in RelationModule does not go through here (fused k_edgeconv_* on the (nq, k) table)
    nb = int(batch_size) if batch_size is not None else int(torch.maximum(batch_x.max(), batch_y.max())) + 1
    nbr = F_.knn_batched(x, _offsets(batch_x, nb), y, batch_y.to(torch.int32).contiguous(), int(k))      # (nq, k), -1 = none
    valid = nbr >= 0
    row = torch.arange(nq, device=x.device).unsqueeze(1).expand_as(nbr)[valid]
    return torch.stack([row, nbr[valid].long()], 0)


class MessagePassing(nn.Module):
    """torch_geometric.nn.MessagePassing, flow='source_to_target', node_dim=0: `propagate(edge_index, **kwargs)` gathers
    `<name>_j = kwargs[name][0][edge_index[0]]` and `<name>_i = kwargs[name][1][edge_index[1]]` (a single tensor serves
    both roles) for the parameters `message()` declares, then aggregates the messages per target row
    (edge_index[1]): 'max' through irx_segment_max (rows without an edge get 0, torch_scatter's fill; the gradient goes to
    the arg-max edge), 'add' / 'mean' through index_add."""

    def __init__(self, aggr='add', flow='source_to_target', node_dim=0):
        super().__init__()
        if flow != 'source_to_target' or node_dim != 0:
            raise NotImplementedError("irx MessagePassing: flow='source_to_target', node_dim=0 only")
        if aggr not in ('max', 'add', 'mean'):
            raise NotImplementedError("irx MessagePassing: aggr %r" % (aggr,))
        self.aggr = aggr
        self._msg_params = [p for p in inspect.signature(self.message).parameters]

    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out

    def propagate(self, edge_index, size=None, **kwargs):
        j, i = edge_index[0], edge_index[1]
        n_dst, args = (size[1] if size is not None else None), {}
        for name in self._msg_params:
            base, _, role = name.rpartition('_')
            if role not in ('i', 'j') or base not in kwargs:
                raise TypeError("message() parameter %r has no matching propagate() argument" % name)
            data = kwargs[base]
            src, dst = data if isinstance(data, (tuple, list)) else (data, data)
            if n_dst is None:
                n_dst = dst.shape[0]
            args[name] = src.index_select(0, j) if role == 'j' else dst.index_select(0, i)
        msg = self.message(**args)
        return self.update(self.aggregate(msg, i, n_dst))

    def aggregate(self, msg, index, n_dst):
        if self.aggr in ('add', 'mean'):
            out = msg.new_zeros((n_dst,) + tuple(msg.shape[1:])).index_add_(0, index, msg)
            if self.aggr == 'mean':
                cnt = torch.bincount(index, minlength=n_dst).clamp(min=1).to(msg.dtype)
                out = out / cnt.view(-1, *([1] * (msg.dim() - 1)))
            return out
        if msg.dim() != 2:
            raise NotImplementedError("irx MessagePassing(max): messages must be (E, C)")
        if index.numel() > 1 and not bool((index[1:] >= index[:-1]).all()):
            order = torch.sort(index, stable=True)[1]          # segments must be contiguous rows
            msg, index = msg.index_select(0, order), index.index_select(0, order)
        return F_.segment_max(msg, _offsets(index, n_dst), n_dst)
