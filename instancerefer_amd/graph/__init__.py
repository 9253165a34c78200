"""torch_geometric-shaped operator surface of the irx library: the two PyG entry points the reference's
models/basic_blocks.py:7,98-133 reaches — `knn` (torch_cluster) and `MessagePassing(aggr='max').propagate`
(torch_scatter max) — over libirx.so (irx_knn_batched, irx_segment_max). `instancerefer_amd/compat/torch_geometric`
aliases this package under the upstream name, so reference-style code keeps importing `torch_geometric.nn`."""
from . import nn  # noqa: F401
from .nn import MessagePassing, knn  # noqa: F401
