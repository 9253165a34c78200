"""`torch_geometric.nn` (MessagePassing, knn) -> instancerefer_amd.graph.nn."""
from instancerefer_amd.graph.nn import MessagePassing, knn  # noqa: F401
