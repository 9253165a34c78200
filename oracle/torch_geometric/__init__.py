"""CPU restatement of the two torch_geometric entry points the reference uses
(models/basic_blocks.py:7,98-133): `knn` (torch_cluster) and `MessagePassing(aggr='max')`
(torch_scatter max). Test oracle — see oracle/__init__.py."""
