"""`torchsparse` as the reference imports it (models/basic_blocks.py:4-6, lib/dataset.py:16-17) -> instancerefer_amd.sparse."""
from instancerefer_amd.sparse import SparseTensor, nn, utils  # noqa: F401
__version__ = "1.2.0+irx"
