"""`torch_geometric` as the reference imports it (models/basic_blocks.py:7) -> instancerefer_amd.graph."""
from instancerefer_amd.graph import nn  # noqa: F401
__version__ = "1.6.1+irx"
