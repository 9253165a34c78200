"""CPU restatement of `torchsparse` (v1.1/v1.2 API) — test oracle, see oracle/__init__.py.
Stands in for the absent package when the reference's modules are imported in the build container
(reference imports: models/basic_blocks.py:4-6, models/attribute_module.py:4-8,
models/scene_module.py:5, lib/dataset.py:16-17)."""
from .tensor import SparseTensor  # noqa: F401
from . import nn  # noqa: F401
from . import utils  # noqa: F401

__version__ = "1.2.0-oracle"
