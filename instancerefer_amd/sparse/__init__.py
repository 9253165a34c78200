"""torchsparse-shaped operator surface of the irx library (SparseTensor, nn, utils)."""
from .tensor import SparseTensor  # noqa: F401
from . import nn  # noqa: F401
from . import utils  # noqa: F401
