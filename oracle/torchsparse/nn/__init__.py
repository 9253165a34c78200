from .modules import Conv3d, BatchNorm, ReLU, GlobalMaxPooling  # noqa: F401
from . import functional  # noqa: F401
