"""`torchsparse.nn` (spnn.Conv3d / BatchNorm / ReLU / GlobalMaxPooling) -> instancerefer_amd.sparse.nn."""
from instancerefer_amd.sparse.nn import *  # noqa: F401,F403
from instancerefer_amd.sparse.nn import BatchNorm, Conv3d, GlobalMaxPooling, ReLU  # noqa: F401
