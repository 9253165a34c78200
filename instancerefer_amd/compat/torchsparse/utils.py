"""`torchsparse.utils` (sparse_quantize, sparse_collate_fn, sparse_collate_tensors) -> instancerefer_amd.sparse.utils."""
from instancerefer_amd.sparse.utils import (sparse_collate, sparse_collate_fn, sparse_collate_tensors,  # noqa: F401
                                            sparse_quantize)
