"""instancerefer_amd — MI355X-native hot path of InstanceRefer behind the reference's module API.

csrc/ + include/irx.h : hand-written HIP kernels for gfx950 behind a C-ABI (libirx.so)
sparse/               : torchsparse-shaped host surface (SparseTensor, nn, utils)
*_module.py, basic_blocks.py, instancerefer.py, loss_helper.py : drop-in mirrors of the reference's
                        models/* and lib/loss_helper.py interfaces
"""
__version__ = "0.1.0"
