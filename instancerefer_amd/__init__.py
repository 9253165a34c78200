"""instancerefer_amd — MI355X-native hot path of InstanceRefer behind the reference's module API.

csrc/ + include/irx.h : hand-written HIP kernels for gfx950 behind a C-ABI (libirx.so)
sparse/               : torchsparse-shaped host surface (SparseTensor, nn, utils)
*_module.py, basic_blocks.py, instancerefer.py, loss_helper.py : drop-in mirrors of the reference's
                        models/* and lib/loss_helper.py interfaces
"""
__version__ = "0.1.0"

import os as _os

def configure_hw_queues(ranks_per_device=None, force=False):
    """Ask the HIP runtime for the number of hardware queues this package's stream layout wants (GPU_MAX_HW_QUEUES; the runtime's
    default is 4). NOT done at import: the variable is process-wide, only read when the runtime initialises, and the
    right value depends on how many processes share a device —
      one rank per GPU (production layout): 8. The bf16 training forward uses three streams beside the caller's preparation
        stream and RCCL adds its own; a stream that shares a hardware queue with another runs behind it (round 5: a fifth stream
        on four queues HALVED the step);
      several ranks on ONE GPU (test rigs, train + eval on one device): 2. Two processes x 8 queues oversubscribe the device's
        queue slots and every launch waits for a queue switch (measured: 31.8 s per step against 20 ms with 2).
    ranks_per_device: None = LOCAL_WORLD_SIZE / visible devices, rounded up. An explicit GPU_MAX_HW_QUEUES in the environment
    always wins unless force=True. Returns the value in force, or None when the runtime was already initialised (too late: a
    warning says so). Callers: bench.py, Solver (before the first GPU call), tests/test_multirank_gpu.py."""
    import warnings
    if not force and "GPU_MAX_HW_QUEUES" in _os.environ:
        return int(_os.environ["GPU_MAX_HW_QUEUES"])
    try:
        import torch
        if torch.cuda.is_initialized():
            warnings.warn("instancerefer_amd.configure_hw_queues: the HIP runtime is already initialised; GPU_MAX_HW_QUEUES "
                          "keeps its start-up value (call this before the first GPU operation)", RuntimeWarning, stacklevel=2)
            return None
        ndev = torch.cuda.device_count()
    except Exception:                                          # pragma: no cover - plumbing only
        ndev = 0
    if ranks_per_device is None:
        local = int(_os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
        ranks_per_device = -(-local // ndev) if ndev > 0 else 1
    if ranks_per_device > 1:
        warnings.warn("instancerefer_amd: %d ranks share one GPU; asking for 2 hardware queues per process (8 per process "
                      "oversubscribes the device's queue slots: ~1000x slower launches)" % ranks_per_device, RuntimeWarning,
                      stacklevel=2)
    val = 8 if ranks_per_device <= 1 else 2
    _os.environ["GPU_MAX_HW_QUEUES"] = str(val)
    return val



# Host-side dispatch cost of the small dense GEMMs of the language / matching heads (M <= a few hundred rows), measured on
# MI355X with PyTorch 2.10: hipBLASLt 18 us per mm / 21 us per nn.Linear, rocBLAS 7 / 16 us (tools/micro/gemm_host.py).
# With ~70 such calls per training step and the step host-bound, that is ~0.6 ms of 12.  The kernels themselves are
# microseconds either way.  IRX_KEEP_BLAS=1 leaves PyTorch's defaults untouched.
if _os.environ.get("IRX_KEEP_BLAS") != "1":
    _os.environ.setdefault("DISABLE_ADDMM_CUDA_LT", "1")      # read once by ATen at the first addmm
    try:
        import torch as _torch
        if _torch.cuda.is_available():
            import warnings as _warnings
            with _warnings.catch_warnings():
                _warnings.simplefilter("ignore")
                _torch.backends.cuda.preferred_blas_library("cublas")     # "cublas" is rocBLAS on ROCm
    except Exception:                                          # pragma: no cover - plumbing only
        pass


_DTYPE_MODES = {"fp32": 0, "f32": 0, "bf16_operands": 1, "bf16op": 1, "bf16": 2}


def set_compute_dtype(name):
    """Compute dtype of the sparse encoders (irx_set_compute_dtype, include/irx.h):
      "fp32"          (default) exact fp32 MFMA; the 1e-4 parity gate;
      "bf16"          BASELINE configs[2]-[4]: bf16 operands with fp32 accumulation in the 32/64/128-channel convs AND bf16
                      storage of every activation / gradient tensor inside the encoder executor (half the HBM bytes);
                      BatchNorm statistics, accumulation, parameters and their gradients, encoder inputs / outputs fp32;
      "bf16_operands" bf16 operands only, every tensor fp32 in HBM (round 1's mode)."""
    from . import _lib
    if name not in _DTYPE_MODES:
        raise ValueError("compute dtype must be one of %s, got %r" % (sorted(set(_DTYPE_MODES)), name))
    from .sparse import encoder_fn
    for lane in range(2):                     # a queued pass reads the mode when its kernels are issued: drain first
        encoder_fn.lane_wait(lane)
    _lib.call("irx_set_compute_dtype", _DTYPE_MODES[name])


def get_compute_dtype():
    from . import _lib
    return ("fp32", "bf16_operands", "bf16")[_lib.load().irx_get_compute_dtype()]
