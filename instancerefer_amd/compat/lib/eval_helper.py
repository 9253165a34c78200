"""`lib.eval_helper` of the reference (lib/solver.py:13-14, scripts/eval.py:20-21) -> the irx drop-in."""
from instancerefer_amd.eval_helper import *  # noqa: F401,F403
from instancerefer_amd import eval_helper as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
