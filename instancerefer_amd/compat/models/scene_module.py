"""`models.scene_module` of the reference (scripts/train.py:18, models/instancerefer.py:20-34 plug-in names) -> the irx drop-in."""
from instancerefer_amd.scene_module import *  # noqa: F401,F403
from instancerefer_amd import scene_module as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
