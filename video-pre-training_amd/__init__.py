"""vpt_amd -- MI355X-native implementation of the VPT (openai/Video-Pre-Training) policy hot path.

The importable name is `vpt_amd` (see /vpt_amd.py at the repo root: the directory name carries a hyphen).
Product code only: nothing here imports `oracle/`.
"""
from . import _native  # noqa: F401


def native_library_path():
    return _native.lib_path()
