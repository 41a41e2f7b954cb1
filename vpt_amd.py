"""Import shim: exposes the package directory `video-pre-training_amd/` (hyphenated, hence not a valid
identifier) under the importable name `vpt_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "video-pre-training_amd")
_spec = importlib.util.spec_from_file_location("vpt_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vpt_amd"] = _mod
_spec.loader.exec_module(_mod)
