"""Import shim: the package directory is named ``manual-whisper_amd`` (not a
valid Python identifier), so ``import manual_whisper_amd`` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "manual-whisper_amd")
_spec = importlib.util.spec_from_file_location(
    "manual_whisper_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["manual_whisper_amd"] = _mod
_spec.loader.exec_module(_mod)
