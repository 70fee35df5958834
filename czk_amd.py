"""Loader shim: makes the package directory `collaborative-zksnark_amd/` (not a valid Python identifier)
importable as `czk_amd` from the repo root."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "collaborative-zksnark_amd")
_spec = importlib.util.spec_from_file_location("czk_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["czk_amd"] = _mod
_spec.loader.exec_module(_mod)
