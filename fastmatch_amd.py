"""Import alias for the package directory ``fast-match_amd/`` (a hyphen is not a valid
identifier).  ``import fastmatch_amd`` returns that package; its submodules are then
importable as ``fastmatch_amd.fastmatch``, ``fastmatch_amd.cache`` ..."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fast-match_amd")
_spec = importlib.util.spec_from_file_location(
    "fastmatch_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["fastmatch_amd"] = _mod
_spec.loader.exec_module(_mod)
