"""Import shim: loads the hyphenated package directory `agilex-ntt_amd/` as `agilex_ntt_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "agilex-ntt_amd")
_spec = importlib.util.spec_from_file_location(
    "agilex_ntt_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["agilex_ntt_amd"] = _mod
_spec.loader.exec_module(_mod)
