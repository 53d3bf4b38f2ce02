"""Import shim: the product package lives in ``wavesandeigenvalues.jl_amd/`` (a directory name Python cannot
import directly because of the dot).  ``import wae_amd`` loads that directory as the package ``wae_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wavesandeigenvalues.jl_amd")
_spec = importlib.util.spec_from_file_location("wae_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["wae_amd"] = _mod
_spec.loader.exec_module(_mod)
