"""Import alias for the package directory ``linearalgebrampi.jl_amd/`` (its name contains a dot,
which the ``import`` statement cannot spell): ``import hpcla_amd as hp``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "linearalgebrampi.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "hpcla_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["hpcla_amd"] = _mod
_spec.loader.exec_module(_mod)
