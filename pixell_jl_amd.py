"""Import shim: makes the package directory `pixell.jl_amd/` (whose name is not a valid Python
identifier) importable as `pixell_jl_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pixell.jl_amd")
_spec = importlib.util.spec_from_file_location("pixell_jl_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["pixell_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
