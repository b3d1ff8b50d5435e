"""Import shim: the package directory is named `leastsquaresoptim.jl_amd` (not a valid Python
identifier), so `import lsq_amd` loads it under this alias."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "leastsquaresoptim.jl_amd")
_spec = importlib.util.spec_from_file_location("lsq_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["lsq_amd"] = _mod
_spec.loader.exec_module(_mod)
