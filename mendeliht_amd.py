"""Import alias: the product package lives in the directory `mendeliht.jl_amd/`, whose name
is not a valid Python identifier; `import mendeliht_amd` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mendeliht.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "mendeliht_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mendeliht_amd"] = _mod
_spec.loader.exec_module(_mod)
