"""Import shim: `import plnerf_amd` loads the package that lives in the directory
`pl-nerf_amd/` (a hyphen is not importable as a module name)."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pl-nerf_amd")
_spec = importlib.util.spec_from_file_location(
    "plnerf_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["plnerf_amd"] = _mod
_spec.loader.exec_module(_mod)
