"""Import shim: the package directory is named `osmo-tetra_amd/` (not a valid Python
identifier), so `import osmo_tetra_amd` resolves here and loads it under this name."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "osmo-tetra_amd")
_spec = importlib.util.spec_from_file_location("osmo_tetra_amd", os.path.join(_d, "__init__.py"),
                                               submodule_search_locations=[_d])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["osmo_tetra_amd"] = _mod
_spec.loader.exec_module(_mod)
