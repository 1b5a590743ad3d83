"""Import shim: the package directory is named `transformer-inertial-poser_amd/` (not a valid Python
identifier), so `import tip_amd` loads it under this name.  Nothing else lives here."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "transformer-inertial-poser_amd")
_spec = importlib.util.spec_from_file_location(
    "tip_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["tip_amd"] = _mod
_spec.loader.exec_module(_mod)
