"""Loads the package next to this directory as `tip_amd` (its directory name, `transformer-inertial-poser_amd`, is not a
Python identifier).  The three modules beside this file are the ONLY ones meant to shadow files of the reference; this helper
has a name no reference file uses."""
import importlib.util
import os
import sys


def load():
    if "tip_amd" in sys.modules:
        return sys.modules["tip_amd"]
    pkg_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("tip_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["tip_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
