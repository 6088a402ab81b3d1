"""Loads the package directory `extendablesparse.jl_amd/` (its name is not a Python identifier)
under the importable alias `extendablesparse_jl_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "extendablesparse.jl_amd")
ALIAS = "extendablesparse_jl_amd"


def load():
    if ALIAS in sys.modules:
        return sys.modules[ALIAS]
    spec = importlib.util.spec_from_file_location(ALIAS, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod
