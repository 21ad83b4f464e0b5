"""Import helper: the product package lives in `minigpt4.cpp_amd/` (dot in the name), so it is registered
under the importable alias `minigpt4_cpp_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "minigpt4.cpp_amd")
ALIAS = "minigpt4_cpp_amd"


def load_package():
    if ALIAS in sys.modules:
        return sys.modules[ALIAS]
    spec = importlib.util.spec_from_file_location(ALIAS, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod
