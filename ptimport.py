"""Loads the package directory `pathtracer-0_amd/` under the importable name `pathtracer_0_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
NAME = "pathtracer_0_amd"


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    pkg = os.path.join(ROOT, "pathtracer-0_amd")
    spec = importlib.util.spec_from_file_location(NAME, os.path.join(pkg, "__init__.py"), submodule_search_locations=[pkg])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    spec.loader.exec_module(mod)
    return mod
