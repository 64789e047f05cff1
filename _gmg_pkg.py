"""Import helper: the package directory is `glimmer-mg_amd/` (a hyphen is not a legal module
name), so it is loaded under the module name `glimmer_mg_amd`."""
import importlib.util
import os
import sys

_NAME = "glimmer_mg_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "glimmer-mg_amd")
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
