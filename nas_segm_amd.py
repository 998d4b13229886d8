"""Import shim: exposes the package directory ``nas-segm-pytorch_amd/`` (not a
valid identifier) under the importable name ``nas_segm_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nas-segm-pytorch_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _pkg
_spec.loader.exec_module(_pkg)
