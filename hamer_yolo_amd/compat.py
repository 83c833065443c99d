"""Import names of the reference tree, served by this package ("keep these importable").

The reference's drivers do ``from yolo.detector import Detector``, ``from hamer.models import load_hamer``,
``from config.yolo_config import yolo_opt``, ``from config.hamer_config import hamer_opt``,
``from hamer.utils.renderer import custom_cam_crop_to_full``, ``from model.rootnet.Model_RGB import get_model``
(hamer/infer.py:15-44, d_infer.py:21) with ``sys.path`` pointing into its checkout.  ``install()`` makes exactly those
dotted names resolve to the modules of ``hamer_yolo_amd`` -- the SAME module objects, not second copies -- through a
meta-path finder, so a caller of the reference switches by adding ``import hamer_yolo_amd.compat as compat;
compat.install()`` before its own imports.  Importing this module installs nothing.

The finder defers to the path finder: a name the caller's own ``sys.path`` can serve (its own ``config.py``, its own
``model/`` package and every submodule found there) keeps resolving to the caller's code; only dotted names nobody else
can find -- and the submodules of packages that already are this package's -- resolve to the aliases below.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import sys

ALIASES = {
    "hamer": "hamer_yolo_amd.hamer",
    "yolo": "hamer_yolo_amd.yolo",
    "config": "hamer_yolo_amd.config",
    "model.rootnet": "hamer_yolo_amd.rootnet",
}


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real_name: str):
        self.real_name = real_name

    def create_module(self, spec):
        return importlib.import_module(self.real_name)      # the one and only module object

    def exec_module(self, module):
        pass


def _others_can_find(fullname, path) -> bool:
    try:
        return importlib.machinery.PathFinder.find_spec(fullname, path) is not None
    except (ImportError, ValueError):
        return False


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        top, _, _rest = fullname.partition(".")
        parent = sys.modules.get(fullname.rpartition(".")[0]) if "." in fullname else None
        ours = parent is not None and getattr(parent, "__name__", "").startswith("hamer_yolo_amd")
        # the caller's own code wins: a top-level name its sys.path can serve, and any submodule found inside a package that
        # is the caller's.  (Submodules of a package that IS one of this package's modules must be aliased here, first:
        # the path finder would otherwise load a second copy of them under the alias name.)
        if not ours and _others_can_find(fullname, path):
            return None
        if fullname == "model":                               # namespace parent of model.rootnet, when the caller has no `model` of its own
            spec = importlib.machinery.ModuleSpec("model", None, is_package=True)
            spec.submodule_search_locations = []
            return spec
        for alias, real in ALIASES.items():
            if fullname == alias or fullname.startswith(alias + "."):
                real_name = real + fullname[len(alias):]
                try:
                    real_mod = importlib.import_module(real_name)
                except ModuleNotFoundError as e:
                    if e.name == real_name:
                        return None
                    raise
                spec = importlib.machinery.ModuleSpec(fullname, _AliasLoader(real_name), is_package=hasattr(real_mod, "__path__"))
                return spec
        return None


_finder = None


def install() -> None:
    global _finder
    if _finder is not None:
        return
    for alias in list(ALIASES) + ["model"]:
        top = alias.split(".")[0]
        mod = sys.modules.get(top)
        if mod is not None and not getattr(mod, "__name__", "").startswith("hamer_yolo_amd") and top != "model":
            raise ImportError(f"hamer_yolo_amd.compat: the name {top!r} is already imported from {getattr(mod, '__file__', '?')}")
    _finder = _AliasFinder()
    sys.meta_path.insert(0, _finder)         # first, but it defers to the path finder for everything the caller's sys.path can serve


def uninstall() -> None:
    global _finder
    if _finder is None:
        return
    sys.meta_path.remove(_finder)
    _finder = None
    for name in [n for n in sys.modules if n == "model" or any(n == a or n.startswith(a + ".") for a in ALIASES)]:
        del sys.modules[name]

