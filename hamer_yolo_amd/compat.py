"""Import names of the reference tree, served by this package ("keep these importable").

The reference's drivers do ``from yolo.detector import Detector``, ``from hamer.models import load_hamer``,
``from config.yolo_config import yolo_opt``, ``from config.hamer_config import hamer_opt``,
``from hamer.utils.renderer import custom_cam_crop_to_full``, ``from model.rootnet.Model_RGB import get_model``
(hamer/infer.py:15-44, d_infer.py:21) with ``sys.path`` pointing into its checkout.  ``install()`` makes exactly those
dotted names resolve to the modules of ``hamer_yolo_amd`` -- the SAME module objects, not second copies -- through a
meta-path finder, so a caller of the reference switches by adding one line, ``import hamer_yolo_amd.compat``, before its
own imports.  Nothing is installed when one of the names is already taken by another package.
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


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == "model":                               # namespace parent of model.rootnet
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
    sys.meta_path.insert(0, _finder)


def uninstall() -> None:
    global _finder
    if _finder is None:
        return
    sys.meta_path.remove(_finder)
    _finder = None
    for name in [n for n in sys.modules if n == "model" or any(n == a or n.startswith(a + ".") for a in ALIASES)]:
        del sys.modules[name]


install()
