"""Import names of the reference tree, served by this package ("keep these importable").

The reference's drivers do ``from yolo.detector import Detector``, ``from hamer.models import load_hamer``,
``from config.yolo_config import yolo_opt``, ``from config.hamer_config import hamer_opt``,
``from hamer.utils.renderer import custom_cam_crop_to_full``, ``from model.rootnet.Model_RGB import get_model``
(hamer/infer.py:15-44, d_infer.py:21) with ``sys.path`` pointing into its checkout.  ``install()`` makes exactly those
dotted names resolve to the modules of ``hamer_yolo_amd`` -- the SAME module objects, not second copies -- through a
meta-path finder, so a caller of the reference switches by adding ``import hamer_yolo_amd.compat as compat;
compat.install()`` before its own imports.  Importing this module installs nothing.

Who wins when the caller's ``sys.path`` can serve a name too:
  * ``hamer`` and ``yolo`` ALWAYS resolve to this package.  The documented caller runs the reference's drivers with ``sys.path``
    pointing into the reference checkout, which has ``hamer/`` and ``yolo/`` at its root: deferring to the path finder there
    would make ``install()`` a silent no-op and import the CUDA reference instead.  A foreign package of that name on the path
    is shadowed, with one ``ImportWarning`` naming it.
  * ``config``: a ``config`` on the path that holds ``hamer_config.py`` / ``yolo_config.py`` is the reference's and is shadowed
    the same way; any other ``config`` (the caller's own ``config.py``) keeps resolving to the caller's code.
  * ``model``: the caller's own ``model/`` package and everything found inside it wins; only ``model.rootnet`` (and ``model``
    itself when nobody has one) fall through to this package.
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


def _path_spec(fullname, path):
    try:
        return importlib.machinery.PathFinder.find_spec(fullname, path)
    except (ImportError, ValueError):
        return None


def _is_reference_config(spec) -> bool:
    """True when a `config` found on the path is the reference checkout's package (it holds hamer_config.py / yolo_config.py)."""
    import os
    for loc in (spec.submodule_search_locations or []):
        if any(os.path.exists(os.path.join(loc, f)) for f in ("hamer_config.py", "yolo_config.py")):
            return True
    return False


_warned = set()


def _shadow_warning(top, spec):
    import warnings
    where = getattr(spec, "origin", None) or list(spec.submodule_search_locations or ["?"])[0]
    if (top, where) not in _warned:
        _warned.add((top, where))
        warnings.warn(f"hamer_yolo_amd.compat: {top!r} resolves to hamer_yolo_amd; the one at {where} on sys.path is shadowed",
                      ImportWarning, stacklevel=3)


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        top, _, _rest = fullname.partition(".")
        parent = sys.modules.get(fullname.rpartition(".")[0]) if "." in fullname else None
        ours = parent is not None and getattr(parent, "__name__", "").startswith("hamer_yolo_amd")
        # (Submodules of a package that IS one of this package's modules must be aliased here, first: the path finder would
        # otherwise load a second copy of them under the alias name.)
        if not ours:
            foreign = _path_spec(fullname, path)
            if foreign is not None:
                shadow = top in ("hamer", "yolo") or (top == "config" and fullname == "config" and _is_reference_config(foreign))
                if not shadow:
                    return None                               # the caller's own code wins
                if fullname == top:
                    _shadow_warning(top, foreign)
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
    sys.meta_path.insert(0, _finder)         # first; what it leaves to the caller's sys.path is stated in the module docstring


def uninstall() -> None:
    global _finder
    if _finder is None:
        return
    sys.meta_path.remove(_finder)
    _finder = None
    for name in [n for n in sys.modules if n == "model" or any(n == a or n.startswith(a + ".") for a in ALIASES)]:
        del sys.modules[name]

