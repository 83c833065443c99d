"""Reading the reference's checkpoints without the reference's classes.

``yolov7_best.pt`` is a pickled ``nn.Module`` tree (``{'model': Model, 'ema': ...}``, loaded by ``attempt_load``,
experimental.py:260-283, which needs ``models.yolo.Model`` & co. importable), and a Lightning ``hamer.ckpt`` may
carry ``yacs`` / ``pytorch_lightning`` objects beside its ``state_dict``.  Neither package tree exists here, and
unpickling arbitrary globals is unsafe anyway, so ``load_checkpoint`` runs ``torch.load`` with an unpickler that
  * lets through tensors, storages, containers and numpy arrays (an allow-list of exact (module, name) pairs), and
  * turns every other global into an inert attribute bag (``Stub``): a pickled ``os.system`` / ``eval`` / import helper is
    never resolved, so calling it (REDUCE) only builds another ``Stub``.
``module_state_dict`` then walks a stubbed module tree (``_parameters`` / ``_buffers`` / ``_modules``) and
returns what ``module.state_dict()`` would have.
"""
from __future__ import annotations

import pickle
from collections import OrderedDict
from typing import Any, Dict

import torch

# Exact (module, name) pairs that are resolved to the real object: constructors of tensors, storages, containers and
# numpy arrays -- nothing that imports, evaluates or calls through a name taken from the pickle.  (A module-PREFIX
# allow-list is not enough: ``torch._utils._import_dotted_name`` or ``numpy.testing._private.utils.runstring`` would be
# reachable through it.)  Everything else becomes an inert ``Stub``.
_TORCH_DTYPES = {n for n in dir(torch) if isinstance(getattr(torch, n, None), torch.dtype)}
_ALLOWED_EXACT = {
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"), ("torch.nn.parameter", "Parameter"),
    ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"),
    ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("copyreg", "_reconstructor"), ("copy_reg", "_reconstructor"), ("builtins", "object"), ("builtins", "set"), ("builtins", "frozenset"),
    ("builtins", "slice"), ("builtins", "range"), ("builtins", "complex"), ("builtins", "bytearray"),
    ("_codecs", "encode"), ("__builtin__", "object"), ("__builtin__", "set"), ("__builtin__", "frozenset"),
}


def _allowed(module: str, name: str) -> bool:
    if (module, name) in _ALLOWED_EXACT:
        return True
    if module == "torch" and (name in _TORCH_DTYPES or (name.endswith("Storage") and name[:-7].isalpha())):
        return True                                 # torch.float32, torch.FloatStorage, ...: plain classes / dtype singletons
    return False


class Stub:
    """An unpickled foreign object: its state, no behaviour."""

    def __init__(self, *args, **kwargs):
        self._stub_args = args

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and all(isinstance(s, (dict, type(None))) for s in state):
            for s in state:
                if s:
                    self.__dict__.update(s)
        else:
            self._stub_state = state

    def __call__(self, *args, **kwargs):        # e.g. a pickled partial / functools object being "called"
        return Stub()

    # dict / list / set subclasses (yacs.CfgNode, ...) are rebuilt through the container protocol
    def __setitem__(self, key, value):
        self.__dict__.setdefault("_stub_items", {})[key] = value

    def __getitem__(self, key):
        return self.__dict__.get("_stub_items", {})[key]

    def append(self, value):
        self.__dict__.setdefault("_stub_list", []).append(value)

    def extend(self, values):
        self.__dict__.setdefault("_stub_list", []).extend(values)

    def add(self, value):
        self.__dict__.setdefault("_stub_list", []).append(value)

    def __repr__(self):
        return f"<Stub {type(self).__module__}.{type(self).__qualname__}>"


_stub_classes: Dict[tuple, type] = {}


def _stub_class(module: str, name: str) -> type:
    key = (module, name)
    if key not in _stub_classes:
        _stub_classes[key] = type(name, (Stub,), {"__module__": module})
    return _stub_classes[key]


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if _allowed(module, name):
            return super().find_class(module, name)
        return _stub_class(module, name)


class _PickleModule:
    """The duck-typed ``pickle_module`` torch.load asks for."""
    __name__ = "hamer_yolo_amd.utils.checkpoint"
    Unpickler = _Unpickler
    UnpicklingError = pickle.UnpicklingError

    @staticmethod
    def load(f, **kwargs):
        return _Unpickler(f, **kwargs).load()


def load_checkpoint(path: str) -> Any:
    """torch.load(path) with every non-tensor class replaced by ``Stub``.  FileNotFoundError when missing."""
    return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_PickleModule)


def module_state_dict(mod: Any, prefix: str = "") -> "OrderedDict[str, torch.Tensor]":
    """``nn.Module.state_dict()`` of a stubbed (or real) module tree."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    d = getattr(mod, "__dict__", {})
    non_persistent = d.get("_non_persistent_buffers_set", set()) or set()
    for name, p in (d.get("_parameters") or {}).items():
        if p is not None:
            out[prefix + name] = (p.data if hasattr(p, "data") else p).detach()
    for name, b in (d.get("_buffers") or {}).items():
        if b is not None and name not in non_persistent:
            out[prefix + name] = b.detach()
    for name, m in (d.get("_modules") or {}).items():
        if m is not None:
            out.update(module_state_dict(m, prefix + name + "."))
    return out


def is_module(obj: Any) -> bool:
    return isinstance(getattr(obj, "__dict__", None), dict) and "_modules" in obj.__dict__ and "_parameters" in obj.__dict__
