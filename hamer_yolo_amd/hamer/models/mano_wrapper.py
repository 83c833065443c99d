"""MANO parameter container (reference: hamer/hamer/models/mano_wrapper.py:12-44, a subclass of
smplx.MANOLayer).  smplx is not a dependency: the arithmetic is the HIP kernel hm_mano_forward;
this class only loads / holds the model arrays.

The licensed MANO arrays are never shipped.  ``MANO.from_pkl(path)`` reads a user-supplied
``MANO_RIGHT.pkl`` (chumpy pickle) with a restricted unpickler; ``MANO.synthetic(seed)`` builds
MANO-shaped random parameters for benchmarks and tests.
"""
import io
import os
import pickle
from typing import Dict

import numpy as np
import torch

from ... import synth

MANO_TO_OPENPOSE = synth.MANO_JOINT_MAP


class _Stub:
    """Stands in for chumpy.ch.Ch / chumpy.reordering.Select objects inside the MANO pickle."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"state": state})


class _RestrictedUnpickler(pickle.Unpickler):
    """Exact (module, name) allow-list for the MANO pickle: numpy array reconstruction and scipy's CSC matrix (the joint
    regressor); chumpy wrappers become ``_Stub``; anything else is refused (no prefix matching: numpy and scipy expose
    helpers that evaluate strings)."""

    def find_class(self, module, name):
        if module.startswith("chumpy."):
            return _Stub
        if (module, name) in (("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct")):
            return (getattr(np, "_core", None) or np.core).multiarray._reconstruct
        if (module, name) == ("numpy", "ndarray"):
            return np.ndarray
        if (module, name) == ("numpy", "dtype"):
            return np.dtype
        if name in ("csc_matrix", "csr_matrix", "coo_matrix") and module in ("scipy.sparse.csc", "scipy.sparse._csc", "scipy.sparse.csr",
                                                                             "scipy.sparse._csr", "scipy.sparse.coo", "scipy.sparse._coo"):
            import scipy.sparse
            return getattr(scipy.sparse, name)
        if module in ("__builtin__", "builtins") and name in ("set", "frozenset", "list", "dict", "tuple"):
            return {"set": set, "frozenset": frozenset, "list": list, "dict": dict, "tuple": tuple}[name]
        if (module, name) in (("_codecs", "encode"), ("copy_reg", "_reconstructor"), ("copyreg", "_reconstructor")):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"MANO pickle: refusing global {module}.{name}")


def _arr(x):
    if isinstance(x, _Stub):
        for key in ("x", "a", "r"):
            if key in x.__dict__:
                return _arr(x.__dict__[key])
        raise ValueError("cannot read array from chumpy object")
    if hasattr(x, "toarray"):
        return np.asarray(x.toarray())
    return np.asarray(x)


class MANO:
    def __init__(self, params: Dict[str, torch.Tensor]):
        self.params = params
        self.faces = params["faces"].cpu().numpy() if "faces" in params else None
        self.joint_map = torch.tensor(MANO_TO_OPENPOSE, dtype=torch.long)

    @classmethod
    def synthetic(cls, seed: int = 0) -> "MANO":
        return cls(synth.mano_params(seed))

    @classmethod
    def from_pkl(cls, path: str) -> "MANO":
        if os.path.isdir(path):
            path = os.path.join(path, "MANO_RIGHT.pkl")
        with open(path, "rb") as f:
            data = _RestrictedUnpickler(io.BytesIO(f.read()), encoding="latin1").load()
        V = _arr(data["v_template"]).shape[0]
        p = {
            "v_template": torch.from_numpy(_arr(data["v_template"]).astype(np.float32)),
            "shapedirs": torch.from_numpy(_arr(data["shapedirs"]).astype(np.float32)[:, :, :10].copy()),
            # smplx stores posedirs as (135, 3V): reshape of (V,3,135) -> (3V,135) -> transpose
            "posedirs": torch.from_numpy(_arr(data["posedirs"]).astype(np.float32).reshape(V * 3, -1).T.copy()),
            "J_regressor": torch.from_numpy(_arr(data["J_regressor"]).astype(np.float32)),
            "lbs_weights": torch.from_numpy(_arr(data["weights"]).astype(np.float32)),
            "faces": torch.from_numpy(_arr(data["f"]).astype(np.int64)),
        }
        kt = _arr(data["kintree_table"]).astype(np.int64)
        parents = kt[0].copy()
        parents[0] = -1
        p["parents"] = torch.from_numpy(parents)
        if parents.tolist() != synth.MANO_PARENTS:
            raise ValueError("unexpected MANO kinematic tree")
        return cls(p)
