"""Model configuration (reference: hamer/hamer/configs/__init__.py:88-113 ``get_config`` on a yacs
CfgNode).  yacs is not a dependency here: ``CfgNode`` below is a small attribute-dict with the
operations infer.py uses (attribute access, ``.get``, ``dict(node)``, ``in``)."""
import os
from typing import Any, Dict, Optional

CACHE_DIR_HAMER = "./_DATA"


class CfgNode(dict):
    def __init__(self, d: Optional[Dict[str, Any]] = None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def merge(self, other: Dict[str, Any]):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else v
        return self

    def defrost(self):
        return self

    def freeze(self):
        return self

    def clone(self):
        return CfgNode(self)


def default_config() -> CfgNode:
    """Values the inference path reads (configs/__init__.py:19-72 defaults,
    configs_hydra/experiment/{default,hamer_vit_transformer}.yaml)."""
    return CfgNode({
        "MODEL": {
            "IMAGE_SIZE": 256, "IMAGE_MEAN": [0.485, 0.456, 0.406], "IMAGE_STD": [0.229, 0.224, 0.225],
            "BACKBONE": {"TYPE": "vit"},
            "MANO_HEAD": {"TYPE": "transformer_decoder", "IN_CHANNELS": 2048,
                          "TRANSFORMER_DECODER": {"depth": 6, "heads": 8, "mlp_dim": 1024, "dim_head": 64, "dropout": 0.0,
                                                  "emb_dropout": 0.0, "norm": "layer", "context_dim": 1280}},
        },
        "EXTRA": {"FOCAL_LENGTH": 5000},
        "MANO": {"DATA_DIR": "_DATA/data/", "MODEL_PATH": "_DATA/data/mano", "GENDER": "neutral", "NUM_HAND_JOINTS": 15,
                 "MEAN_PARAMS": "_DATA/data/mano_mean_params.npz", "CREATE_BODY_POSE": False},
    })


def get_config(config_file: Optional[str], merge: bool = True, update_cachedir: bool = False) -> CfgNode:
    cfg = default_config() if merge else CfgNode()
    if config_file and os.path.exists(config_file):
        import yaml
        with open(config_file) as f:
            cfg.merge(yaml.safe_load(f) or {})
    if update_cachedir:
        for key in ("MODEL_PATH", "MEAN_PARAMS"):
            p = cfg.MANO[key]
            if not os.path.isabs(p):
                cfg.MANO[key] = os.path.join(CACHE_DIR_HAMER, p)
    return cfg
