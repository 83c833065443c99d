"""Model-loader API (reference: hamer/hamer/models/__init__.py:32-47).

``load_hamer(checkpoint_path) -> (model, model_cfg)`` keeps the reference's contract:
``model_config.yaml`` is looked up at ``<ckpt>/../../model_config.yaml`` (:35), BBOX_SHAPE is
forced to [192, 256] for the ViT backbone (:39-43).  Accepted checkpoints:
  * a Lightning ``.ckpt`` / ``.pt`` whose ``state_dict`` has the reference's keys
    (``backbone.*``, ``mano_head.*``); MANO arrays come from ``cfg.MANO.MODEL_PATH/MANO_RIGHT.pkl``
    and mean parameters from ``cfg.MANO.MEAN_PARAMS`` when the checkpoint lacks them;
  * ``"synthetic:<seed>"``: seeded random-init weights and MANO-shaped parameters (what tests and
    bench.py use -- neither repository ships weights).
The reference overwrites the argument with a hard-coded path (:45); that quirk is not reproduced.
"""
import os
from pathlib import Path

import numpy as np
import torch

from ... import synth
from ..configs import CACHE_DIR_HAMER, get_config
from .hamer import HAMER
from .mano_wrapper import MANO

DEFAULT_CHECKPOINT = f"{CACHE_DIR_HAMER}/hamer_ckpts/checkpoints/hamer.ckpt"


def _read_state_dict(path: str):
    """``state_dict`` of a Lightning checkpoint (or a bare state dict).  Read with the class-free unpickler: the
    hyper-parameters stored beside the weights hold yacs / pytorch_lightning objects, which this build does not import."""
    from ...utils.checkpoint import load_checkpoint
    ck = load_checkpoint(path)
    sd = ck.get("state_dict", ck) if isinstance(ck, dict) else ck
    return {k: v for k, v in sd.items() if isinstance(v, torch.Tensor)}


def load_hamer(checkpoint_path=DEFAULT_CHECKPOINT, dtype=torch.float16):
    checkpoint_path = str(checkpoint_path)
    if checkpoint_path.startswith("synthetic"):
        seed = int(checkpoint_path.split(":")[1]) if ":" in checkpoint_path else 0
        model_cfg = get_config(None)
        model_cfg.MODEL.BBOX_SHAPE = [192, 256]
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        sd = synth.hamer_state_dict(synth.HamerConfig(), seed=seed, device=dev)
        return HAMER(model_cfg, sd, MANO.synthetic(seed), dtype=dtype), model_cfg

    model_cfg = str(Path(checkpoint_path).parent.parent / "model_config.yaml")
    model_cfg = get_config(model_cfg, update_cachedir=True)
    if (model_cfg.MODEL.BACKBONE.TYPE == "vit") and ("BBOX_SHAPE" not in model_cfg.MODEL):
        assert model_cfg.MODEL.IMAGE_SIZE == 256, f"MODEL.IMAGE_SIZE ({model_cfg.MODEL.IMAGE_SIZE}) should be 256 for ViT backbone"
        model_cfg.MODEL.BBOX_SHAPE = [192, 256]
    sd = _read_state_dict(checkpoint_path)            # FileNotFoundError when the weights are missing
    if "mano_head.init_hand_pose" not in sd:          # buffers normally travel in the checkpoint
        mean = np.load(model_cfg.MANO.MEAN_PARAMS)     # mano_head.py:53-59
        sd["mano_head.init_hand_pose"] = torch.from_numpy(mean["pose"].astype(np.float32)).unsqueeze(0)
        sd["mano_head.init_betas"] = torch.from_numpy(mean["shape"].astype(np.float32)).unsqueeze(0)
        sd["mano_head.init_cam"] = torch.from_numpy(mean["cam"].astype(np.float32)).unsqueeze(0)
    mano = MANO.from_pkl(os.path.join(model_cfg.MANO.MODEL_PATH, "MANO_RIGHT.pkl"))
    return HAMER(model_cfg, sd, mano, dtype=dtype), model_cfg
