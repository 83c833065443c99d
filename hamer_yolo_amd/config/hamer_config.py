"""HaMeR paths (reference: config/hamer_config.py:4-11; same attribute names).

The reference hard-codes the author's checkpoint directory; here the root comes from
``HAMER_MODEL_ROOT`` and ``ckpt_path`` may also be ``"synthetic:<seed>"`` to run on seeded
random-init weights (no checkpoint ships with either repository).
"""
import os


class Config:
    root_dir = os.environ.get("HAMER_MODEL_ROOT", "/home/pt/fbs/model")
    ckpt_path = os.environ.get("HAMER_CKPT", os.path.join(root_dir, "hamer/_DATA/hamer_ckpts/checkpoints/hamer.ckpt"))
    model_cfg = os.path.join(root_dir, "hamer/_DATA/hamer_ckpts/model_config.yaml")
    onnx_path = os.path.join(root_dir, "hamer/_DATA/hamer_ckpts/onnx/hamer_inferpy.onnx")
    use_onnx = False      # the ONNX path is out of scope (BASELINE.json north_star): True raises


hamer_opt = Config()
