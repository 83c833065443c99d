"""HAMER model object with the call surface infer.py uses (reference: hamer/hamer/models/hamer.py:
``forward_step`` :99-156, ``forward`` :269-277).  All arithmetic is one hm_hamer_forward enqueue
(HamerEngine); this class assembles the reference's output dictionaries from the kernel outputs.
Training (losses, discriminator, Lightning hooks) is out of scope.
"""
from typing import Dict, Optional, Tuple

import torch

from ... import synth
from ...engine import HamerEngine
from ...lib import HipLibraryError
from .mano_wrapper import MANO


class HAMER:
    def __init__(self, cfg, state_dict: Dict[str, torch.Tensor], mano: MANO, dtype=torch.float16,
                 hamer_cfg: Optional[synth.HamerConfig] = None):
        self.cfg = cfg
        self.mano = mano
        self.dtype = dtype
        self._sd = state_dict
        self._hc = hamer_cfg or synth.HamerConfig(image_size=int(cfg.MODEL.IMAGE_SIZE), focal_length=float(cfg.EXTRA.FOCAL_LENGTH))
        self._engine: Optional[HamerEngine] = None
        self.device = torch.device("cpu")
        self.training = False

    # -- nn.Module-like surface used by hamer_inference.__init__ (infer.py:139-140)
    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise HipLibraryError("HAMER runs on an MI355X only: the HIP hot path has no CPU fallback")
        if self._engine is None or self.device != device:
            self._engine = HamerEngine(self._sd, self.mano.params, self._hc, device=device, dtype=self.dtype)
            self.device = device
        return self

    def eval(self):
        self.training = False
        return self

    def forward_step(self, batch: Dict, train: bool = False) -> Tuple[Dict, Dict]:
        if self._engine is None:
            raise HipLibraryError("call model.to('cuda') before the first forward")
        x = batch["img"]
        B = x.shape[0]
        o = self._engine.forward(x.to(self.device, torch.float32))
        R = o["rotmats"]
        pred_mano_params = {"global_orient": R[:, :1], "hand_pose": R[:, 1:], "betas": o["betas"]}
        output = {
            "pred_cam": o["pred_cam"],
            "pred_mano_params": {k: v.clone() for k, v in pred_mano_params.items()},
            "pred_cam_t": o["pred_cam_t"],
            "focal_length": self.cfg.EXTRA.FOCAL_LENGTH * torch.ones(B, 2, device=self.device, dtype=torch.float32),
            "pred_keypoints_3d": o["pred_keypoints_3d"].reshape(B, -1, 3),
            "pred_vertices": o["pred_vertices"].reshape(B, -1, 3),
            "pred_keypoints_2d": o["pred_keypoints_2d"].reshape(B, -1, 2),
        }
        pred_mano_params["trans"] = o["pred_cam_t"]
        return output, pred_mano_params

    def forward(self, batch: Dict) -> Tuple[Dict, Dict]:
        return self.forward_step(batch, train=False)

    __call__ = forward
