"""``EstimateRGB`` with the interface d_infer.py uses (reference: rootnet/Model_RGB.py:304-336,:494-498,:572-639):
``get_model()`` -> object with ``estimate_root_depth_custom(img, K, bbox) -> float`` (absolute root depth) and
``calculate_k``.  Only the ResNet-34 backbone and the ResRootNet head run (as in the reference's depth path, which reads
the backbone features through a forward hook and never needs the SAR mesh head)."""
from __future__ import annotations

import numpy as np
import torch

from .. import lib as L
from .. import ops, synth
from .engine import RootNetEngine
from .preprocessing import process_bbox


class EstimateRGB:
    def __init__(self, cfg):
        self.cfg = cfg
        self.mode = 'estimate'
        ck = str(cfg.checkpoint)
        if ck.startswith("synthetic"):
            seed = int(ck.split(":")[1]) if ":" in ck else 0
            net, root = synth.rootnet_state_dict(seed)
        else:
            from ..utils.checkpoint import load_checkpoint
            checkpoint = load_checkpoint(ck)                       # FileNotFoundError when missing
            net = checkpoint['net'] if 'net' in checkpoint else checkpoint['network']      # Model_RGB.py:321-324
            if 'rootnet' not in checkpoint:
                raise RuntimeError("RootNet is not loaded in the checkpoint!")          # :586-587
            root = checkpoint['rootnet']
        self.device = torch.device(cfg.device if torch.cuda.is_available() else 'cpu')
        if self.device.type != 'cuda':
            raise L.HipLibraryError("EstimateRGB runs on an MI355X only: the HIP hot path has no CPU fallback")
        self.engine = RootNetEngine(net, root, device=self.device)
        self.rootnet = self.engine
        self.mean = 255.0 * np.array([0.485, 0.456, 0.406])
        self.std = 255.0 * np.array([0.229, 0.224, 0.225])

    def calculate_k(self, bbox, fx, fy):
        """Model_RGB.py:494-498: sqrt(real_area * fx * fy / bbox_area), shape (1,)."""
        area = bbox[-1] * bbox[-2]
        real_area = torch.tensor(self.cfg.bbox_real[0] * self.cfg.bbox_real[1])
        return torch.sqrt(real_area * fx * fy / (area)).unsqueeze(0).to(self.device)

    def patch(self, img: np.ndarray, bbox_processed) -> torch.Tensor:
        """generate_patch_image + BGR->RGB + ToTensor + Normalize (:596-610) for one box, on the GPU."""
        frame = torch.from_numpy(np.ascontiguousarray(img)).to(self.device)
        cx, cy = float(bbox_processed[0] + 0.5 * bbox_processed[2]), float(bbox_processed[1] + 0.5 * bbox_processed[3])
        rec = ops.crop_boxes([(cx, cy, float(bbox_processed[2]), False)]).to(self.device)
        return ops.crop_batch(frame, rec, self.mean, self.std)

    @torch.no_grad()
    def estimate_root_depth_custom(self, img, K, bbox):
        """Model_RGB.py:572-639.  img HxWx3 uint8 BGR, K 3x3, bbox [x1, y1, x2, y2] -> root depth (float)."""
        x1, y1, x2, y2 = bbox
        height, width = img.shape[:2]
        bbox_processed = process_bbox([x1, y1, x2 - x1, y2 - y1], width, height, self.cfg.input_img_shape, 1.5)
        if bbox_processed is None:
            raise ValueError("empty bounding box")
        fx, fy = (K[0, 0], K[1, 1]) if isinstance(K, np.ndarray) else (K[0][0], K[1][1])
        k_value = self.calculate_k(bbox_processed, float(fx), float(fy))
        depth = self.engine.forward(self.patch(img, bbox_processed), k_value)
        return depth.item()


def get_model():
    from .sar_config_stage_1 import rgb_opt
    return EstimateRGB(rgb_opt)
