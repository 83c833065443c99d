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
from .preprocessing import patch_boxes, process_bbox


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

    def _k_host(self, bbox, fx, fy):
        area = bbox[-1] * bbox[-2]
        real_area = torch.tensor(self.cfg.bbox_real[0] * self.cfg.bbox_real[1])
        return torch.sqrt(real_area * fx * fy / (area)).unsqueeze(0)

    def calculate_k(self, bbox, fx, fy):
        """Model_RGB.py:494-498: sqrt(real_area * fx * fy / bbox_area), shape (1,)."""
        return self._k_host(bbox, fx, fy).to(self.device)

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


    # -------------------------------------------------------------------------------- batched form (d_infer's folder driver)
    def valid_boxes(self, dets, width, height):
        """Which detections [[label, [x1, y1, x2, y2]], ...] of a width x height frame have a RootNet patch at all (the ones
        estimate_root_depth_custom would raise on -- the reference's per-hand try/except skips those hands)."""
        if not dets:
            return np.zeros(0, dtype=bool)
        xywh = np.array([[d[1][0], d[1][1], d[1][2] - d[1][0], d[1][3] - d[1][1]] for d in dets], dtype=np.float64)
        return patch_boxes(xywh, width, height, self.cfg.input_img_shape, 1.5)[1]

    @torch.no_grad()
    def estimate_root_depths_frames(self, frames, K, dets_lists) -> torch.Tensor:
        """estimate_root_depth_custom for ALL hands of several device-resident frames ((H,W,3) uint8 BGR tensors) in one
        RootNet forward: one crop launch per frame into one batch tensor, one pass of the backbone.  Per hand the same
        box arithmetic, the same crop and the same k as the one-hand call, so the depths are the same numbers.  Every
        detection must have a patch (filter with valid_boxes first).  Returns (n,) fp32 on the device, hands in
        ``dets_lists`` order."""
        fx, fy = (K[0, 0], K[1, 1]) if isinstance(K, np.ndarray) else (K[0][0], K[1][1])
        P = int(self.cfg.input_img_shape[0])
        recs, kvs, counts = [], [], []
        for fr, dets in zip(frames, dets_lists):
            height, width = int(fr.shape[0]), int(fr.shape[1])
            for _, (x1, y1, x2, y2) in dets:
                bp = process_bbox([x1, y1, x2 - x1, y2 - y1], width, height, self.cfg.input_img_shape, 1.5)
                if bp is None:
                    raise ValueError("empty bounding box")
                recs.append((float(bp[0] + 0.5 * bp[2]), float(bp[1] + 0.5 * bp[3]), float(bp[2]), False))
                kvs.append(self._k_host(bp, float(fx), float(fy)))          # host arithmetic (correctly rounded ops: the same bits), one upload
            counts.append(len(dets))
        n = len(recs)
        if n == 0:
            return torch.empty(0, device=self.device)
        rec = ops.crop_boxes(recs, P).to(self.device)
        rsz = rec.numel() // n
        img = torch.empty(n, 3, P, P, device=self.device, dtype=torch.float32)
        off = 0
        for fr, k in zip(frames, counts):
            if k:
                ops.crop_batch(fr, rec[off * rsz:(off + k) * rsz], self.mean, self.std, P, out=img[off:off + k])
                off += k
        return self.engine.forward(img, torch.cat(kvs))


def get_model():
    from .sar_config_stage_1 import rgb_opt
    return EstimateRGB(rgb_opt)
