"""``Detector`` with the reference's interface (reference: yolo/detector.py:22-153).

``Detector(config)`` reads ``config.{weights, imgsz, augment, conf_thres, iou_thres, classes, agnostic_nms,
device, save_path}`` (config/yolo_config.py:4-13); ``detect(image)`` takes an HxWx3 uint8 BGR frame and returns
``(pred, dets_list)``: ``pred`` a list with one (n, 6) tensor [x1, y1, x2, y2, conf, cls] in frame pixels (rounded),
``dets_list`` a list with one list of ``[label, [x1, y1, x2, y2]]``, label 'right' iff cls == 1
(detector.py:144-147).  ``augment`` is ignored exactly as in the reference (TracedModel.forward drops it,
utils/torch_utils.py:371-374).  Everything between the frame upload and the box list is HIP
(letterbox, 92 implicit-GEMM convolutions, pooling, decode, NMS, scale_coords).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import lib as L
from .. import synth
from .engine import YoloEngine


def attempt_load(weights: str):
    """experimental.py:260-283 equivalent: returns an UNFUSED state dict.  ``"synthetic:<seed>"`` draws seeded
    random-init weights; otherwise a torch-saved dict of tensors (``{'model': state_dict}`` or a bare state dict)."""
    weights = str(weights)
    if weights.startswith("synthetic"):
        seed = int(weights.split(":")[1]) if ":" in weights else 0
        return synth.yolo_state_dict(seed=seed, nc=3), 3
    ck = torch.load(weights, map_location="cpu", weights_only=True)      # FileNotFoundError when missing
    sd = ck.get("model", ck.get("state_dict", ck)) if isinstance(ck, dict) else ck
    if not isinstance(sd, dict):
        raise TypeError("expected a state dict; convert pickled YOLOv7 modules with tools/convert_yolo_checkpoint.py")
    det = [k for k in sd if k.endswith(".m.0.weight")]
    nc = sd[det[0]].shape[0] // 3 - 5
    return sd, nc


class _Model:
    """What callers read from ``detector.model`` (names, stride)."""

    def __init__(self, engine: YoloEngine):
        self.engine = engine
        self.names = engine.names
        self.stride = torch.tensor([8.0, 16.0, 32.0])


class Detector():
    def __init__(self, config):
        weights, imgsz, self.device = config.weights, config.imgsz, config.device
        self.device = torch.device(self.device if torch.cuda.is_available() else 'cpu')
        if self.device.type != 'cuda':
            raise L.HipLibraryError("Detector runs on an MI355X only: the HIP hot path has no CPU fallback")
        sd, nc = attempt_load(weights)
        stride = 32
        self.imgsz = int(np.ceil(imgsz / stride) * stride)               # check_img_size, general.py:126-131
        self.engine = YoloEngine(sd, nc=nc, device=self.device, new_shape=self.imgsz, stride=stride)
        self.model = _Model(self.engine)
        self.opt = config
        self.detect_savepath = config.save_path

    @torch.no_grad()
    def detect(self, image: np.ndarray):
        opt = self.opt
        frame = torch.from_numpy(np.ascontiguousarray(image)).to(self.device)
        p = self.engine.forward(frame)
        det = self.engine.nms(p, opt.conf_thres, opt.iou_thres, opt.classes, opt.agnostic_nms, scale=True)
        dets = []
        for row in det.tolist():
            dets.append(['right' if row[-1] == 1 else 'left', row[:4]])
        return [det], [dets]
