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


def checkpoint_state_dict(ck):
    """The UNFUSED fp32 state dict inside what ``torch.save`` wrote: the reference's ``{'model': Model, 'ema': Model|None,
    ...}`` with pickled modules (the EMA weights win when present, experimental.py:266), a ``{'model': state_dict}`` /
    ``{'state_dict': ...}`` wrapper, or a bare state dict.  Keys come out as ``model.<i>...`` (the ``Model.model``
    Sequential), as ``Model.state_dict()`` names them."""
    from ..utils.checkpoint import is_module, module_state_dict
    if isinstance(ck, dict) and ("model" in ck or "ema" in ck or "state_dict" in ck):
        ck = ck.get("ema") or ck.get("model") or ck.get("state_dict")
    sd = module_state_dict(ck) if is_module(ck) else ck
    if not isinstance(sd, dict) or not all(isinstance(v, torch.Tensor) for v in sd.values()):
        raise TypeError("not a YOLOv7 checkpoint: expected pickled modules or a state dict of tensors")
    return {k: v.float() for k, v in sd.items() if v.is_floating_point()}


def attempt_load(weights: str):
    """experimental.py:260-283 equivalent: returns (UNFUSED state dict, nc, class names) -- BN / RepConv / implicit folding
    happens in YoloEngine (fuse.py); the detect head's ``anchor_grid`` buffer stays in the state dict and is what the decode
    uses.  ``"synthetic:<seed>"`` draws seeded random-init weights; a file is read with the class-free
    unpickler (utils/checkpoint.py), so the reference's own ``yolov7_best.pt`` loads without its ``models`` package."""
    weights = str(weights)
    if weights.startswith("synthetic"):
        seed = int(weights.split(":")[1]) if ":" in weights else 0
        return synth.yolo_state_dict(seed=seed, nc=3), 3, ['0', '1', '2']
    from ..utils.checkpoint import load_checkpoint
    ck = load_checkpoint(weights)                                      # FileNotFoundError when missing
    sd = checkpoint_state_dict(ck)
    det = [k for k in sd if k.endswith(".m.0.weight")]
    if not det:
        raise TypeError("YOLOv7 checkpoint without a detect head (no '*.m.0.weight')")
    nc = sd[det[0]].shape[0] // 3 - 5
    mod = (ck.get("ema") or ck.get("model")) if isinstance(ck, dict) else ck
    names = getattr(mod, "names", None)                                # Model.names (yolo.py:533), read at detector.py:157
    names = [str(n) for n in names] if isinstance(names, (list, tuple)) and len(names) == nc else [str(i) for i in range(nc)]
    return sd, nc, names


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
        sd, nc, names = attempt_load(weights)
        stride = 32
        self.imgsz = int(np.ceil(imgsz / stride) * stride)               # check_img_size, general.py:126-131
        self.engine = YoloEngine(sd, nc=nc, device=self.device, new_shape=self.imgsz, stride=stride, names=names)
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
