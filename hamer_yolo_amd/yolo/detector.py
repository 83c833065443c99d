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
    if weights.startswith("synthetic"):                                # synthetic[:seed[:obj_bias[:cls_bias]]]
        f = weights.split(":")[1:]
        kw = {"seed": int(f[0]) if len(f) > 0 else 0}
        if len(f) > 1:
            kw["obj_bias"] = float(f[1])
        if len(f) > 2:
            kw["cls_bias"] = float(f[2])
        return synth.yolo_state_dict(nc=3, **kw), 3, ['0', '1', '2']
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

    @staticmethod
    def _labelled(rows):
        return [['right' if row[-1] == 1 else 'left', row[:4]] for row in rows]        # detector.py:144-147

    @torch.no_grad()
    def detect(self, image: np.ndarray):
        opt = self.opt
        frame = torch.from_numpy(np.ascontiguousarray(image)).to(self.device)
        p = self.engine.forward(frame)
        det = self.engine.nms(p, opt.conf_thres, opt.iou_thres, opt.classes, opt.agnostic_nms, scale=True)
        return [det], [self._labelled(det.tolist())]

    @torch.no_grad()
    def detect_frames_enqueue(self, frames):
        """First half of ``detect_frames`` for callers that keep several passes in flight (the folder drivers): the batched
        pass, NMS and an asynchronous copy of every frame's box rows and counts into page-locked host memory of THIS call, all
        on the current stream, no host sync.  The plan's device buffers are shared by every pass of that shape, so a later
        pass on the same stream may overwrite them; the host copy is what ``detect_frames_finish`` reads."""
        opt = self.opt
        p = self.engine.forward(list(frames))
        self.engine.nms_enqueue(p, opt.conf_thres, opt.iou_thres, opt.classes, opt.agnostic_nms, scale=True)
        nb = p["nb"]
        counts = torch.empty(nb, dtype=torch.int32, pin_memory=True)
        rows = torch.empty(nb * 300, 6, dtype=torch.float32, pin_memory=True)
        counts.copy_(p["count"], non_blocking=True)
        rows.copy_(p["dets"], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return counts, rows, ev, nb

    def detect_frames_finish(self, token):
        """Second half: wait for that pass (its event only) and return (preds, dets_lists) as ``detect_frames`` does, the
        prediction rows as host tensors."""
        counts, rows, ev, nb = token
        ev.synchronize()
        k = counts.tolist()
        rows = rows.reshape(nb, 300, 6)
        return ([rows[i, :int(n)].clone() for i, n in enumerate(k)],
                [self._labelled(rows[i, :int(n)].tolist()) for i, n in enumerate(k)])

    @torch.no_grad()
    def detect_frames(self, frames):
        """``detect`` for a list of equally sized (H,W,3) uint8 BGR DEVICE frames: one batched pass through the network, one
        NMS launch per frame, ONE host transfer for all box lists.  Returns (preds, dets_lists) with one entry per frame, each
        as ``detect`` returns it for a single image."""
        opt = self.opt
        p = self.engine.forward(list(frames))
        self.engine.nms_enqueue(p, opt.conf_thres, opt.iou_thres, opt.classes, opt.agnostic_nms, scale=True)
        counts = p["count"].tolist()                                   # the only host sync of the chunk
        dets = p["dets"].reshape(p["nb"], 300, 6)
        preds = [dets[i, :int(k)].clone() for i, k in enumerate(counts)]
        host = dets.cpu()
        return preds, [self._labelled(host[i, :int(k)].tolist()) for i, k in enumerate(counts)]
