#!/usr/bin/env python3
"""Generate tests/golden/yolo_*.npz from the REFERENCE's YOLOv7 code (read in place from
/root/reference; build container only).

Loaded from the reference: ``yolo/yolov7/models/yolo.py`` ``Model`` (graph from
``cfg/training/yolov7.yaml``, IDetect head), ``utils/torch_utils.py`` ``TracedModel`` /
``fuse_conv_and_bn``, ``utils/general.py`` ``non_max_suppression`` / ``scale_coords``.
Import-time stand-ins (names only, nothing on the arithmetic path except ``torchvision.ops.nms``,
see below): ``cv2`` (setNumThreads), ``torchvision`` package tree, ``seaborn``.
``torchvision.ops.nms`` is absent from this image; the reference's ``non_max_suppression`` is run
with the oracle's greedy NMS plugged into that one call, so the golden pins everything around it
(candidate filter, conf = obj*cls, best class, class filter, xywh->xyxy, cap) but not nms itself.
"""
import os
import sys
import types

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("HAMER_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

from hamer_yolo_amd import synth  # noqa: E402
from hamer_yolo_amd.yolo import arch, fuse  # noqa: E402
from oracle import yolo_ref  # noqa: E402


def load_reference():
    cv2 = types.ModuleType("cv2"); cv2.setNumThreads = lambda n: None
    sys.modules["cv2"] = cv2
    for name in ("torchvision", "torchvision.ops", "torchvision.utils", "torchvision.models", "torchvision.transforms", "seaborn"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].ops = sys.modules["torchvision.ops"]
    sys.modules["torchvision.ops"].nms = lambda boxes, scores, thr: yolo_ref.nms_greedy(boxes, scores, thr)
    for nm in ("DeformConv2d", "roi_pool", "roi_align", "ps_roi_pool", "ps_roi_align"):   # imported, never called here
        setattr(sys.modules["torchvision.ops"], nm, None)
    sys.modules["torchvision.utils"].save_image = None
    # Both roots, as in the reference's own deployment: models/yolo.py imports ``yolov7.models.common``
    # (yolo.py:17) while common.py / experimental.py import ``yolo.yolov7...`` (common.py:20).  Side effect in the
    # reference itself: two distinct ``Conv`` classes exist, so Model.fuse()'s ``type(m) is Conv`` (yolo.py:710)
    # skips the 7 convolutions inside SPPCSPC, which keep running as conv -> BatchNorm(eval) -> SiLU.  Same
    # function, different state-dict keys; the build folds them too (fuse.py) and the forward golden covers it.
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "yolo"))
    from yolo.yolov7.models.yolo import Model
    from yolo.yolov7.utils import general
    from yolo.yolov7.utils.torch_utils import TracedModel
    return Model, TracedModel, general


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    Model, TracedModel, general = load_reference()
    nc = 3
    with open(os.path.join(REF, "yolo/yolov7/cfg/training/yolov7.yaml")) as f:
        cfg = yaml.safe_load(f)
    model = Model(cfg, ch=3, nc=nc)
    sd = synth.yolo_state_dict(seed=0, nc=nc)
    own = model.state_dict()
    missing = [k for k in own if k not in sd and not k.endswith("num_batches_tracked") and "anchor" not in k]
    extra = [k for k in sd if k not in own]
    assert not missing and not extra, (missing[:5], extra[:5])
    model.load_state_dict(sd, strict=False)
    for p in model.parameters():
        p.requires_grad_(False)          # IDetect.fuse updates biases in place (yolo.py:192)
    layers = arch.yolov7_layers()
    specs = arch.conv_specs(layers, 3, nc)
    assert [list(map(float, a)) for a in cfg["anchors"]] == [list(map(float, a)) for a in arch.ANCHORS]
    assert [int(s) for s in model.stride.tolist()] == arch.STRIDES

    # ---- A8: load-time folding (reference Model.fuse vs the build's fuse_state_dict)
    model = model.float().fuse().eval()
    fsd = model.state_dict()
    mine = fuse.fuse_state_dict(sd, specs)
    worst = 0.0
    for name, (w, b) in mine.items():
        if name + ".bias" not in fsd:            # SPPCSPC inner convs: left unfused by the reference (see above)
            assert name.startswith("model.51.cv")
            continue
        worst = max(worst, float((fsd[name + ".weight"] - w).abs().max()), float((fsd[name + ".bias"] - b).abs().max()))
    print("fuse: max |dw|,|db| vs reference Model.fuse():", worst)
    assert worst < 1e-5
    picks = ["model.0.conv", "model.37.conv", "model.50.conv", "model.103.rbr_reparam", "model.105.m.1"]
    np.savez_compressed(os.path.join(OUT, "yolo_fuse.npz"), seed=0, nc=nc,
                        **{f"{n}.bias": fsd[n + ".bias"].numpy() for n in picks},
                        **{f"{n}.wsub": fsd[n + ".weight"].flatten()[::97].numpy() for n in picks})

    # ---- A3/A4: fused forward through TracedModel on a seeded letterboxed input
    traced = TracedModel(model, torch.device("cpu"), 640)
    x = synth.frame_u8(384, 640, seed=2).permute(2, 0, 1).float()[None] / 255.0      # (1,3,384,640)
    with torch.no_grad():
        pred = traced(x)[0]
        mine_pred, _ = yolo_ref.yolo_forward(layers, {k: v for k, v in mine.items()}, x, nc, arch.ANCHORS, arch.STRIDES)
    print("pred", tuple(pred.shape), "oracle vs reference max abs:", float((pred - mine_pred).abs().max()),
          "n(obj>0.25):", int((pred[..., 4] > 0.25).sum()))
    assert pred.shape == (1, 15120, 8)
    assert torch.allclose(pred, mine_pred, rtol=1e-4, atol=1e-4)      # box sizes reach ~2e3 px: relative bound

    # ---- A6/A7 on the reference prediction
    dets = general.non_max_suppression(pred.clone(), 0.25, 0.35, classes=[0, 1, 2], agnostic=True)[0]
    dets_cls = general.non_max_suppression(pred.clone(), 0.25, 0.35, classes=[1], agnostic=False)[0]
    print("nms kept", len(dets), "class-1 only, non-agnostic:", len(dets_cls))
    boxes = synth.uniform("golden.boxes", (16, 4), 300.0, 320.0, seed=1)
    sc = general.scale_coords((384, 640), boxes.clone(), (1080, 1920, 3))
    sc2 = general.scale_coords((448, 640), boxes.clone(), (565, 848, 3))
    np.savez_compressed(os.path.join(OUT, "yolo_forward.npz"), seed=0, nc=nc, frame_seed=2,
                        pred_rows=pred[0, ::9].numpy(), pred_sum=pred[0].double().sum(0).numpy(),
                        pred_cand=pred[0][pred[0, :, 4] > 0.25].numpy(),
                        dets=dets.numpy(), dets_cls1=dets_cls.numpy(),
                        boxes=boxes.numpy(), scaled_1080=sc.numpy(), scaled_565=sc2.numpy())


if __name__ == "__main__":
    main()
