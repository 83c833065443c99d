#!/usr/bin/env python3
"""Generate tests/golden/rootnet_head.npz from the REFERENCE's RootNet code (read in place from /root/reference; build
container only): ``process_bbox`` / ``sanitize_bbox`` (rootnet/preprocessing.py:152-188), ``ResRootNet.forward``
(rootnet/Model_RGB.py:240-292) and ``EstimateRGB.calculate_k`` (:494-498) on seeded inputs.
Import-time stand-ins, names only and off the arithmetic path: cv2, plyfile, torchvision(.models/.transforms), and the
sibling modules Model_RGB.py imports but the depth head never touches (convnext, vis_tool, mano).  The ResNet-34 backbone
itself is torchvision's and cannot be captured here (torchvision is not installed)."""
import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("HAMER_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def load_reference():
    for name in ("cv2", "plyfile", "torchvision", "torchvision.models", "torchvision.transforms"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = None
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    pkg = types.ModuleType("refrootnet")
    pkg.__path__ = [os.path.join(REF, "rootnet")]
    sys.modules["refrootnet"] = pkg
    for sib, names in (("convnext", ["convnext_base"]), ("vis_tool", ["draw_pose", "draw_2d_skeleton"]), ("mano", ["MANO"])):
        m = types.ModuleType("refrootnet." + sib)
        for n in names:
            setattr(m, n, None)
        sys.modules["refrootnet." + sib] = m
    prep = importlib.import_module("refrootnet.preprocessing")
    model = importlib.import_module("refrootnet.Model_RGB")
    return prep, model


def main():
    prep, model = load_reference()
    rng = np.random.RandomState(0)
    boxes = np.array([[200.0, 150.0, 100.0, 60.0], [-20.0, 400.0, 100.0, 200.0], [600.0, 10.0, 80.0, 300.0], [0.0, 0.0, 640.0, 480.0],
                      [300.5, 200.25, 33.0, 47.5], [630.0, 470.0, 50.0, 50.0], [10.0, 10.0, 0.0, 50.0]])
    W, H = 640, 480
    proc = []
    for b in boxes:
        r = prep.process_bbox(b.copy(), W, H, (256, 256), 1.5)
        proc.append(np.full(4, np.nan, np.float32) if r is None else np.asarray(r, np.float32))
    torch.manual_seed(0)
    net = model.ResRootNet(inplanes=512).eval()
    feats = torch.randn(5, 512, 8, 8)
    kval = torch.rand(5) * 3 + 0.2
    with torch.no_grad():
        depth = net(feats, kval)
    self_like = types.SimpleNamespace(cfg=types.SimpleNamespace(bbox_real=(0.3, 0.3), device="cpu"))
    ks = [float(model.EstimateRGB.calculate_k(self_like, torch.tensor(p), 900.0, 880.0)[0]) for p in proc if not np.isnan(p[0])]
    np.savez_compressed(os.path.join(OUT, "rootnet_head.npz"), boxes=boxes, img_wh=np.array([W, H]), processed=np.stack(proc),
                        feats=feats.numpy(), k_value=kval.numpy(), depth_w=net.depth_layer.weight.detach().numpy(),
                        depth_b=net.depth_layer.bias.detach().numpy(), depth=depth.numpy(), k_of_processed=np.array(ks, np.float32),
                        fx_fy=np.array([900.0, 880.0]))
    print("processed:\n", np.stack(proc), "\ndepth:", depth.reshape(-1).numpy(), "\nk:", ks)


if __name__ == "__main__":
    main()
