#!/usr/bin/env python3
"""Generate tests/golden/yolo_tiny_ckpt.pt + .npz: a checkpoint written by the REFERENCE's own classes
(``Model`` from yolo/yolov7/models/yolo.py on a 10-layer yaml that uses the module types of the deploy graph:
Conv, Concat, MP, SPPCSPC, RepConv, nn.Upsample, IDetect), saved the way yolov7 training saves it
(``{'model': model.half(), 'ema': ..., ...}``), and the state dict / names / nc it must yield.  The .pt is data:
tests load it with hamer_yolo_amd.utils.checkpoint (no reference classes exist at test time)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_golden_yolo import load_reference, OUT  # noqa: E402

CFG = {
    "nc": 3, "depth_multiple": 1.0, "width_multiple": 1.0,
    "anchors": [[12, 16, 19, 36, 40, 28], [36, 75, 76, 55, 72, 146], [142, 110, 192, 243, 459, 401]],
    "backbone": [
        [-1, 1, "Conv", [8, 3, 1]],            # 0
        [-1, 1, "Conv", [16, 3, 2]],           # 1
        [-1, 1, "Conv", [8, 1, 1]],            # 2
        [-2, 1, "Conv", [8, 1, 1]],            # 3
        [[-1, -2], 1, "Concat", [1]],          # 4
        [-1, 1, "MP", []],                     # 5
        [-1, 1, "SPPCSPC", [16]],              # 6
    ],
    "head": [
        [-1, 1, "RepConv", [16, 3, 1]],        # 7
        [-1, 1, "nn.Upsample", [None, 2, "nearest"]],   # 8
        [[4, 8, 7], 1, "IDetect", ["nc", "anchors"]],   # 9
    ],
}


def main():
    Model, _, _ = load_reference()
    torch.manual_seed(0)
    model = Model(CFG, ch=3, nc=3)
    g = torch.Generator().manual_seed(1)
    for p in model.parameters():
        p.data = torch.randn(p.shape, generator=g) * 0.1
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean = torch.randn(m.running_mean.shape, generator=g) * 0.1
            m.running_var = torch.rand(m.running_var.shape, generator=g) + 0.5
    model.names = ["left", "right", "other"]
    ema = None
    ck = {"epoch": 7, "best_fitness": 0.5, "model": model.half(), "ema": ema, "updates": 0, "optimizer": None,
          "training_results": None, "wandb_id": None}
    path = os.path.join(OUT, "yolo_tiny_ckpt.pt")
    torch.save(ck, path)
    sd = {k: v.float().numpy() for k, v in model.state_dict().items() if v.is_floating_point()}
    np.savez_compressed(os.path.join(OUT, "yolo_tiny_ckpt.npz"), names=np.array(model.names), nc=3,
                        keys=np.array(list(sd.keys())), **{f"t{i}": v for i, v in enumerate(sd.values())})
    print(path, os.path.getsize(path), "bytes;", len(sd), "tensors; classes pickled from", type(model).__module__)


if __name__ == "__main__":
    main()
