#!/usr/bin/env python3
"""Compute synth.YOLO_CALIB: per-convolution weight-width factors (3 decimals) such that, layer by
layer, the pre-activation rms of the fused convolution is ~1 on the seeded calibration frame
(detect head: ~2).  Prints the dict literal to paste into hamer_yolo_amd/synth.py."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamer_yolo_amd import synth  # noqa: E402
from hamer_yolo_amd.yolo import arch, fuse  # noqa: E402


def main():
    torch.set_num_threads(8)
    layers = arch.yolov7_layers()
    specs = arch.conv_specs(layers, 3, 3)
    synth.YOLO_CALIB.clear()
    x = synth.frame_u8(384, 640, seed=2).permute(2, 0, 1).float()[None] / 255.0
    calib = {}

    def run_conv(name, inp, k, s, target=1.0, act=True):
        synth.YOLO_CALIB.update(calib)
        sd = synth.yolo_state_dict(seed=0, nc=3)
        w, b = fuse.fuse_state_dict(sd, {name: specs[name]})[name]
        pre = F.conv2d(inp, w, None, stride=s, padding=k // 2)
        f = round(float(target / pre.pow(2).mean().sqrt()), 3)
        calib[name] = f
        y = F.conv2d(inp, w * f, b, stride=s, padding=k // 2)
        return F.silu(y) if act else y

    ys = []
    with torch.no_grad():
        for i, (frm, kind, args) in enumerate(layers):
            srcs = frm if isinstance(frm, list) else [frm]
            inp = [x if (s == -1 and i == 0) else ys[s if s >= 0 else i + s] for s in srcs]
            if kind == "conv":
                o = run_conv(f"model.{i}.conv", inp[0], args[1], args[2])
            elif kind == "repconv":
                o = run_conv(f"model.{i}.rbr_reparam", inp[0], 3, 1)
            elif kind == "mp":
                o = F.max_pool2d(inp[0], 2, 2)
            elif kind == "up":
                o = F.interpolate(inp[0], scale_factor=2, mode="nearest")
            elif kind == "concat":
                o = torch.cat(inp, 1)
            elif kind == "sppcspc":
                cv = lambda j, t, k: run_conv(f"model.{i}.cv{j}.conv", t, k, 1)
                x1 = cv(4, cv(3, cv(1, inp[0], 1), 3), 1)
                y1 = cv(6, cv(5, torch.cat([x1] + [F.max_pool2d(x1, k, 1, k // 2) for k in (5, 9, 13)], 1), 1), 3)
                o = cv(7, torch.cat((y1, cv(2, inp[0], 1)), 1), 1)
            elif kind == "detect":
                for l, t in enumerate(inp):
                    run_conv(f"model.{i}.m.{l}", t, 1, 1, target=2.0, act=False)
                break
            ys.append(o)
            print(i, kind, float(o.pow(2).mean().sqrt()), file=sys.stderr)
    print("YOLO_CALIB: Dict[str, float] = {")
    items = list(calib.items())
    for j in range(0, len(items), 4):
        print("    " + " ".join(f'"{k}": {v},' for k, v in items[j:j + 4]))
    print("}")


if __name__ == "__main__":
    main()
