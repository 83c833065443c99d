"""YOLOv7 graph (reference: yolo/yolov7/cfg/deploy/yolov7.yaml:1-140 parsed by ``parse_model``,
models/yolo.py:744-839), built here from its three repeating blocks instead of a yaml file.

Every entry is ``(from, kind, args)`` with the reference's layer indices (0..105) so that weights
keyed ``model.<i>.*`` in a YOLOv7 checkpoint map one to one:
  ("conv", (c_out, k, s))  Conv = Conv2d(bias=False)+BN+SiLU, fused at load time (common.py:103-115)
  ("mp", ())               MaxPool2d(2, 2)                                  (common.py:34-40)
  ("concat", ())           channel concat                                   (common.py:60-66)
  ("up", ())               nn.Upsample(scale 2, nearest)
  ("sppcspc", (c_out,))    SPPCSPC                                          (common.py:266-284)
  ("repconv", (c_out,))    RepConv 3x3 (re-parameterised)                   (common.py:467-504)
  ("detect", ())           Detect / IDetect head                            (yolo.py:31-85,:105-184)
"""
from __future__ import annotations

from typing import Dict, List, Tuple

ANCHORS = [[12, 16, 19, 36, 40, 28], [36, 75, 76, 55, 72, 146], [142, 110, 192, 243, 459, 401]]
STRIDES = [8, 16, 32]


def yolov7_layers() -> List[Tuple[object, str, tuple]]:
    L: List[Tuple[object, str, tuple]] = []

    def add(frm, kind, args=()):
        L.append((frm, kind, args))
        return len(L) - 1

    def elan(c_mid, c_out):           # backbone E-ELAN: 2 x 1x1, 4 x 3x3, concat 4, 1x1
        add(-1, "conv", (c_mid, 1, 1)); add(-2, "conv", (c_mid, 1, 1))
        for _ in range(4):
            add(-1, "conv", (c_mid, 3, 1))
        add([-1, -3, -5, -6], "concat")
        return add(-1, "conv", (c_out, 1, 1))

    def down(c):                      # MP branch + strided-conv branch
        add(-1, "mp"); add(-1, "conv", (c, 1, 1)); add(-3, "conv", (c, 1, 1)); add(-1, "conv", (c, 3, 2))

    def elan_head(c_mid, c_out):      # head E-ELAN: 2 x 1x1, 4 x 3x3 (half width), concat 6, 1x1
        add(-1, "conv", (c_mid, 1, 1)); add(-2, "conv", (c_mid, 1, 1))
        for _ in range(4):
            add(-1, "conv", (c_mid // 2, 3, 1))
        add([-1, -2, -3, -4, -5, -6], "concat")
        return add(-1, "conv", (c_out, 1, 1))

    add(-1, "conv", (32, 3, 1)); add(-1, "conv", (64, 3, 2)); add(-1, "conv", (64, 3, 1)); add(-1, "conv", (128, 3, 2))
    elan(64, 256)                                   # 4..11
    down(128); add([-1, -3], "concat"); p3 = elan(128, 512)      # 12..24
    down(256); add([-1, -3], "concat"); p4 = elan(256, 1024)     # 25..37
    down(512); add([-1, -3], "concat"); elan(256, 1024)          # 38..50
    spp = add(-1, "sppcspc", (512,))                              # 51
    add(-1, "conv", (256, 1, 1)); add(-1, "up"); add(p4, "conv", (256, 1, 1)); add([-1, -2], "concat")
    n63 = elan_head(256, 256)                                     # 56..63
    add(-1, "conv", (128, 1, 1)); add(-1, "up"); add(p3, "conv", (128, 1, 1)); add([-1, -2], "concat")
    n75 = elan_head(128, 128)                                     # 68..75
    down(128); add([-1, -3, n63], "concat"); n88 = elan_head(256, 256)    # 76..88
    down(256); add([-1, -3, spp], "concat"); n101 = elan_head(512, 512)   # 89..101
    a = add(n75, "repconv", (256,)); b = add(n88, "repconv", (512,)); c = add(n101, "repconv", (1024,))
    add([a, b, c], "detect")
    assert len(L) == 106 and (p3, p4, spp, n63, n75, n88, n101) == (24, 37, 51, 63, 75, 88, 101)
    return L


def resolve(layers) -> List[Tuple[List[int], str, tuple]]:
    """Absolute source indices for every layer (``-1`` on layer 0 is the image, index -1)."""
    out = []
    for i, (frm, kind, args) in enumerate(layers):
        srcs = frm if isinstance(frm, list) else [frm]
        out.append(([s if s >= 0 else i + s for s in srcs], kind, args))
    return out


def channels(layers, c_in: int = 3, nc: int = 3) -> List[int]:
    """Output channels per layer."""
    ch: List[int] = []
    for srcs, kind, args in resolve(layers):
        cin = [c_in if s < 0 else ch[s] for s in srcs]
        if kind in ("conv", "sppcspc", "repconv"):
            ch.append(args[0])
        elif kind in ("mp", "up"):
            ch.append(cin[0])
        elif kind == "concat":
            ch.append(sum(cin))
        elif kind == "detect":
            ch.append(3 * (5 + nc))
    return ch


def conv_specs(layers, c_in: int = 3, nc: int = 3) -> Dict[str, Tuple[int, int, int, int]]:
    """name -> (c_out, c_in, k, s) of every fused convolution, keyed like the fused reference model
    (``model.<i>.conv``, ``model.51.cv<j>.conv``, ``model.<i>.rbr_reparam``, ``model.105.m.<l>``)."""
    ch = channels(layers, c_in, nc)
    specs: Dict[str, Tuple[int, int, int, int]] = {}
    for i, (srcs, kind, args) in enumerate(resolve(layers)):
        cin = c_in if srcs[0] < 0 else ch[srcs[0]]
        if kind == "conv":
            specs[f"model.{i}.conv"] = (args[0], cin, args[1], args[2])
        elif kind == "repconv":
            specs[f"model.{i}.rbr_reparam"] = (args[0], cin, 3, 1)
        elif kind == "sppcspc":
            c_ = args[0]          # int(2 * c2 * 0.5)
            for name, (co, ci, k) in {"cv1": (c_, cin, 1), "cv2": (c_, cin, 1), "cv3": (c_, c_, 3), "cv4": (c_, c_, 1),
                                      "cv5": (c_, 4 * c_, 1), "cv6": (c_, c_, 3), "cv7": (args[0], 2 * c_, 1)}.items():
                specs[f"model.{i}.{name}.conv"] = (co, ci, k, 1)
        elif kind == "detect":
            for l, s in enumerate(srcs):
                specs[f"model.{i}.m.{l}"] = (3 * (5 + nc), ch[s], 1, 1)
    return specs
