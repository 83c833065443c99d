"""ORACLE (test infrastructure): CPU restatement of the detector path ``Detector.detect``
(yolo/detector.py:106-153): letterbox, the fused YOLOv7 forward, Detect decode,
non_max_suppression, scale_coords -- fp32 PyTorch / numpy on the CPU.

Pinning (tools/gen_golden_yolo.py, build container): the model forward, the load-time weight
folding and ``non_max_suppression`` / ``scale_coords`` are checked against the reference's own
``models/yolo.py`` ``Model`` + ``TracedModel`` and ``utils/general.py`` on seeded weights; the
outputs are committed under tests/golden/yolo_*.npz.
PARITY UNPINNED for two third-party calls absent from /root/reference and from this image:
``torchvision.ops.nms`` (restated as greedy score-sorted NMS, suppress IoU > thr, the published
algorithm) and ``cv2.resize`` / ``cv2.copyMakeBorder`` inside ``letterbox`` (restated from the
published 8-bit INTER_LINEAR algorithm: half-pixel centres, 11-bit fixed-point coefficients,
two-pass rounding).  Geometry of letterbox is pinned by SURVEY 8a's hand-derived known answers.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------- letterbox (A2)
def resize_linear_u8(img: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv::resize, 8-bit, INTER_LINEAR (classic fixed-point path): src = (dst + 0.5) * scale - 0.5,
    coefficients rounded to 1/2048, horizontal pass in int32, vertical pass
    ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2."""
    sh, sw = img.shape[:2]
    COEF = 2048

    def taps(dn, sn):
        scale = 1.0 / (dn / sn)
        d = np.arange(dn)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        fr = (f - s).astype(np.float32)
        lo = s < 0
        s[lo], fr[lo] = 0, 0.0
        hi = s >= sn - 1
        s[hi], fr[hi] = sn - 1, 0.0
        a1 = np.rint(fr * COEF).astype(np.int64)           # saturate_cast<short>(fr * 2048)
        a0 = np.rint((np.float32(1.0) - fr) * COEF).astype(np.int64)
        return s, np.minimum(s + 1, sn - 1), a0, a1

    x0, x1, ax0, ax1 = taps(dw, sw)
    y0, y1, ay0, ay1 = taps(dh, sh)
    src = img.astype(np.int64)
    rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]       # (sh, dw, c)
    r0, r1 = rows[y0], rows[y1]
    out = (((ay0[:, None, None] * (r0 >> 4)) >> 16) + ((ay1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox_geometry(h: int, w: int, new_shape: int = 640, stride: int = 32) -> Dict[str, float]:
    """utils/datasets.py:999-1029 with auto=True, scaleup=True: returns new_unpad (w,h), pads, out size."""
    r = min(new_shape / h, new_shape / w)
    nw, nh = int(round(w * r)), int(round(h * r))
    dw, dh = new_shape - nw, new_shape - nh
    dw, dh = float(np.mod(dw, stride)) / 2, float(np.mod(dh, stride)) / 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return {"r": r, "nw": nw, "nh": nh, "top": top, "bottom": bottom, "left": left, "right": right,
            "out_h": nh + top + bottom, "out_w": nw + left + right, "dw": dw, "dh": dh}


def letterbox(img_bgr: np.ndarray, new_shape: int = 640, stride: int = 32, color: int = 114) -> Tuple[np.ndarray, Dict]:
    """letterbox + LoadImage.process_img (datasets.py:137-141): returns CHW RGB uint8."""
    h, w = img_bgr.shape[:2]
    g = letterbox_geometry(h, w, new_shape, stride)
    im = img_bgr if (w, h) == (g["nw"], g["nh"]) else resize_linear_u8(img_bgr, g["nw"], g["nh"])
    out = np.full((g["out_h"], g["out_w"], 3), color, dtype=np.uint8)
    out[g["top"]:g["top"] + g["nh"], g["left"]:g["left"] + g["nw"]] = im
    return np.ascontiguousarray(out[:, :, ::-1].transpose(2, 0, 1)), g


# ----------------------------------------------------------------------------- fused forward (A3, A4)
def _h(t: Tensor, emu) -> Tensor:
    """emu == "fp16": round to half where hm_conv2d_nhwc stores half (weights at load time, every activation tensor)."""
    return t.half().float() if emu else t


def _conv(x: Tensor, wb: Tuple[Tensor, Tensor], k: int, s: int, act: bool = True, emu=False, out_f32: bool = False) -> Tensor:
    y = F.conv2d(x, _h(wb[0], emu), wb[1], stride=s, padding=k // 2)      # fp32 accumulation, fp32 bias
    y = F.silu(y) if act else y
    return y if out_f32 else _h(y, emu)


def layer_forward(layers, fused: Dict[str, Tuple[Tensor, Tensor]], i: int, inp: List[Tensor], nc: int = 3, emu=False):
    """One layer of Model.forward_once (yolo.py:609-639) on explicit inputs (what the layer's sources produced).  The detect
    layer returns its three raw maps (B, 3, ny, nx, 5+nc) before the sigmoid."""
    _, kind, args = layers[i]
    if kind == "conv":
        return _conv(inp[0], fused[f"model.{i}.conv"], args[1], args[2], emu=emu)
    if kind == "repconv":
        return _conv(inp[0], fused[f"model.{i}.rbr_reparam"], 3, 1, emu=emu)
    if kind == "mp":
        return F.max_pool2d(inp[0], 2, 2)
    if kind == "up":
        return F.interpolate(inp[0], scale_factor=2, mode="nearest")
    if kind == "concat":
        return torch.cat(inp, 1)
    if kind == "sppcspc":                                        # common.py:279-284
        cv = lambda j, t, k: _conv(t, fused[f"model.{i}.cv{j}.conv"], k, 1, emu=emu)
        x1 = cv(4, cv(3, cv(1, inp[0], 1), 3), 1)
        y1 = cv(6, cv(5, torch.cat([x1] + [F.max_pool2d(x1, k, 1, k // 2) for k in (5, 9, 13)], 1), 1), 3)
        return cv(7, torch.cat((y1, cv(2, inp[0], 1)), dim=1), 1)
    if kind == "detect":
        raws, no = [], nc + 5
        for l, t in enumerate(inp):
            r = _conv(t, fused[f"model.{i}.m.{l}"], 1, 1, act=False, emu=emu, out_f32=True)
            bs, _, ny, nx = r.shape
            raws.append(r.view(bs, 3, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous())
        return raws
    raise ValueError(kind)


def yolo_forward(layers, fused: Dict[str, Tuple[Tensor, Tensor]], x: Tensor, nc: int = 3,
                 anchors: Sequence[Sequence[int]] = (), strides: Sequence[int] = (8, 16, 32), emu=False) -> Tuple[Tensor, List[Tensor]]:
    """Model.forward_once over the fused graph (yolo.py:609-639) + IDetect.fuseforward (yolo.py:148-184).
    x: (B,3,H,W) in [0,1].  Returns (pred (B, sum(3*ny*nx), 5+nc), the three raw head maps).
    ``emu="fp16"``: the arithmetic of the HIP path (reference GPU branch, detector.py:110-112 ``half()``): half weights and
    activations, fp32 accumulation and bias, the three head maps and the decode in fp32."""
    ys: List[Optional[Tensor]] = []
    x = _h(x, emu)
    for i, (frm, kind, args) in enumerate(layers):
        srcs = frm if isinstance(frm, list) else [frm]
        inp = [x if (s == -1 and i == 0) else ys[s if s >= 0 else i + s] for s in srcs]
        y = layer_forward(layers, fused, i, inp, nc, emu)
        if kind == "detect":
            z, no = [], nc + 5
            for l, r in enumerate(y):
                bs, _, ny, nx, _ = r.shape
                yv, xv = torch.meshgrid([torch.arange(ny), torch.arange(nx)], indexing="ij")
                grid = torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2).float()
                ag = torch.tensor(anchors[l]).float().view(1, 3, 1, 1, 2)
                yy = r.sigmoid()
                yy[..., 0:2] = (yy[..., 0:2] * 2. - 0.5 + grid) * strides[l]
                yy[..., 2:4] = (yy[..., 2:4] * 2) ** 2 * ag
                z.append(yy.view(bs, -1, no))
            return torch.cat(z, 1), y
        ys.append(y)
    raise ValueError("graph has no detect layer")


# ----------------------------------------------------------------------------- NMS (A6) and scaling (A7)
def nms_greedy(boxes: Tensor, scores: Tensor, iou_thres: float) -> Tensor:
    """torchvision.ops.nms restated (published algorithm): visit boxes by descending score, keep a box
    unless its IoU with an already kept box is > iou_thres.  Returns kept indices in score order."""
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order].numpy().astype(np.float32)
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    n = len(b)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        xx1 = np.maximum(b[i, 0], b[i + 1:, 0]); yy1 = np.maximum(b[i, 1], b[i + 1:, 1])
        xx2 = np.minimum(b[i, 2], b[i + 1:, 2]); yy2 = np.minimum(b[i, 3], b[i + 1:, 3])
        w = np.maximum(np.float32(0), xx2 - xx1); h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(invalid="ignore", divide="ignore"):
            ovr = inter / (areas[i] + areas[i + 1:] - inter)      # 0/0 -> nan -> not suppressed, as in C++
        suppressed[i + 1:] |= ovr > np.float32(iou_thres)
    return order[torch.tensor(keep, dtype=torch.long)]


def xywh2xyxy(x: Tensor) -> Tensor:
    y = x.clone()
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def non_max_suppression(prediction: Tensor, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                        max_det: int = 300) -> List[Tensor]:
    """utils/general.py:611-703 (best-class branch, no labels, no merge)."""
    max_wh, max_nms = 4096, 30000
    xc = prediction[..., 4] > conf_thres
    output = [torch.zeros((0, 6))] * prediction.shape[0]
    for xi, x in enumerate(prediction):
        x = x[xc[xi]].clone()
        if not x.shape[0]:
            continue
        x[:, 5:] *= x[:, 4:5]
        box = xywh2xyxy(x[:, :4])
        conf, j = x[:, 5:].max(1, keepdim=True)
        x = torch.cat((box, conf, j.float()), 1)[conf.view(-1) > conf_thres]
        if classes is not None:
            x = x[(x[:, 5:6] == torch.tensor(classes)).any(1)]
        if not x.shape[0]:
            continue
        if x.shape[0] > max_nms:                                   # general.py:679-680 (stable: equal scores keep row order)
            x = x[torch.argsort(x[:, 4], descending=True, stable=True)[:max_nms]]
        c = x[:, 5:6] * (0 if agnostic else max_wh)
        i = nms_greedy(x[:, :4] + c, x[:, 4], iou_thres)[:max_det]
        output[xi] = x[i]
    return output


def scale_coords(img1_shape, coords: Tensor, img0_shape) -> Tensor:
    """utils/general.py:323-344 (ratio_pad None), in place."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    coords[:, 0].clamp_(0, img0_shape[1]); coords[:, 1].clamp_(0, img0_shape[0])
    coords[:, 2].clamp_(0, img0_shape[1]); coords[:, 3].clamp_(0, img0_shape[0])
    return coords


def detect(layers, fused, img_bgr: np.ndarray, nc=3, anchors=(), conf_thres=0.25, iou_thres=0.35,
           classes=(0, 1, 2), agnostic=True, emu=False):
    """Detector.detect, yolo/detector.py:106-153 (CPU branch: fp32; ``emu="fp16"``: the GPU branch's half arithmetic)."""
    chw, g = letterbox(img_bgr)
    x = torch.from_numpy(chw).float() / 255.0
    pred, _ = yolo_forward(layers, fused, x[None], nc, anchors, emu=emu)
    dets = non_max_suppression(pred, conf_thres, iou_thres, list(classes), agnostic)
    out_list = []
    for det in dets:
        d = []
        if len(det):
            det[:, :4] = scale_coords(x.shape[1:], det[:, :4], img_bgr.shape).round()
            for row in det.tolist():
                d.append(['right' if row[-1] == 1 else 'left', row[:4]])
        out_list.append(d)
    return dets, out_list, pred
