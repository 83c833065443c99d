"""YoloEngine: device-resident fused YOLOv7 weights + a planned op list per input size.

Load time (host, once): BN / RepConv / implicit folding (fuse.py), weights re-laid out as
[Cout][ky][kx][Cin] 16-bit rows padded to a multiple of 64 (the implicit-GEMM K axis).
Plan time (host, once per letterboxed size): every tensor gets a home in an NHWC arena; a tensor
that feeds a Concat lives directly in a channel slice of the concat's buffer, so Concat costs
nothing (reference: Model.forward_once, yolo.py:609-639, copies on every torch.cat).
Run time: hm_letterbox -> hm_yolo_run (one enqueue for ~110 kernels) -> 3 x hm_yolo_decode -> hm_yolo_nms.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import lib as L
from . import arch, fuse


def _kpad(k: int) -> int:
    return (k + 63) // 64 * 64


def detect_anchors(state_dict: Dict[str, torch.Tensor]) -> List[List[float]]:
    """Anchor sizes in pixels per detect level, [3][w0, h0, w1, h1, w2, h2].  IDetect decodes with its ``anchor_grid``
    buffer (yolo.py:164), which travels in the checkpoint and may have been rewritten by autoanchor for a custom-trained
    detector, so the checkpoint's values win: ``*.anchor_grid`` (pixels), else ``*.anchors`` (in stride units, yolo.py:
    537-539) times the strides, else the yolov7.yaml defaults."""
    grid = [v for k, v in state_dict.items() if k.endswith(".anchor_grid")]
    unit = [v for k, v in state_dict.items() if k.endswith(".anchors")]
    if grid:
        a = grid[0].detach().float().cpu().reshape(-1, 6)
    elif unit:
        a = unit[0].detach().float().cpu().reshape(-1, 3, 2) * torch.tensor(arch.STRIDES, dtype=torch.float32).view(-1, 1, 1)
        a = a.reshape(-1, 6)
    else:
        return [[float(v) for v in lvl] for lvl in arch.ANCHORS]
    if a.shape != (len(arch.STRIDES), 6) or not bool(torch.isfinite(a).all()) or not bool((a > 0).all()):
        raise ValueError(f"detect head anchors: expected {len(arch.STRIDES)} levels x 3 positive (w, h) pairs, got shape {tuple(a.shape)}")
    return [[float(v) for v in lvl] for lvl in a]


class YoloEngine:
    def __init__(self, state_dict: Dict[str, torch.Tensor], nc: int = 3, device="cuda", dtype=torch.float16,
                 new_shape: int = 640, stride: int = 32, names: Optional[List[str]] = None):
        if not torch.cuda.is_available():
            raise L.HipLibraryError("YoloEngine needs an MI355X (HIP device); there is no CPU fallback")
        self.lib = L.load()
        self.device = torch.device(device)
        self.dtype = dtype
        self.dt = L.HM_DTYPE_BF16 if dtype == torch.bfloat16 else L.HM_DTYPE_F16
        self.nc, self.no = nc, nc + 5
        self.new_shape, self.stride = new_shape, stride
        self.layers = arch.yolov7_layers()
        self.specs = arch.conv_specs(self.layers, 3, nc)
        fused = fuse.fuse_state_dict(state_dict, self.specs)
        self.w: Dict[str, Tuple[torch.Tensor, torch.Tensor, int, int, int, int]] = {}
        for name, (co, ci, k, s) in self.specs.items():
            w, b = fused[name]
            cin = 8 if ci == 3 else ci                      # the image is stored with 8 channels (3 real)
            wk = torch.zeros(co, k, k, cin, dtype=torch.float32)
            wk[:, :, :, :ci] = w.permute(0, 2, 3, 1)
            flat = torch.zeros(co, _kpad(k * k * cin), dtype=torch.float32)
            flat[:, :k * k * cin] = wk.reshape(co, -1)
            self.w[name] = (flat.to(self.device, dtype).contiguous(), b.to(self.device, torch.float32).contiguous(), cin, k, s, co)
        self.zeros = torch.zeros(64, dtype=torch.uint8, device=self.device)
        self.anchors = detect_anchors(state_dict)            # pixels, per level: the checkpoint's anchor_grid (yolo.py:164)
        self.names = list(names) if names is not None else [str(i) for i in range(nc)]   # Model.names of the checkpoint
        self._plans: Dict[Tuple[int, int], dict] = {}
        self._stacked: Dict[Tuple[str, str], Tuple[torch.Tensor, torch.Tensor]] = {}   # stacked weights of fused 1x1 pairs
        self.fuse_pairs = True        # E-ELAN cv1 / cv2 as one launch (round 3); False: one launch per convolution, as the reference's graph
        self.split_k = True           # give the library split-K scratch for the small maps of the neck (round 3)
        self.fuse_stem = True         # Conv 0 + Conv 1 as one launch, the 32-channel full-size map never written (round 4); False: two launches

    # ------------------------------------------------------------------ planning
    def _plan(self, H: int, W: int, nb: int = 1) -> dict:
        """Plan for `nb` frames of size HxW processed as one batch (every conv launch covers all of them)."""
        key = (H, W, nb)
        if key in self._plans:
            self._plans[key] = self._plans.pop(key)           # (most recently used last)
            return self._plans[key]
        self._evict_plans()
        lib = self.lib
        lp = L.LetterboxPlan()
        L.check(lib.hm_letterbox_plan_make(H, W, self.new_shape, self.stride, C.byref(lp)), "hm_letterbox_plan_make")
        tab = (C.c_int32 * (3 * lp.new_w + 3 * lp.new_h))()
        L.check(lib.hm_letterbox_tables(C.byref(lp), tab), "hm_letterbox_tables")
        tab_dev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.int32).clone().to(self.device)

        layers = arch.resolve(self.layers)
        ch = arch.channels(self.layers, 3, self.nc)
        # spatial size per layer
        hw: List[Tuple[int, int]] = []
        for i, (srcs, kind, args) in enumerate(layers):
            h, w = (lp.out_h, lp.out_w) if srcs[0] < 0 else hw[srcs[0]]
            if kind == "conv" and args[2] == 2:
                h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1      # k3 s2 p1
            elif kind == "mp":
                h, w = h // 2, w // 2
            elif kind == "up":
                h, w = 2 * h, 2 * w
            hw.append((h, w))
        # homes: (buffer id, channel offset, ld); concat members live inside the concat buffer
        home: Dict[int, Tuple[int, int, int]] = {}
        sizes: Dict[int, int] = {}                          # buffer id -> elements
        for j, (srcs, kind, _) in enumerate(layers):
            if kind == "concat":
                off = 0
                for s in srcs:
                    assert s not in home, "a tensor may join one concat only"
                    home[s] = (j, off, ch[j])
                    off += ch[s]
                sizes[j] = nb * hw[j][0] * hw[j][1] * ch[j]
        for i, (srcs, kind, _) in enumerate(layers):
            if kind == "concat":
                home.setdefault(i, (i, 0, ch[i]))
            elif kind != "detect" and i not in home:
                home[i] = (i, 0, ch[i])
                sizes[i] = nb * hw[i][0] * hw[i][1] * ch[i]
        # extra buffers: image (8 ch), SPPCSPC internals, detect raw maps (f32)
        IMG = -1
        sizes[IMG] = nb * lp.out_h * lp.out_w * 8
        home[IMG] = (IMG, 0, 8)
        spp_i = next(i for i, (_, kind, _) in enumerate(layers) if kind == "sppcspc")
        c_ = layers[spp_i][2][0]
        sh, sw = hw[spp_i]
        spp = {"t1": 1000, "t3": 1001, "cat4": 1002, "t5": 1003, "cat2": 1004}
        for nm, cc in (("t1", c_), ("t3", c_), ("cat4", 4 * c_), ("t5", c_), ("cat2", 2 * c_)):
            sizes[spp[nm]] = nb * sh * sw * cc
        offs, total = {}, 0
        for b, n in sizes.items():
            offs[b] = total
            total += (n * 2 + 255) // 256 * 256
        arena = torch.zeros(total, dtype=torch.uint8, device=self.device)
        base = arena.data_ptr()
        esz = 2

        def addr(buf, ch_off=0):
            return base + offs[buf] + ch_off * esz

        def loc(i):
            b, o, ld = home[i]
            return addr(b, o), ld

        ops: List[L.YoloOp] = []
        pending = []                                           # convolutions in launch order, before pair fusion

        readers: Dict[int, int] = {}                           # layer -> how many layers read its output
        for srcs, _, _ in layers:
            for s_ in srcs:
                if s_ >= 0:
                    readers[s_] = readers.get(s_, 0) + 1
        sole = set()                                           # output pointers with exactly one reader

        def conv(name, xptr, ldx, h, w, yptr, ldy, act=1, out_f32=0):
            pending.append(("conv", name, xptr, ldx, h, w, yptr, ldy, act, out_f32))

        def emit_conv(wt, bs, cin, k, s, co, xptr, ldx, h, w, yptr, ldy, act, out_f32, kind=0):
            a = L.ConvArgs(xptr, wt.data_ptr(), yptr, bs.data_ptr(), self.zeros.data_ptr(), nb, h, w, cin, co, k, s, ldx, ldy,
                           wt.shape[1], act, out_f32, self.dt, None, 0, None, 0)
            ops.append(L.YoloOp(kind, 0, a))

        def flush():
            """Emit the pending convolutions.  Two consecutive 1x1 convolutions that read the SAME tensor and write ADJACENT
            channel slices of one buffer -- cv1 / cv2 at the head of every E-ELAN block (yolov7.yaml: `[-1, 1, Conv, ..]`,
            `[-2, 1, Conv, ..]`, both into the block's Concat) -- become ONE launch with the weight rows stacked: same
            arithmetic per output channel, one pass over the input instead of two, a wider N tile."""
            i = 0
            while i < len(pending):
                _, name, xptr, ldx, h, w, yptr, ldy, act, out_f32 = pending[i]
                wt, bs, cin, k, s, co = self.w[name]
                if self.fuse_stem and i + 1 < len(pending) and yptr in sole and (cin, k, s, co, ldx, ldy, act, out_f32) == (8, 3, 1, 32, 8, 32, 1, 0):
                    # Conv 0 -> Conv 1 (yolov7.yaml backbone: [-1, 1, Conv, [32, 3, 1]], [-1, 1, Conv, [64, 3, 2]]): nobody but the second
                    # reads the first's output -- HM_OP_CONV_PAIR: one launch, the intermediate is not written (hm_conv2d_stem_pair)
                    _, name2, xptr2, ldx2, h2, w2, yptr2, ldy2, act2, out2 = pending[i + 1]
                    wt2, bs2, cin2, k2, s2, co2 = self.w[name2]
                    if (xptr2, ldx2, h2, w2, cin2, k2, s2, co2, act2, out2) == (yptr, ldy, h, w, 32, 3, 2, 64, 1, 0):
                        emit_conv(wt, bs, cin, k, s, co, xptr, ldx, h, w, yptr, ldy, act, out_f32, kind=3)
                        emit_conv(wt2, bs2, cin2, k2, s2, co2, xptr2, ldx2, h2, w2, yptr2, ldy2, act2, out2)
                        i += 2
                        continue
                if self.fuse_pairs and i + 1 < len(pending):
                    _, name2, xptr2, ldx2, h2, w2, yptr2, ldy2, act2, out2 = pending[i + 1]
                    wt2, bs2, cin2, k2, s2, co2 = self.w[name2]
                    same_in = (xptr2, ldx2, h2, w2, cin2, k2, s2, act2, out2, ldy2) == (xptr, ldx, h, w, cin, 1, 1, act, out_f32, ldy)
                    if same_in and k == 1 and s == 1 and not out_f32 and (yptr == yptr2 + co2 * esz or yptr2 == yptr + co * esz):
                        first_is_2 = yptr == yptr2 + co2 * esz
                        key = (name2, name) if first_is_2 else (name, name2)
                        if key not in self._stacked:
                            a_, b_ = (self.w[key[0]], self.w[key[1]])
                            self._stacked[key] = (torch.cat([a_[0], b_[0]], 0).contiguous(), torch.cat([a_[1], b_[1]], 0).contiguous())
                        wst, bst = self._stacked[key]
                        emit_conv(wst, bst, cin, 1, 1, co + co2, xptr, ldx, h, w, yptr2 if first_is_2 else yptr, ldy, act, 0)
                        i += 2
                        continue
                emit_conv(wt, bs, cin, k, s, co, xptr, ldx, h, w, yptr, ldy, act, out_f32)
                i += 1
            pending.clear()

        def pool(xptr, ldx, h, w, c, yptr, ldy, k, s, pad):
            flush()
            a = L.ConvArgs(xptr, None, yptr, None, None, nb, h, w, c, c, k, s, ldx, ldy, 0, 0, 0, self.dt)
            ops.append(L.YoloOp(1, pad, a))

        raws = []
        for i, (srcs, kind, args) in enumerate(layers):
            if kind == "concat":
                continue
            s0 = srcs[0] if srcs[0] >= 0 else IMG
            xptr, ldx = loc(s0)
            h, w = (lp.out_h, lp.out_w) if s0 == IMG else hw[s0]
            if kind == "conv":
                yptr, ldy = loc(i)
                if readers.get(i, 0) == 1 and home[i][0] == i:
                    sole.add(yptr)
                conv(f"model.{i}.conv", xptr, ldx, h, w, yptr, ldy)
            elif kind == "repconv":
                yptr, ldy = loc(i)
                conv(f"model.{i}.rbr_reparam", xptr, ldx, h, w, yptr, ldy)
            elif kind == "mp":
                yptr, ldy = loc(i)
                pool(xptr, ldx, h, w, ch[s0], yptr, ldy, 2, 2, 0)
            elif kind == "up":
                yptr, ldy = loc(i)
                flush()
                a = L.ConvArgs(xptr, None, yptr, None, None, nb, h, w, ch[s0], ch[s0], 1, 1, ldx, ldy, 0, 0, 0, self.dt)
                ops.append(L.YoloOp(2, 0, a))
            elif kind == "sppcspc":                          # common.py:279-284
                p = f"model.{i}."
                conv(p + "cv1.conv", xptr, ldx, h, w, addr(spp["t1"]), c_)
                conv(p + "cv3.conv", addr(spp["t1"]), c_, h, w, addr(spp["t3"]), c_)
                conv(p + "cv4.conv", addr(spp["t3"]), c_, h, w, addr(spp["cat4"], 0), 4 * c_)          # x1
                for step in range(3):                        # maxpool 5, 9 = 5o5, 13 = 5o5o5 (stride 1, -inf padding)
                    pool(addr(spp["cat4"], step * c_), 4 * c_, h, w, c_, addr(spp["cat4"], (step + 1) * c_), 4 * c_, 5, 1, 2)
                conv(p + "cv5.conv", addr(spp["cat4"]), 4 * c_, h, w, addr(spp["t5"]), c_)
                conv(p + "cv6.conv", addr(spp["t5"]), c_, h, w, addr(spp["cat2"], 0), 2 * c_)           # y1
                conv(p + "cv2.conv", xptr, ldx, h, w, addr(spp["cat2"], c_), 2 * c_)                     # y2
                yptr, ldy = loc(i)
                conv(p + "cv7.conv", addr(spp["cat2"]), 2 * c_, h, w, yptr, ldy)
            elif kind == "detect":
                for l, s in enumerate(srcs):
                    xp, ldxx = loc(s)
                    hh, ww = hw[s]
                    raw = torch.empty(nb * hh * ww, 3 * self.no, dtype=torch.float32, device=self.device)
                    raws.append((raw, hh, ww))
                    conv(f"model.{i}.m.{l}", xp, ldxx, hh, ww, raw.data_ptr(), 3 * self.no, act=0, out_f32=1)
        flush()
        # split-K scratch (hm_conv_args.splitk_ws), shared by the plan's layers (they run one after another): the library says
        # how much each convolution needs (its rule looks at one image's output, so results do not depend on nb)
        ws_bytes = max([self.lib.hm_conv_splitk_bytes(C.byref(op.conv)) for op in ops if op.kind == 0] + [0]) if self.split_k else 0
        splitk_ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=self.device)
        if ws_bytes:
            for op in ops:
                if op.kind == 0:
                    op.conv.splitk_ws, op.conv.splitk_ws_bytes = splitk_ws.data_ptr(), ws_bytes
        op_arr = (L.YoloOp * len(ops))(*ops)
        n_pred = sum(3 * hh * ww for _, hh, ww in raws)
        plan = {
            "lp": lp, "tab": tab_dev, "arena": arena, "img_ptr": addr(IMG), "ops": op_arr, "n_ops": len(ops), "raws": raws,
            "nb": nb, "n_pred": n_pred,
            "pred": torch.empty(nb * n_pred, self.no, dtype=torch.float32, device=self.device),
            "dets": torch.zeros(nb * 300, 6, dtype=torch.float32, device=self.device),
            "count": torch.zeros(nb, dtype=torch.int32, device=self.device),
            "nms_ws": torch.empty(self.lib.hm_nms_workspace_bytes(n_pred), dtype=torch.uint8, device=self.device),
            "u8": torch.empty(3, lp.out_h, lp.out_w, dtype=torch.uint8, device=self.device),
            "home": home, "hw": hw, "ch": ch, "offs": offs, "splitk_ws": splitk_ws,
        }
        self._plans[key] = plan
        return plan

    PLAN_BYTES_BUDGET = 32 << 30      # arenas of cached plans (a 48-frame 1080p plan is ~7 GB; the folder drivers' last pass may be any size)

    def _evict_plans(self) -> None:
        """Before a NEW plan is built: drop the least recently used plans while the cached arenas exceed the budget (the two
        most recent always stay).  A dropped arena may still be read by queued kernels, so the device is synchronised first --
        this runs once per new (size, frame count), never in steady state."""
        total = sum(int(p["arena"].numel()) for p in self._plans.values())
        if total <= self.PLAN_BYTES_BUDGET or len(self._plans) <= 2:
            return
        torch.cuda.synchronize(self.device)
        for key in list(self._plans)[:-2]:
            if total <= self.PLAN_BYTES_BUDGET:
                break
            total -= int(self._plans[key]["arena"].numel())
            del self._plans[key]

    def layer_output(self, p: dict, i: int) -> torch.Tensor:
        """(C, H, W) fp32 copy of layer i's output inside the arena (debug / tests)."""
        b, o, ld = p["home"][i]
        h, w = p["hw"][i]
        start = p["offs"][b]
        buf = p["arena"][start:start + h * w * ld * 2].view(self.dtype).reshape(h, w, ld)
        return buf[:, :, o:o + p["ch"][i]].permute(2, 0, 1).float().cpu()

    # ------------------------------------------------------------------ run
    def letterbox(self, frame: torch.Tensor, want_u8: bool = False, plan: Optional[dict] = None, index: int = 0):
        H, W, _ = frame.shape
        p = plan or self._plan(H, W)
        lp = p["lp"]
        L.check(self.lib.hm_letterbox(frame.data_ptr(), C.byref(lp), p["tab"].data_ptr(), p["img_ptr"] + index * lp.out_h * lp.out_w * 16,
                                      self.dt, p["u8"].data_ptr() if want_u8 else None, L.current_stream()), "hm_letterbox")
        return p

    def forward(self, frame, want_u8: bool = False) -> dict:
        """frame: (H, W, 3) uint8 BGR device tensor, or a list of equally sized frames (one batched pass).
        Returns the plan dict with ``pred`` (nb * n, 5+nc) filled, image i in rows [i*n, (i+1)*n)."""
        frames = list(frame) if isinstance(frame, (list, tuple)) else [frame]
        for f in frames:
            if not f.is_cuda or f.dtype != torch.uint8 or f.shape != frames[0].shape:
                raise L.HipLibraryError("YoloEngine.forward takes uint8 device frames of one size")
        H, W, _ = frames[0].shape
        p = self._plan(H, W, len(frames))
        # frames that are equally spaced slices of one device tensor (the folder drivers upload a chunk as one tensor): ONE
        # letterbox launch for the pass; otherwise one per frame
        nb = len(frames)
        stride = frames[1].data_ptr() - frames[0].data_ptr() if nb > 1 else 0
        batched = nb > 1 and stride >= H * W * 3 and all(f.is_contiguous() for f in frames) and \
            all(frames[i].data_ptr() - frames[0].data_ptr() == i * stride for i in range(nb))
        if batched:
            L.check(self.lib.hm_letterbox_batch(frames[0].data_ptr(), stride, nb, C.byref(p["lp"]), p["tab"].data_ptr(), p["img_ptr"],
                                                self.dt, L.current_stream()), "hm_letterbox_batch")
        else:
            for i, f in enumerate(frames):
                self.letterbox(f.contiguous(), want_u8 and len(frames) == 1, p, i)
        st = L.current_stream()
        L.check(self.lib.hm_yolo_run(p["ops"], p["n_ops"], st), "hm_yolo_run")
        row0 = 0
        for l, (raw, hh, ww) in enumerate(p["raws"]):        # one decode launch per level for all images of the pass
            anc = (C.c_float * 6)(*self.anchors[l])
            L.check(self.lib.hm_yolo_decode_batch(raw.data_ptr(), 3 * self.no, p["pred"].data_ptr(), row0, hh, ww, self.nc,
                                                  float(arch.STRIDES[l]), anc, p["nb"], p["n_pred"], st), "hm_yolo_decode_batch")
            row0 += 3 * hh * ww
        return p

    def nms_enqueue(self, p: dict, conf_thres: float, iou_thres: float, classes: Optional[List[int]], agnostic: bool,
                    scale: bool = True, max_det: int = 300) -> None:
        """Enqueue NMS for every image of the plan; results stay on the device (``dets`` rows [i*300, ..), ``count[i]``)."""
        mask = 0xFFFFFFFF if classes is None else sum(1 << int(c) for c in classes)
        n = p["n_pred"]
        for i in range(p["nb"]):
            L.check(self.lib.hm_yolo_nms(p["pred"].data_ptr() + i * n * self.no * 4, n, self.nc, conf_thres, iou_thres, mask,
                                         int(bool(agnostic)), max_det, C.byref(p["lp"]) if scale else None,
                                         p["dets"].data_ptr() + i * 300 * 24, p["count"].data_ptr() + i * 4,
                                         p["nms_ws"].data_ptr(), p["nms_ws"].numel(), L.current_stream()), "hm_yolo_nms")

    def nms(self, p: dict, conf_thres: float, iou_thres: float, classes: Optional[List[int]], agnostic: bool,
            scale: bool = True, max_det: int = 300):
        """NMS + one host sync (the box list is host data).  One (k, 6) tensor, or a list of them for a batched plan."""
        self.nms_enqueue(p, conf_thres, iou_thres, classes, agnostic, scale, max_det)
        counts = p["count"].tolist()
        outs = [p["dets"][i * 300:i * 300 + int(k)].clone() for i, k in enumerate(counts)]
        return outs[0] if p["nb"] == 1 else outs
