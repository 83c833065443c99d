"""Time of the batched detector pass (letterbox + YOLOv7 + decode + NMS) for 16 1080p frames, with the per-kernel-family split."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hamer_yolo_amd import lib as L, synth
from hamer_yolo_amd.yolo.detector import Detector
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))), 'tools'))
from runlog import banner
banner()

class YCfg:
    weights = "synthetic:2:-2.2:0"; imgsz = 640; augment = True; conf_thres = 0.25; iou_thres = 0.35
    classes = [0, 1, 2]; agnostic_nms = True; device = "cuda"; save_path = "./output"
det = Detector(YCfg)
frames = [synth.frame_u8(1080, 1920, seed=i % 8).cuda() for i in range(16)]
for _ in range(3):
    det.detect_frames(frames)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    det.detect_frames(frames)
torch.cuda.synchronize()
print("detect_frames(16 x 1080p): %.2f ms per pass = %.3f ms per frame" % ((time.perf_counter() - t0) * 100, (time.perf_counter() - t0) * 100 / 16))
with L.profile(capacity=4096) as prof:
    det.detect_frames(frames)
    torch.cuda.synchronize()
by = {}
for kind, epi, M, N, K, ms in prof.records:
    by[kind] = by.get(kind, 0.0) + ms
fl = sum(2.0 * M * N * K for kind, epi, M, N, K, ms in prof.records if kind == "conv")
cms = by.get("conv", 0.0)
print({k: round(v, 3) for k, v in sorted(by.items(), key=lambda kv: -kv[1])}, "conv TFLOP/s: %.0f" % (fl / cms / 1e9 if cms else 0), "launches", len(prof.records))
top = sorted([r for r in prof.records if r[0] == "conv"], key=lambda r: -r[5])[:8]
for r in top:
    print("  conv M=%d N=%d K=%d: %.3f ms  %.0f TF/s" % (r[2], r[3], r[4], r[5], 2.0 * r[2] * r[3] * r[4] / r[5] / 1e9))
