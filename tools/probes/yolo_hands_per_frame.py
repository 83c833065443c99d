"""Boxes per 1080p frame the synthetic detector finds as a function of its objectness bias (synthetic:<seed>:<obj_bias>:<cls_bias>),
over the 8 seeded frames the e2e bench uses: picks the bias for BASELINE configs[2]'s "~4 hands/frame"."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hamer_yolo_amd import synth
from hamer_yolo_amd.yolo.detector import Detector
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from runlog import banner
banner()
frames = [synth.frame_u8(1080, 1920, seed=i).numpy() for i in range(8)]
for bias in [float(v) for v in (sys.argv[1:] or ["-2.2", "-2.6", "-3.0", "-3.4", "-3.8", "-4.2"])]:
    class YCfg:
        weights = f"synthetic:2:{bias}:0"; imgsz = 640; augment = True; conf_thres = 0.25; iou_thres = 0.35
        classes = [0, 1, 2]; agnostic_nms = True; device = "cuda"; save_path = "./output"
    det = Detector(YCfg)
    counts = [len(det.detect(f)[1][0]) for f in frames]
    print(f"obj_bias {bias:5.2f}: boxes per frame {counts}  mean {sum(counts)/len(counts):.2f}", flush=True)
