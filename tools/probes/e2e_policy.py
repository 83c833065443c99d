"""Folder-driver policies A/B in one process (same models, same files): detector passes on their own stream or on the first
HaMeR stream, the folder's last hands as full forwards + remainder or in equal parts.  hands/s, best of 3, for 64 and 192 frames.
Usage: python tools/probes/e2e_policy.py"""
import os, sys, tempfile, shutil, time, contextlib, io, itertools
sys.path.insert(0, ".")
import numpy as np, torch
from PIL import Image
from hamer_yolo_amd import infer, synth
from hamer_yolo_amd.yolo.detector import Detector


class YCfg:
    weights = "synthetic:2:-2.53:0"; imgsz = 640; augment = True; conf_thres = 0.25; iou_thres = 0.35
    classes = [0, 1, 2]; agnostic_nms = True; device = "cuda"; save_path = "./output"


class HCfg:
    ckpt_path = "synthetic:0"; model_cfg = None; use_onnx = False; onnx_path = None


root = tempfile.mkdtemp(dir="/dev/shm")
try:
    hi, det = infer.hamer_inference(HCfg), Detector(YCfg)
    for N in (64, 192):
        ind = os.path.join(root, f"rgb{N}"); os.makedirs(ind)
        for i in range(8):
            Image.fromarray(synth.frame_u8(1080, 1920, seed=i).numpy()[:, :, ::-1]).save(os.path.join(ind, f"f{i:05d}.bmp"))
        for i in range(8, N):
            os.link(os.path.join(ind, f"f{i % 8:05d}.bmp"), os.path.join(ind, f"f{i:05d}.bmp"))
        k = 0
        for overlap, balance, excl in ((True, False, None), (True, True, None)):
            best, sizes, times = 1e9, None, []
            for rep in range(6):
                k += 1
                t0 = time.perf_counter()
                with contextlib.redirect_stdout(io.StringIO()):
                    st = infer.process_batch_manopara(ind, os.path.join(root, f"o{N}_{k}"), None, hamer=hi, detector=det,
                                                      overlap_detector=overlap, balance_tail=balance, det_ramp=excl)
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                if rep:
                    best = min(best, el); times.append(round(el * 1e3, 1))
            print(f"{N:4d} frames  detector on its own stream={overlap!s:5}  balanced tail={balance!s:5}  first detector passes={excl!s:8}: {st['hands'] / best:7.1f} hands/s  "
                  f"({best * 1e3:6.1f} ms, {st['hands']} hands, {st['forwards']} forwards, {st['det_passes']} detector passes; all passes {times})", flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)
