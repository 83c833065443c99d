"""Host timeline of one pass of the folder driver (HAMER_E2E_TRACE=1): when detector passes and HaMeR batches are enqueued,
harvested and finished, relative to the start of the pass.  Usage: python tools/probes/e2e_trace.py [frames=64]"""
import os, sys, tempfile, shutil, time, contextlib, io
sys.path.insert(0, ".")
os.environ["HAMER_E2E_TRACE"] = "1"
import numpy as np, torch
from PIL import Image
from hamer_yolo_amd import infer, synth
from hamer_yolo_amd.yolo.detector import Detector
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64


class YCfg:
    weights = "synthetic:2:-2.53:0"; imgsz = 640; augment = True; conf_thres = 0.25; iou_thres = 0.35
    classes = [0, 1, 2]; agnostic_nms = True; device = "cuda"; save_path = "./output"


class HCfg:
    ckpt_path = "synthetic:0"; model_cfg = None; use_onnx = False; onnx_path = None


root = tempfile.mkdtemp(dir="/dev/shm")
try:
    ind = os.path.join(root, "rgb"); os.makedirs(ind)
    for i in range(8):
        Image.fromarray(synth.frame_u8(1080, 1920, seed=i).numpy()[:, :, ::-1]).save(os.path.join(ind, f"f{i:05d}.bmp"))
    for i in range(8, N):
        os.link(os.path.join(ind, f"f{i % 8:05d}.bmp"), os.path.join(ind, f"f{i:05d}.bmp"))
    hi, det = infer.hamer_inference(HCfg), Detector(YCfg)
    for rep in range(4):
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            st = infer.process_batch_manopara(ind, os.path.join(root, f"o{rep}"), None, hamer=hi, detector=det)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) * 1e3
    print(f"pass {el:.1f} ms, {st['hands']} hands, {st['forwards']} forwards, {st['det_passes']} detector passes")
    for t, w in st["trace"]:
        print(f"  {t:8.2f}  {w}")
finally:
    shutil.rmtree(root, ignore_errors=True)
