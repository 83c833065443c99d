"""cProfile of one pass of the folder driver (infer.process_batch_manopara; DEPTH=1: d_infer.process_batch_manopara with RootNet) over 64 seeded 1080p .bmp frames: where the HOST time goes."""
import cProfile, io, os, pstats, shutil, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from PIL import Image
from hamer_yolo_amd import infer, synth
from hamer_yolo_amd.yolo.detector import Detector
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))), 'tools'))
from runlog import banner
banner()

class YCfg:
    weights = os.environ.get("YOLO_WEIGHTS", "synthetic:2:-2.53:0"); imgsz = 640; augment = True; conf_thres = 0.25; iou_thres = 0.35
    classes = [0, 1, 2]; agnostic_nms = True; device = "cuda"; save_path = "./output"
class HCfg:
    ckpt_path = "synthetic:0"; model_cfg = None; use_onnx = False; onnx_path = None
root = tempfile.mkdtemp(prefix="e2e_prof_", dir="/dev/shm")
try:
    ind, outd = os.path.join(root, "rgb"), os.path.join(root, "out")
    os.makedirs(ind)
    for i in range(64):
        Image.fromarray(synth.frame_u8(1080, 1920, seed=i % 8).numpy()[:, :, ::-1]).save(os.path.join(ind, f"f{i:04d}.bmp"))
    hi, det = infer.hamer_inference(HCfg), Detector(YCfg)
    import contextlib
    import numpy as np
    if os.environ.get("DEPTH") == "1":                  # the d_infer flow: + RootNet root depth per hand
        from hamer_yolo_amd import d_infer
        from hamer_yolo_amd.rootnet.Model_RGB import get_model
        sar, k_real = get_model(), np.array([[1400.0, 0, 960], [0, 1400.0, 540], [0, 0, 1]], np.float32)
        run = lambda: d_infer.process_batch_manopara(ind, outd, k_real, hamer=hi, detector=det, sar=sar)
    else:
        run = lambda: infer.process_batch_manopara(ind, outd, None, hamer=hi, detector=det)
    with contextlib.redirect_stdout(io.StringIO()):
        run(); run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pr = cProfile.Profile(); pr.enable()
        run()
        torch.cuda.synchronize()
        pr.disable()
    print("pass: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
finally:
    shutil.rmtree(root, ignore_errors=True)
