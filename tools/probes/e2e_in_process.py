"""Why is the e2e pass slower as a side configuration of bench.py than in a process of its own?  One process, the e2e pass timed
after each piece of bench.py's main phase is added."""
import contextlib, io, os, shutil, sys, tempfile, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from hamer_yolo_amd import infer, synth
from hamer_yolo_amd.engine import HamerEngine
from hamer_yolo_amd.yolo.detector import Detector
from runlog import banner
banner()


class YCfg:
    weights = "synthetic:2:-2.53:0"; imgsz = 640; augment = True; conf_thres = 0.25; iou_thres = 0.35
    classes = [0, 1, 2]; agnostic_nms = True; device = "cuda"; save_path = "./output"


class HCfg:
    ckpt_path = "synthetic:0"; model_cfg = None; use_onnx = False; onnx_path = None


root = tempfile.mkdtemp(prefix="e2e_inproc_", dir="/dev/shm")
ind, outd = os.path.join(root, "rgb"), os.path.join(root, "out")
os.makedirs(ind)
seeded = [synth.frame_u8(1080, 1920, seed=i).numpy() for i in range(8)]
for i in range(64):
    Image.fromarray(seeded[i % 8][:, :, ::-1]).save(os.path.join(ind, f"f{i:04d}.bmp"))
hi, det = infer.hamer_inference(HCfg), Detector(YCfg)


def e2e(tag):
    with contextlib.redirect_stdout(io.StringIO()):
        infer.process_batch_manopara(ind, outd, None, hamer=hi, detector=det)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            shutil.rmtree(outd, ignore_errors=True)
            t0 = time.perf_counter()
            infer.process_batch_manopara(ind, outd, None, hamer=hi, detector=det)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{tag:52s} passes: " + " ".join(f"{t:6.1f}" for t in ts) + " ms", flush=True)


try:
    e2e("e2e engines only")
    cfg = synth.HamerConfig()
    sd = synth.hamer_state_dict(cfg, seed=0, device="cuda")
    e2e("+ a second fp32 state dict on the device")
    eng = HamerEngine(sd, synth.mano_params(seed=0), cfg)
    ctxs = eng.contexts(64, 2)
    e2e("+ main HamerEngine and two contexts")
    img = synth.normalize_crops(synth.crops_u8(64, seed0=0)).cuda()
    for i in range(25):
        c = ctxs[i % 2]
        with torch.cuda.stream(c.stream):
            eng.forward(img, c.out, workspace=c.workspace)
    torch.cuda.synchronize()
    e2e("+ 25 main steps on the contexts")
    torch.cuda.empty_cache()
    e2e("+ empty_cache")
finally:
    shutil.rmtree(root, ignore_errors=True)
