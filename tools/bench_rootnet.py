"""RootNet (ResNet-34 + depth head) forward time per batch of 256x256 patches."""
import sys, time, torch
sys.path.insert(0, ".")
from hamer_yolo_amd import synth
from hamer_yolo_amd.rootnet.engine import RootNetEngine
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tools'))
from runlog import banner
banner()
net, root = synth.rootnet_state_dict(0)
eng = RootNetEngine(net, root)
for B in (1, 4, 64):
    img = synth.normalize_crops(synth.crops_u8(B, seed0=0)).cuda()
    kv = torch.ones(B)
    for _ in range(3): eng.forward(img, kv)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eng.forward(img, kv)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
    print(f"B={B:3d}: {1e3*t:.3f} ms  ({B/t:.0f} patches/s, {B*7.3/t/1e3:.1f} TFLOP/s at 7.3 GFLOP/patch)", flush=True)
