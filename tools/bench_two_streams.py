"""Throughput with two batches in flight: consecutive forwards alternate between two HIP streams (own workspace and
outputs each), so the HBM-bound phases of one batch (LayerNorm, attention, GEMM epilogues, decoder) can overlap the MFMA
phases of the other.  Compared with the same number of forwards on one stream."""
import sys, time, torch
sys.path.insert(0, ".")
from hamer_yolo_amd import synth
from hamer_yolo_amd.engine import HamerEngine
cfg = synth.HamerConfig()
sd = synth.hamer_state_dict(cfg, seed=0, device="cuda")
eng = HamerEngine(sd, synth.mano_params(seed=0), cfg)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
img = synth.normalize_crops(synth.crops_u8(B, seed0=0)).cuda()
n = eng.lib.hm_hamer_workspace_bytes(__import__("ctypes").byref(eng.w), B)
NS = 4
wss = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(NS)]
outs = [eng.alloc_outputs(B) for _ in range(NS)]
import os
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tools'))
from runlog import banner
banner()
PRI = os.environ.get("PRI", "0") == "1"
streams = [torch.cuda.Stream(priority=(-1 if (PRI and i % 2 == 0) else 0)) for i in range(NS)]
def run(nstreams, steps=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % nstreams
        with torch.cuda.stream(streams[k]):
            eng.forward(img, outs[k], workspace=wss[k])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps
for _ in range(2):
    run(1, 6); run(2, 6)
ref = {k: v.clone() for k, v in outs[0].items()}
for rep in range(2):
    ts = [run(k, 48) for k in (1, 2, 3, 4)]
    print(f"B={B} " + " | ".join(f"{k} streams {1e3*t:.3f} ms ({B/t:.0f}/s)" for k, t in zip((1, 2, 3, 4), ts)), flush=True)
print("identical:", all(torch.equal(ref[k], outs[1][k]) and torch.equal(ref[k], outs[0][k]) for k in ref))
