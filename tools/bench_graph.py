"""Few-hands forward: stream launches vs one hipGraph replay (torch.cuda.CUDAGraph capturing the library's launches)."""
import sys, time, torch
sys.path.insert(0, ".")
from hamer_yolo_amd import synth
from hamer_yolo_amd.engine import HamerEngine
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tools'))
from runlog import banner
banner()
cfg = synth.HamerConfig()
sd = synth.hamer_state_dict(cfg, seed=0, device="cuda")
eng = HamerEngine(sd, synth.mano_params(seed=0), cfg)
for B in (1, 4, 16, 64):
    img = synth.normalize_crops(synth.crops_u8(B, seed0=0)).cuda()
    out = eng.alloc_outputs(B)
    for _ in range(3):
        eng.forward(img, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.forward(img, out)
    torch.cuda.synchronize()
    t_stream = (time.perf_counter() - t0) / 20
    ref = {k: v.clone() for k, v in out.items()}
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.forward(img, out)
        with torch.cuda.graph(g, stream=s):
            eng.forward(img, out)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / 20
    same = all(torch.equal(ref[k], out[k]) for k in ref)
    print(f"B={B:3d} stream {1e3*t_stream:7.3f} ms  graph {1e3*t_graph:7.3f} ms  identical={same}", flush=True)
