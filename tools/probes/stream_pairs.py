"""Does it matter WHICH two HIP streams carry the two batches in flight?  Eight streams created in order (HIP deals streams onto its
hardware queues in creation order); every pair (i, j) runs the contract step (B = 64, two batches in flight) for STEPS steps,
pairs interleaved over ROUNDS rounds.  Also: GPU_MAX_HW_QUEUES as set in the environment."""
import itertools, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamer_yolo_amd import synth
from hamer_yolo_amd.engine import HamerEngine, ForwardContext
from runlog import banner
banner("GPU_MAX_HW_QUEUES=" + os.environ.get("GPU_MAX_HW_QUEUES", "(default)"))
import ctypes as C
cfg = synth.HamerConfig()
eng = HamerEngine(synth.hamer_state_dict(cfg, seed=0, device="cuda"), synth.mano_params(seed=0), cfg)
B, NS = 64, int(os.environ.get("NSTREAMS", 6))
img = synth.normalize_crops(synth.crops_u8(B, seed0=0)).cuda()
streams = [torch.cuda.Stream() for _ in range(NS)]
nbytes = eng.lib.hm_hamer_workspace_bytes(C.byref(eng.w), B)
ws = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
outs = [eng.alloc_outputs(B) for _ in range(2)]
pairs = list(itertools.combinations(range(NS), 2))
steps, rounds = int(os.environ.get("STEPS", 12)), int(os.environ.get("ROUNDS", 3))
times = {p: [] for p in pairs}


def run(pair, n):
    for i in range(n):
        with torch.cuda.stream(streams[pair[i % 2]]):
            eng.forward(img, outs[i % 2], workspace=ws[i % 2])


for r in range(rounds):
    for p in pairs:
        run(p, 2); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(p, steps); torch.cuda.synchronize()
        times[p].append((time.perf_counter() - t0) / steps * 1e3)
res = sorted((sorted(v)[len(v) // 2], p) for p, v in times.items())
for t, p in res:
    print(f"streams {p}: {t:7.3f} ms/step", flush=True)
