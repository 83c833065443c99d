"""Why is the 1024-crop shard job (configs[3], N = 1) a few per cent below the contract line?  Same process, same engine:
(a) 16 forwards of the SAME resident batch on two alternating contexts (the contract line's step, 16 times),
(b) 16 forwards over 16 DIFFERENT 64-crop slices of a resident 1024-crop tensor,
(c) (b) + pack_mano into the job's [1024][157] buffer,
(d) ShardJob.step() (forward_on: stream waits + record_stream; end-of-job join; gather)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamer_yolo_amd import shard, synth
from hamer_yolo_amd.engine import HamerEngine
from runlog import banner
banner()
cfg = synth.HamerConfig()
eng = HamerEngine(synth.hamer_state_dict(cfg, seed=0, device="cuda"), synth.mano_params(seed=0), cfg)
crops = synth.normalize_crops(synth.crops_u8(1024, seed0=0)).cuda()
job = shard.ShardJob(eng, crops, 1024, batch=64, in_flight=2)
ctxs = job.ctxs
packed = torch.zeros(1024, shard.PARAMS_PER_HAND, device="cuda")
same = crops[:64].contiguous()


def a():
    for j in range(16):
        c = ctxs[j % 2]
        with torch.cuda.stream(c.stream):
            eng.forward(same, c.out, workspace=c.workspace)


def b():
    for j in range(16):
        c = ctxs[j % 2]
        with torch.cuda.stream(c.stream):
            eng.forward(crops[64 * j:64 * j + 64], c.out, workspace=c.workspace)


def c_():
    for j in range(16):
        c = ctxs[j % 2]
        with torch.cuda.stream(c.stream):
            eng.forward(crops[64 * j:64 * j + 64], c.out, workspace=c.workspace)
            packed[64 * j:64 * j + 64] = shard.pack_mano(c.out)


for name, fn in (("same batch x16", a), ("16 slices", b), ("16 slices + pack", c_), ("ShardJob.step", job.step)):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        fn(); fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 2 * 1e3)
    ts.sort()
    print(f"{name:18s} median {ts[2]:7.2f} ms per 1024 crops = {1024 / ts[2] * 1e3:7.1f} hands/s", flush=True)
