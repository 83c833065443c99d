#!/usr/bin/env python3
"""HaMeR forward time against the number of hands in the batch (one process, two forwards in flight as the drivers run them):
which batch sizes fall off the tuned B = 64 shape (persistent 256 x 256 GEMM tiles: 48 M-tiles at B = 64)?
Env: BATCHES="56,60,64,66,68,72,80,96,128"."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hamer_yolo_amd import synth, lib as L
from hamer_yolo_amd.engine import HamerEngine
from runlog import banner
banner()
cfg = synth.HamerConfig()
eng = HamerEngine(synth.hamer_state_dict(cfg, seed=0, device="cuda"), synth.mano_params(seed=0), cfg)
for B in [int(b) for b in os.environ.get("BATCHES", "16,24,32,40,48,56,60,64,68,72,76,80,88,96,128").split(",")]:
    ctxs = eng.contexts(B, 2)
    img = synth.normalize_crops(synth.crops_u8(B, seed0=0)).cuda()
    def run(n):
        for i in range(n):
            c = ctxs[i % 2]
            with torch.cuda.stream(c.stream):
                eng.forward(img, c.out, workspace=c.workspace)
    res = {}
    for rule in (1, 0, 1, 0):                            # interleaved: round 2's tile rule (1) and the rate model (0)
        L.check(L.load().hm_set_option(L.HM_OPT_GEMM_TILE_RULE, rule))
        run(4); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); run(10); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10 * 1e3)
        res.setdefault(rule, []).append(sorted(ts)[1])
    L.check(L.load().hm_set_option(L.HM_OPT_GEMM_TILE_RULE, 0))
    t1, t0_ = min(res[1]), min(res[0])
    print(f"B={B:4d}: round-2 rule {t1:7.3f} ms ({t1 / B * 1e3:6.1f} us per hand)   rate model {t0_:7.3f} ms ({t0_ / B * 1e3:6.1f} us per hand, {B / t0_ * 1e3:7.1f} hands/s)", flush=True)
    del ctxs, img
    torch.cuda.empty_cache()
