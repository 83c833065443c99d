#!/usr/bin/env python3
"""Interleaved A/B of WHOLE forwards (BASELINE configs[1]: B = 64 crops resident in HBM) under different library switches, in
one process on one device: ROUNDS rounds, every round times STEPS steps of every configuration (wall clock around a
synchronised region, the bench.py protocol).  Box-to-box noise is +-2 %; only numbers from one process compare.
Configurations: CONFIGS="name:in_flight:opt=val,opt=val;..." with opt in {resid_epi, variant, px_grid, px_lds_epi, fold_ln}.
Default: serial and two-in-flight, residual in the epilogue (round 2) vs inside the K loop (round 3)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hamer_yolo_amd import lib as L
from hamer_yolo_amd import synth
from hamer_yolo_amd.engine import HamerEngine
from runlog import banner

banner()
spec = os.environ.get("CONFIGS", "serial_old:1:resid_epi=1;serial_new:1:;two_old:2:resid_epi=1;two_new:2:;three_new:3:")
rounds, steps, B = int(os.environ.get("ROUNDS", 5)), int(os.environ.get("STEPS", 20)), int(os.environ.get("BATCH", 64))
cfg = synth.HamerConfig()
sd = synth.hamer_state_dict(cfg, seed=0, device="cuda")
mano = synth.mano_params(seed=0)
lib = L.load()
engines = {}


def engine(fold):
    if fold not in engines:
        engines[fold] = HamerEngine(sd, mano, cfg, fold_ln=fold)
    return engines[fold]


configs = []
for item in [c for c in spec.split(";") if c]:
    name, nfl, opts = item.split(":")
    o = dict(kv.split("=") for kv in opts.split(",") if kv)
    eng = engine(o.get("fold_ln") == "1")
    configs.append((name, int(nfl), o, eng, None))
# ONE set of contexts (HIP streams, workspaces) per engine, shared by every configuration: streams map onto a few hardware
# queues, and two configurations on different stream pairs do not overlap alike
ctx_of = {id(e): e.contexts(B, max(c[1] for c in configs if c[3] is e)) for e in engines.values()}
configs = [(n, k, o, e, ctx_of[id(e)][:k]) for (n, k, o, e, _) in configs]
img = synth.normalize_crops(synth.crops_u8(B, seed0=0)).cuda()


def apply(o):
    L.check(lib.hm_set_option(L.HM_OPT_RESID_IN_EPILOGUE, int(o.get("resid_epi", 0))))
    L.check(lib.hm_set_option(L.HM_OPT_PX_GRID, int(o.get("px_grid", 0))))
    L.check(lib.hm_set_option(L.HM_OPT_PX_LDS_EPILOGUE, int(o.get("px_lds_epi", 0))))
    L.check(lib.hm_gemm_set_variant(int(o.get("variant", -1))))


def run(eng, ctxs, n):
    for i in range(n):
        c = ctxs[i % len(ctxs)]
        with torch.cuda.stream(c.stream):
            eng.forward(img, c.out, workspace=c.workspace)


times = {c[0]: [] for c in configs}
for name, nfl, o, eng, ctxs in configs:
    apply(o); run(eng, ctxs, 4)
torch.cuda.synchronize()
for r in range(rounds):
    for name, nfl, o, eng, ctxs in configs:
        apply(o)
        run(eng, ctxs, 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(eng, ctxs, steps)
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / steps * 1e3)
apply({})
for name, *_ in configs:
    t = sorted(times[name])
    med = t[len(t) // 2]
    print(f"{name:14s} median {med:7.3f} ms/step  min {t[0]:7.3f}  {B / med * 1e3:7.1f} hands/s  {B / med * 1e3 * 251.03e9 / 2.5e15 * 100:5.2f} % of peak", flush=True)
