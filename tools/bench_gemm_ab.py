#!/usr/bin/env python3
"""Interleaved A/B of GEMM tile variants on the ViT-H shapes (random data): ROUNDS rounds, every round runs
every variant REPS times per shape; reports the median over all launches of a variant (200 / 201: the default tile with
the fp32 residual fetched in the epilogue / inside the K loop; >= 100 otherwise: default tile with group_m = v - 100).  One process, one
device, variants interleaved so clock drift hits them equally.  Env: VARIANTS=8,9,10 ROUNDS=6 REPS=5"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamer_yolo_amd import lib as L
from hamer_yolo_amd import ops

if os.environ.get("ABLATION_LIB"):     # python -m hamer_yolo_amd.build --ablations: variants that skip work (wrong results, timing only)
    L.LIB_PATH = L.LIB_PATH.replace(".so", "_abl.so")

variants = [int(v) for v in os.environ.get("VARIANTS", "0,8").split(",")]      # v >= 100: default variant with group_m = v - 100
rounds, reps = int(os.environ.get("ROUNDS", 6)), int(os.environ.get("REPS", 5))
M = int(os.environ.get("BATCH", 64)) * 192
DT = torch.float16 if os.environ.get("DTYPE", "fp16") == "fp16" else torch.bfloat16
ONLY = [n for n in os.environ.get("SHAPES", "").split(",") if n]
dev = "cuda"
torch.manual_seed(0)
shapes = [("qkv", 1280, 3840, L.HM_EPI_STORE), ("proj", 1280, 1280, L.HM_EPI_RESID_F32), ("fc1", 1280, 5120, L.HM_EPI_GELU),
          ("fc2", 5120, 1280, L.HM_EPI_RESID_F32), ("kv", 1280, 6144, L.HM_EPI_STORE)]
if ONLY:
    shapes = [sh for sh in shapes if sh[0] in ONLY]
if os.environ.get("EPI_STORE"):                # every shape with the plain 16-bit store epilogue (K-loop comparisons)
    shapes = [(n, k, nn, L.HM_EPI_STORE) for (n, k, nn, _) in shapes]
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tools'))
from runlog import banner
banner()
lib = L.load()
res = {}


def setv(v):
    L.check(lib.hm_set_option(L.HM_OPT_RESID_IN_EPILOGUE, 1 if v == 200 else 0))
    L.check(lib.hm_set_option(L.HM_OPT_PX_LDS_EPILOGUE, 1 if v == 202 else (2 if v == 201 else 0)))     # 202 / 201: persistent GEMM epilogue through LDS / by lane swaps
    if v in (200, 201, 202):             # 200 / 201: default tile with the fp32 residual fetched in the epilogue (round 2) / inside the K loop (round 3)
        L.check(lib.hm_gemm_set_variant(-1)); L.check(lib.hm_gemm_set_group_m(8))
    elif v >= 100:
        L.check(lib.hm_gemm_set_variant(-1)); L.check(lib.hm_gemm_set_group_m(v - 100))
    else:
        L.check(lib.hm_gemm_set_variant(v)); L.check(lib.hm_gemm_set_group_m(8))


for (name, K, N, epi) in shapes:
    PAD = int(os.environ.get("PAD", 0))            # extra elements per operand row (row pitch = K + PAD): L2 channel spread experiment
    x = torch.randn(M, K + PAD, device=dev).to(DT)[:, :K]
    w = (torch.randn(N, K + PAD, device=dev) * 0.02).to(DT)[:, :K]
    b = torch.randn(N, device=dev)
    f32 = epi == L.HM_EPI_RESID_F32
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else DT)
    r = torch.randn(M, N, device=dev) if f32 else None
    if f32 and os.environ.get("COLD_RESID"):      # as in a forward: the residual stream comes from HBM, not from the Infinity Cache
        flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    times = {v: [] for v in variants}
    for v in variants:                        # warm
        setv(v); ops.gemm(x, w, b, epi, resid=r, out=out)
    torch.cuda.synchronize()
    for _ in range(rounds):
        for v in variants:
            setv(v)
            with L.profile(capacity=reps + 2) as prof:
                for _ in range(reps):
                    if f32 and os.environ.get("COLD_RESID"):
                        flush.add_(1)                 # 512 MiB read + written: evicts the residual from the Infinity Cache
                    ops.gemm(x, w, b, epi, resid=r, out=out)
                torch.cuda.synchronize()
            times[v] += [rec[5] for rec in prof.records]
    if os.environ.get("YARDSTICK"):           # the vendor library on the same operands (plain x @ w.T + b, 16-bit out): a yardstick, not a product path
        lin = lambda: torch.nn.functional.linear(x.contiguous(), w.contiguous(), b.to(DT))
        xc, wc, bc = x.contiguous(), w.contiguous(), b.to(DT)
        for _ in range(3):
            torch.nn.functional.linear(xc, wc, bc)
        ts = []
        for _ in range(rounds * reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); torch.nn.functional.linear(xc, wc, bc); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort(); med = ts[len(ts) // 2]
        print(f"{name:5s} hipBLASLt (torch linear, bias, 16-bit out) median {med*1e3:7.1f} us  min {ts[0]*1e3:7.1f}  {2.0*M*N*K/med/1e9:7.1f} TF", flush=True)
    for v in variants:
        t = sorted(times[v]); med = t[len(t) // 2]
        res[(name, v)] = med
        print(f"{name:5s} v{v:<3d} median {med*1e3:7.1f} us  min {t[0]*1e3:7.1f}  {2.0*M*N*K/med/1e9:7.1f} TF", flush=True)
print("per-layer sum (us):", {v: round(sum(res[(n, v)] for n, *_ in shapes) * 1e3, 1) for v in variants})
