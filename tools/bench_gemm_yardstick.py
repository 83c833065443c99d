#!/usr/bin/env python3
"""Like-for-like yardstick: hm_gemm (HM_EPI_STORE: bias, 16-bit out) against the vendor library's kernel for the same
x @ w.T + b (torch.nn.functional.linear -> hipBLASLt), interleaved launch by launch groups so that clock drift hits both
alike.  A measuring tool only: nothing in the package calls the vendor library.  Env: BATCH=64 ROUNDS=8 REPS=4 DTYPE=fp16"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamer_yolo_amd import lib as L
from hamer_yolo_amd import ops
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tools'))
from runlog import banner
banner()

rounds, reps = int(os.environ.get("ROUNDS", 8)), int(os.environ.get("REPS", 4))
M = int(os.environ.get("BATCH", 64)) * 192
DT = torch.float16 if os.environ.get("DTYPE", "fp16") == "fp16" else torch.bfloat16
dev = "cuda"
torch.manual_seed(0)
shapes = [("qkv", 1280, 3840), ("proj", 1280, 1280), ("fc1", 1280, 5120), ("fc2", 5120, 1280)]


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, K, N in shapes:
    x = torch.randn(M, K, device=dev).to(DT)
    w = (torch.randn(N, K, device=dev) * 0.02).to(DT)
    b = torch.randn(N, device=dev)
    b16 = b.to(DT)
    out = torch.empty(M, N, device=dev, dtype=DT)
    mine = lambda: ops.gemm(x, w, b, L.HM_EPI_STORE, out=out)
    theirs = lambda: torch.nn.functional.linear(x, w, b16)
    for _ in range(3):
        mine(); theirs()
    torch.cuda.synchronize()
    tm, tt = [], []
    for _ in range(rounds):
        tm.append(timed(mine)); tt.append(timed(theirs))
    tm.sort(); tt.sort()
    f = 2.0 * M * N * K / 1e9
    print(f"{name:5s} M={M} N={N} K={K}: hm_gemm median {tm[len(tm)//2]*1e3:7.1f} us ({f/tm[len(tm)//2]:6.0f} TF, best {f/tm[0]:6.0f})   "
          f"library median {tt[len(tt)//2]*1e3:7.1f} us ({f/tt[len(tt)//2]:6.0f} TF, best {f/tt[0]:6.0f})", flush=True)
