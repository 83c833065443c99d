#!/usr/bin/env python3
"""Run one GEMM shape a few times (for rocprofv3 --pmc runs).  Env: VARIANT, GM, GN, GK, EPI, REPS."""
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

M, N, K = int(os.environ.get("GM", 12288)), int(os.environ.get("GN", 1280)), int(os.environ.get("GK", 5120))
epi = int(os.environ.get("EPI", L.HM_EPI_STORE))
L.check(L.load().hm_gemm_set_variant(int(os.environ.get("VARIANT", 8))))
torch.manual_seed(0)
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
b = torch.randn(N, device="cuda")
f32 = epi in (2, 3)
out = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
res = torch.randn(M, N, device="cuda") if epi == 2 else None
for _ in range(int(os.environ.get("REPS", 5))):
    ops.gemm(x, w, b, epi, resid=res, out=out)
torch.cuda.synchronize()
print("done")
