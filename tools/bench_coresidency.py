"""Can an HBM-bound LayerNorm run UNDER a GEMM of another stream?  GEMM (fc2 shape) on stream A, LayerNorm launches on stream
B, together vs apart, for GEMM tile variants with different register footprints (10: 206 VGPRs, 24: 248, 0: 117 x 2 blocks)."""
import sys, time, torch
sys.path.insert(0, ".")
from hamer_yolo_amd import ops, lib as L
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tools'))
from runlog import banner
banner()
lib = L.load()
M = 12288
x = torch.randn(M, 5120, device="cuda").bfloat16(); w = (torch.randn(1280, 5120, device="cuda") * 0.02).bfloat16()
b = torch.randn(1280, device="cuda"); res = torch.randn(M, 1280, device="cuda")
xl = torch.randn(M, 1280, device="cuda"); g = torch.randn(1280, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
import os
STORE = os.environ.get("STORE", "0") == "1"
o16 = torch.empty(M, 1280, device="cuda", dtype=torch.bfloat16)
def run(gemm_n, ln_n, reps=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        with torch.cuda.stream(sa):
            for _ in range(gemm_n):
                if STORE: ops.gemm(x, w, b, L.HM_EPI_STORE, out=o16)
                else: ops.gemm(x, w, b, L.HM_EPI_RESID_F32, resid=res, out=res)
        with torch.cuda.stream(sb):
            for _ in range(ln_n): ops.layernorm(xl, g, g, 1e-6, torch.bfloat16)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
for v in (10, 24, 0):
    L.check(lib.hm_gemm_set_variant(v))
    run(4, 8, 3)
    tg, tl, tb = run(4, 0), run(0, 8), run(4, 8)
    print(f"variant {v:2d}: 4 GEMMs {tg:7.1f} us | 8 LayerNorms {tl:7.1f} us | together {tb:7.1f} us (sum {tg+tl:7.1f})", flush=True)
lib.hm_gemm_set_variant(-1)
