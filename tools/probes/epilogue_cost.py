"""fc1-shaped GEMM (M=12288, N=5120, K=1280, fp16) with three epilogues, interleaved in one process: plain store, SiLU
(exp + rcp per value), GELU (packed erf form).  What does the activation cost on top of the store epilogue?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hamer_yolo_amd import lib as L, ops
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))), 'tools'))
from runlog import banner
banner()

M, N, K = 12288, 5120, 1280
torch.manual_seed(0)
x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * 0.02).half(); b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.float16)
epis = {"store": L.HM_EPI_STORE, "silu": L.HM_EPI_SILU, "gelu": L.HM_EPI_GELU}
for v in (26, 24):
    L.check(L.load().hm_gemm_set_variant(v))
    times = {k: [] for k in epis}
    for k, e in epis.items():
        ops.gemm(x, w, b, e, out=out)
    torch.cuda.synchronize()
    for _ in range(8):
        for k, e in epis.items():
            with L.profile(capacity=8) as prof:
                for _ in range(4):
                    ops.gemm(x, w, b, e, out=out)
                torch.cuda.synchronize()
            times[k] += [r[5] for r in prof.records]
    print("variant", v, {k: "%.1f us" % (sorted(t)[len(t) // 2] * 1e3) for k, t in times.items()})
L.load().hm_gemm_set_variant(-1)
