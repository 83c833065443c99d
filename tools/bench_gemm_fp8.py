"""hm_gemm_fp8 on the ViT-H shapes vs the bf16 kernel (random data, per-launch hipEvent timings)."""
import sys, torch
sys.path.insert(0, ".")
from hamer_yolo_amd import ops, lib as L
from oracle import fp8_ref as Q   # (tool only: builds test operands)
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tools'))
from runlog import banner
banner()
torch.manual_seed(0)
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    with L.profile(capacity=4 * n) as prof:
        for _ in range(n): fn()
        torch.cuda.synchronize()
    ms = sorted(r[-1] for r in prof.records)
    return ms[len(ms) // 2]
for B in (64, 256):
    M = B * 192
    for name, K, N, epi8, epi16 in (("qkv", 1280, 3840, L.HM_EPI_STORE, L.HM_EPI_STORE), ("fc1", 1280, 5120, L.HM_EPI_GELU_MX8, L.HM_EPI_GELU),
                                    ("fc2", 5120, 1280, L.HM_EPI_RESID_F32, L.HM_EPI_RESID_F32)):
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        x8 = (x * 8).to(torch.float8_e4m3fn).view(torch.uint8)
        xs = torch.full((K // 32, M), 124, device="cuda", dtype=torch.uint8)
        w8 = (w * 100).to(torch.float8_e4m3fn).view(torch.uint8)
        ws = torch.full((N,), 0.01, device="cuda")
        bias = torch.randn(N, device="cuda")
        res = torch.randn(M, N, device="cuda") if epi8 == L.HM_EPI_RESID_F32 else None
        t8 = timeit(lambda: ops.gemm_fp8(x8, xs, w8, ws, bias, epi8, resid=res, out=res if res is not None else None))
        xb, wb = x.bfloat16(), w.bfloat16()
        t16 = timeit(lambda: ops.gemm(xb, wb, bias, epi16, resid=res, out=res if res is not None else None))
        fl = 2.0 * M * N * K
        print(f"B={B:3d} {name}: fp8 {1e3*t8:7.1f} us {fl/t8/1e9:7.0f} TF | bf16 {1e3*t16:7.1f} us {fl/t16/1e9:7.0f} TF | x{t16/t8:.2f}", flush=True)
