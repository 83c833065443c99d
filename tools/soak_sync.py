"""Race screen for the hand-synchronised kernels (gemm_x3_kernel: counted vmcnt over a 3-slot ring; vit_attention_kernel:
hidden LDS-DMA double buffer): many launches on several shapes, every result compared bit for bit with the first and, for
the GEMM on exact-integer data, with the exact answer."""
import sys, torch
sys.path.insert(0, ".")
from hamer_yolo_amd import ops, lib as L
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tools'))
from runlog import banner
banner()
lib = L.load()
bad = 0
L.check(lib.hm_gemm_set_variant(24))
for (M, N, K) in ((12288, 1280, 1280), (12288, 3840, 1280), (12288, 1280, 5120), (3000, 516, 192), (257, 260, 64), (5000, 1284, 448)):
    x = (torch.arange(M * K, dtype=torch.int64).reshape(M, K) * 7 % 9 - 4).float()
    w = ((torch.arange(N * K, dtype=torch.int64).reshape(N, K) * 5 + torch.arange(N)[:, None]) % 7 - 3).float()
    ref = (x.cuda() @ w.cuda().t()).cpu()
    xd, wd = x.cuda().bfloat16(), w.cuda().bfloat16()
    res = torch.zeros(M, N, device="cuda")
    for it in range(150):
        out = ops.gemm(xd, wd, None, L.HM_EPI_RESID_F32, resid=res)
        if it % 10 == 0 or it > 140:
            if not torch.equal(out.cpu(), ref):
                bad += 1
                print("GEMM MISMATCH", M, N, K, it, flush=True)
    print("gemm", M, N, K, "ok", flush=True)
lib.hm_gemm_set_variant(-1)
for B in (64, 7, 128):
    qkv = (torch.randn(B * 192, 3840, device="cuda") * 0.7).bfloat16()
    first = ops.vit_attention(qkv, B, 192, 16, 80, 80 ** -0.5).clone()
    for it in range(150):
        o = ops.vit_attention(qkv, B, 192, 16, 80, 80 ** -0.5)
        if not torch.equal(o, first):
            bad += 1
            print("ATTENTION MISMATCH", B, it, flush=True)
    print("attention", B, "ok", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
