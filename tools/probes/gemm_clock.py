"""Which clock does the chip hold inside the persistent GEMM, and how many shader cycles does a K-step take?  Diagnostic build
(experiments library, variant 35: gemm_px_kernel<..., STAMP>): every workgroup stamps s_memtime (shader cycles) and
s_memrealtime (100 MHz) around the kernel and around its K loops.  Launched back to back for >= 2 s first, so the chip is at the
clock it holds under this load (MI355X_MICROARCH.md, DVFS give-back item 6).  Usage: python tools/probes/gemm_clock.py"""
import os, sys, time
sys.path.insert(0, ".")
import torch
from hamer_yolo_amd import lib as L
L.LIB_PATH = L.LIB_PATH.replace(".so", "_abl.so")
from hamer_yolo_amd import ops
lib = L.load()
M = 64 * 192
torch.manual_seed(0)
for name, K, N in (("qkv", 1280, 3840), ("fc1", 1280, 5120), ("kv", 1280, 6144), ("fc2-shape, store epilogue", 5120, 1280)):
    x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * 0.02).half(); b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    stamps = torch.zeros(256 * 6, dtype=torch.int64, device="cuda")
    L.check(lib.hm_gemm_set_variant(26))
    t0 = time.time()
    while time.time() - t0 < 2.0:                      # warm the clock governor with the production kernel
        for _ in range(50):
            ops.gemm(x, w, b, L.HM_EPI_STORE, out=out)
        torch.cuda.synchronize()
    L.check(lib.hm_gemm_set_variant(35))
    for _ in range(20):
        ops.gemm(x, w, b, L.HM_EPI_STORE, out=out, ln_stats=stamps)
    torch.cuda.synchronize()
    s = stamps.cpu().reshape(256, 6).double()
    s = s[s[:, 5] > 0]
    clk = (s[:, 0] / s[:, 1]).median().item() * 100e6
    kclk = (s[:, 2] / s[:, 3]).median().item() * 100e6
    cyc_per_step = (s[:, 2] / s[:, 4]).median().item()
    us_per_step = (s[:, 3] / s[:, 4]).median().item() / 100.0
    print(f"{name:28s} K={K:5d} N={N:5d}: clock {clk / 1e9:.3f} GHz (K loops {kclk / 1e9:.3f}), {cyc_per_step:7.1f} shader cycles = {us_per_step:.3f} us per K-step "
          f"-> MFMA pipe {2048.0 / cyc_per_step * 100:.1f} % busy in the K loop (2048 MFMA cycles per step and SIMD); {int(s.shape[0])} workgroups, {s[:, 5].median().item():.0f} tiles each")
L.check(lib.hm_gemm_set_variant(-1))
