#!/usr/bin/env python3
"""Kernel microbenchmarks on the ViT-H shapes at B=64 (random data): GEMM per epilogue, LayerNorm,
attention.  Per-launch hipEvent timings through the library profiler.  Usage: python tools/bench_kernels.py [reps]"""
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

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda"
M = 64 * 192
torch.manual_seed(0)


def rnd(*shape, dt=torch.bfloat16, s=1.0):
    return (torch.randn(*shape, device=dev) * s).to(dt)


def timeit(name, fn, flops=None, nbytes=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    with L.profile(capacity=reps * 4) as prof:
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    ms = sorted(r[5] for r in prof.records)
    med = ms[len(ms) // 2]
    extra = ""
    if flops:
        extra += f"  {flops / med / 1e9:8.1f} TFLOP/s"
    if nbytes:
        extra += f"  {nbytes / med / 1e9:7.2f} TB/s"
    print(f"{name:34s} median {med * 1e3:8.1f} us  min {ms[0] * 1e3:8.1f} us{extra}", flush=True)


variants = [int(v) for v in os.environ.get("VARIANTS", "0").split(",")]
for dt in (torch.float16,):
  for variant in variants:
    L.check(L.load().hm_gemm_set_variant(variant))
    print(f"--- gemm tile variant {variant}", flush=True)
    for (name, K, N, epi) in (("qkv    K1280 N3840 store", 1280, 3840, L.HM_EPI_STORE), ("proj   K1280 N1280 resid", 1280, 1280, L.HM_EPI_RESID_F32),
                              ("fc1    K1280 N5120 gelu", 1280, 5120, L.HM_EPI_GELU), ("fc2    K5120 N1280 resid", 5120, 1280, L.HM_EPI_RESID_F32),
                              ("kv     K1280 N6144 store", 1280, 6144, L.HM_EPI_STORE), ("patch  K768  N1280 resid", 768, 1280, L.HM_EPI_RESID_F32)):
        x, w, b = rnd(M, K, dt=dt), rnd(N, K, dt=dt, s=0.02), torch.randn(N, device=dev)
        f32 = epi in (L.HM_EPI_RESID_F32, L.HM_EPI_F32)
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else dt)
        res = torch.randn(M, N, device=dev) if epi == L.HM_EPI_RESID_F32 else None
        timeit("gemm " + name, lambda: ops.gemm(x, w, b, epi, resid=res, out=out), flops=2.0 * M * N * K)

x = torch.randn(M, 1280, device=dev)
g, bb = torch.randn(1280, device=dev), torch.randn(1280, device=dev)
timeit("layernorm 12288x1280 -> fp16", lambda: ops.layernorm(x, g, bb, 1e-6, torch.float16), nbytes=M * 1280 * 6)
qkv = rnd(M, 3840, dt=torch.float16)
timeit("attention B64 H16 T192 d80", lambda: ops.vit_attention(qkv, 64, 192, 16, 80, 80 ** -0.5), flops=64 * 16 * 4.0 * 192 * 192 * 80, nbytes=M * 5120 * 2)
