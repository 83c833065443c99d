"""Feasibility probe for a CU-partitioned forward: do the MFMA GEMMs keep their speed on 240 of the 256 CUs (their tile counts --
240 / 720 / 960 at B = 64 -- are whole rounds of 240) while the HBM-bound kernels (LayerNorm, attention) of ANOTHER batch run
beside them on the remaining 16 (2 per XCD), and how fast are those kernels on 16 CUs?
Streams with CU masks come from hipExtStreamCreateWithCUMask (ctypes on libamdhip64), wrapped as torch ExternalStreams.
The memory-bound stream gets MCU / 8 CUs of every XCC (MCU=16 by default), the GEMM stream the rest."""
import ctypes as C
import os
import re
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamer_yolo_amd import lib as L
from hamer_yolo_amd import ops
from runlog import banner

banner()
MCU = int(os.environ.get("MCU", 16))
# bit b of the mask <-> XCC b % 8, CU slot b / 8 of that XCC (profiles/r03_cu_mask_map.txt: tools/probes/cu_mask_map.bin; an XCC
# whose bits are all zero is left unrestricted by the runtime, so every XCC must keep at least one bit)
mem_bits = list(range(MCU))                     # MCU / 8 CUs of every XCC
gemm_bits = list(range(MCU, 256))
print(f"memory-bound stream: {len(mem_bits)} CUs (mask bits 0..{MCU - 1}); GEMM stream: {len(gemm_bits)} CUs")
hip = C.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


torch.zeros(1, device="cuda")
sg, sm = masked_stream(gemm_bits), masked_stream(mem_bits)
lib = L.load()
B, T, D, H = 64, 192, 1280, 16
M = B * T
dt = torch.float16
torch.manual_seed(0)
x32 = torch.randn(M, D, device="cuda")
g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
h16 = torch.randn(M, D, device="cuda").to(dt)
mlp16 = torch.randn(M, 4 * D, device="cuda").to(dt)
qkv16 = torch.randn(M, 3 * D, device="cuda").to(dt)
W = {n: (torch.randn(nn, k, device="cuda") * 0.02).to(dt) for n, (nn, k) in {"qkv": (3 * D, D), "proj": (D, D), "fc1": (4 * D, D), "fc2": (D, 4 * D)}.items()}
bias = {n: torch.randn(w.shape[0], device="cuda") for n, w in W.items()}
o_qkv, o_fc1 = torch.empty(M, 3 * D, device="cuda", dtype=dt), torch.empty(M, 4 * D, device="cuda", dtype=dt)
xr = torch.randn(M, D, device="cuda")


def block_gemms():
    ops.gemm(h16, W["qkv"], bias["qkv"], L.HM_EPI_STORE, out=o_qkv)
    ops.gemm(h16, W["proj"], bias["proj"], L.HM_EPI_RESID_F32, resid=xr, out=xr)
    ops.gemm(h16, W["fc1"], bias["fc1"], L.HM_EPI_GELU, out=o_fc1)
    ops.gemm(mlp16, W["fc2"], bias["fc2"], L.HM_EPI_RESID_F32, resid=xr, out=xr)


def block_mem():
    ops.layernorm(x32, g, b, 1e-6, out_dtype=dt)
    ops.vit_attention(qkv16, B, T, H, D // H, (D // H) ** -0.5)
    ops.layernorm(x32, g, b, 1e-6, out_dtype=dt)


def timed(fn, stream, n):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(n):
            fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


cur = torch.cuda.current_stream()
print(f"GEMMs of one block, whole chip (256 workgroups):      {timed(block_gemms, cur, 40):8.1f} us")
L.check(lib.hm_set_option(L.HM_OPT_PX_GRID, (256 - MCU) // 8 * 8))
print(f"GEMMs of one block, whole chip, persistent grid 256-MCU:  {timed(block_gemms, cur, 40):8.1f} us")
print(f"GEMMs of one block, masked to {len(gemm_bits)} CUs, same grid:      {timed(block_gemms, sg, 40):8.1f} us")
print(f"LN + attention + LN of one block, whole chip:         {timed(block_mem, cur, 40):8.1f} us")
print(f"LN + attention + LN of one block, masked to {len(mem_bits)} CUs:     {timed(block_mem, sm, 20):8.1f} us")
print(f"LayerNorm alone, masked to {len(mem_bits)} CUs:                      {timed(lambda: ops.layernorm(x32, g, b, 1e-6, out_dtype=dt), sm, 20):8.1f} us")
print(f"attention alone, masked to {len(mem_bits)} CUs:                      {timed(lambda: ops.vit_attention(qkv16, B, T, H, D // H, (D // H) ** -0.5), sm, 20):8.1f} us")
# both at once: n blocks of GEMMs on the GEMM stream, memory blocks on the other one until the GEMMs are done
for nmem in (0, 1):
    torch.cuda.synchronize()
    n = 40
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sg):
        e0.record()
        for _ in range(n):
            block_gemms()
        e1.record()
    k = 0
    if nmem:
        with torch.cuda.stream(sm):
            m0.record()
            for _ in range(n):
                block_mem(); k += 1
            m1.record()
    torch.cuda.synchronize()
    msg = f"concurrent: GEMM stream {e0.elapsed_time(e1) / n * 1e3:8.1f} us per block"
    if nmem:
        msg += f" | memory stream {m0.elapsed_time(m1) / k * 1e3:8.1f} us per block"
    print(msg)
L.check(lib.hm_set_option(L.HM_OPT_PX_GRID, 0))
