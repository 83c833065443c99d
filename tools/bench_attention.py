"""Attention kernel timing at several batch sizes (workgroups = 16 x B); HM_ATT_MODE selects the phase ablation."""
import sys, torch
sys.path.insert(0, ".")
from hamer_yolo_amd import ops, lib as L
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'tools'))
from runlog import banner
banner()
for B in (4, 8, 16, 32, 48, 64, 128):
    qkv = (torch.randn(B * 192, 3840, device="cuda") * 0.5).to(torch.bfloat16)
    for _ in range(5):
        ops.vit_attention(qkv, B, 192, 16, 80, 80 ** -0.5)
    torch.cuda.synchronize()
    with L.profile(capacity=64) as prof:
        for _ in range(20):
            ops.vit_attention(qkv, B, 192, 16, 80, 80 ** -0.5)
        torch.cuda.synchronize()
    ms = sorted(r[-1] for r in prof.records)
    print(f"B={B:4d} blocks={16*B:5d} median {1e3*ms[len(ms)//2]:7.1f} us", flush=True)
