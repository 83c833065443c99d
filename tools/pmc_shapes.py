#!/usr/bin/env python3
"""Launch every GEMM shape of one HaMeR step (B=64, fp16 operands, default tile choice) twice, plus LayerNorm and attention,
in a fixed order; run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` / `--pmc SQ_VALU_MFMA_BUSY_CYCLES
GRBM_GUI_ACTIVE` (separate passes) to get traffic beyond L2 and MFMA-pipe busy cycles per launch.  tools/pmc_parse.py turns the
counter CSVs into profiles/rNN_pmc_traffic.json / rNN_pmc_mfma_busy.json (tools/collect_profiles.sh drives all of it)."""
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

M = 64 * 192
DT = torch.float16
SHAPES = [("patch", 768, 1280, L.HM_EPI_RESID_F32), ("qkv", 1280, 3840, L.HM_EPI_STORE), ("proj", 1280, 1280, L.HM_EPI_RESID_F32),
          ("fc1", 1280, 5120, L.HM_EPI_GELU), ("fc2", 5120, 1280, L.HM_EPI_RESID_F32), ("kv", 1280, 6144, L.HM_EPI_STORE)]
KERNELS = ("gemm_tn_kernel", "gemm_x3_kernel", "gemm_x3r_kernel", "gemm_px_kernel", "layernorm_rows_kernel", "layernorm_kernel", "vit_attention_kernel")
if __name__ == "__main__":
    torch.manual_seed(0)
    bufs = []
    for (name, K, N, epi) in SHAPES:     # distinct buffers per shape, all allocated first
        f32 = epi == L.HM_EPI_RESID_F32
        bufs.append((torch.randn(M, K, device="cuda").to(DT), (torch.randn(N, K, device="cuda") * 0.02).to(DT), torch.randn(N, device="cuda"),
                     torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else DT),
                     torch.randn(M, N, device="cuda") if f32 else None, epi))
    x = torch.randn(M, 1280, device="cuda"); g = torch.randn(1280, device="cuda"); qkv = torch.randn(M, 3840, device="cuda").to(DT)
    torch.cuda.synchronize()
    for rep in range(2):
        for (xx, w, b, out, res, epi) in bufs:
            ops.gemm(xx, w, b, epi, resid=res, out=out)
        ops.layernorm(x, g, g, 1e-6, DT)
        ops.vit_attention(qkv, 64, 192, 16, 80, 80 ** -0.5)
    torch.cuda.synchronize()
    print("done")
