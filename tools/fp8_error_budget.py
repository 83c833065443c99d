#!/usr/bin/env python3
"""Error budget of BASELINE configs[4] ("fp8 ViT-H weights"), per operand: distance of theta / beta /
vertices from the fp32 path (oracle.hamer_ref, fp32 master weights, ViT-H geometry, seeded crops) when e4m3 is switched on
operand class by operand class.  CPU emulation with the rounding points of the HIP kernels:
  weights  : e4m3 with one fp32 scale per output channel (oracle/fp8_ref.quantize_weight) on the named GEMMs
  inputs   : MXFP8 (e4m3 elements, one E8M0 scale per 32 K-elements; fake_quant_mx8) on the named GEMM inputs
  the rest : the 16-bit operand type given by BASE (fp16 | bf16), as emu=BASE of the oracle.
Usage: python tools/fp8_error_budget.py [n_crops=4] [BASE=fp16]     -> one JSON line per configuration (also appended to
gpurun_out/parity_report.jsonl when that directory exists)."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamer_yolo_amd import synth
from oracle import fp8_ref as Q
from oracle import hamer_ref as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
BASE = sys.argv[2] if len(sys.argv) > 2 else "fp16"
torch.set_num_threads(len(os.sched_getaffinity(0)))
cfg = synth.HamerConfig()
v = cfg.vit
sd = synth.hamer_state_dict(cfg, seed=0)
mp = synth.mano_params(seed=0)
img = synth.normalize_crops(synth.crops_u8(n, seed0=0))
_wcache = {}


def lin(x, name, W8, X8):
    """x . W^T + b for the GEMM `name` (e.g. blocks.3.attn.qkv) with its weight in e4m3 (W8) and / or its input in MXFP8 (X8)."""
    w, b = sd["backbone." + name + ".weight"], sd["backbone." + name + ".bias"]
    kind = name.split(".")[-1]
    if kind in W8:
        if name not in _wcache:
            w8, ws = Q.quantize_weight(w)
            _wcache[name] = Q.dequantize_weight(w8, ws)
        wq = _wcache[name]
    else:
        wq = R._q(w, BASE)
    xq = Q.fake_quant_mx8(x) if kind in X8 else R._q(x, BASE)
    return F.linear(xq, wq, b)


def forward(W8, X8):
    D = v.embed_dim
    x = img[:, :, :, 32:-32]
    t = R.patch_embed(sd, x, v, BASE)
    pos = sd["backbone.pos_embed"]
    t = t + pos[:, 1:] + pos[:, :1]
    B, N = t.shape[:2]
    for i in range(v.depth):
        p = f"blocks.{i}."
        h = F.layer_norm(t, (D,), sd["backbone." + p + "norm1.weight"], sd["backbone." + p + "norm1.bias"], v.ln_eps)
        qkv = R._q(lin(h, p + "attn.qkv", W8, X8), BASE).reshape(B, N, 3, v.heads, -1).permute(2, 0, 3, 1, 4)
        q, k, vv = qkv[0], qkv[1], qkv[2]
        s = (q @ k.transpose(-2, -1)) * v.head_dim ** -0.5
        pun = torch.exp(s - s.amax(dim=-1, keepdim=True))
        o = ((R._q(pun, BASE) @ vv) / pun.sum(dim=-1, keepdim=True)).transpose(1, 2).reshape(B, N, -1)
        t = t + lin(o, p + "attn.proj", W8, X8)
        h = F.layer_norm(t, (D,), sd["backbone." + p + "norm2.weight"], sd["backbone." + p + "norm2.bias"], v.ln_eps)
        y = F.gelu(lin(h, p + "mlp.fc1", W8, X8))
        t = t + lin(y, p + "mlp.fc2", W8, X8)
    feats = F.layer_norm(t, (D,), sd["backbone.last_norm.weight"], sd["backbone.last_norm.bias"], v.ln_eps)
    pose, betas, cam = R.mano_head_forward(sd, R._q(feats, BASE), cfg.dec, BASE)
    rot = R.rot6d_to_rotmat(pose).reshape(-1, 16, 3, 3)
    verts, _ = R.mano_forward(mp, betas, rot)
    return pose, betas, rot, verts


ALLW = ("qkv", "proj", "fc1", "fc2")
CONFIGS = [
    ("16-bit only (%s)" % BASE, (), ()),
    ("e4m3 weights: qkv fc1 fc2 proj; 16-bit activations", ALLW, ()),
    ("e4m3 weights: fc1 fc2 only", ("fc1", "fc2"), ()),
    ("e4m3 weights: qkv proj only", ("qkv", "proj"), ()),
    ("all weights + MXFP8 qkv / fc1 inputs (LayerNorm outputs)", ALLW, ("qkv", "fc1")),
    ("all weights + MXFP8 fc2 input (GELU output)", ALLW, ("fc2",)),
    ("all weights + MXFP8 proj input (attention output)", ALLW, ("proj",)),
    ("all weights + all four MXFP8 inputs (the shipped configs[4] engine)", ALLW, ALLW),
    ("16-bit weights, all four MXFP8 inputs", (), ALLW),
]
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
with torch.no_grad():
    ref = R.hamer_forward(sd, mp, img, cfg)
    rref = torch.cat([ref["global_orient"], ref["hand_pose"]], 1)
    for name, W8, X8 in CONFIGS:
        pose, betas, rot, verts = forward(W8, X8)
        rec = {"test": "fp8_error_budget", "base": BASE, "crops": n, "config": name,
               "pose6d": float((pose - ref["pose6d"]).abs().max()), "betas": float((betas - ref["betas"]).abs().max()),
               "rotmats": float((rot - rref).abs().max()), "vertices": float((verts - ref["pred_vertices"]).abs().max())}
        rec["meets_1e-3_theta_beta_vertices"] = bool(max(rec["betas"], rec["rotmats"], rec["vertices"]) < 1e-3)
        print(json.dumps(rec), flush=True)
        if os.path.isdir(out_dir):
            with open(os.path.join(out_dir, "parity_report.jsonl"), "a") as f:
                f.write(json.dumps(rec) + "\n")
