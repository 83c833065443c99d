#!/usr/bin/env python3
"""CPU study behind DESIGN.md's precision table: distance of the 16-bit operand paths (bf16 / fp16 emulation of
oracle.hamer_ref, same rounding points as the HIP kernels) from the fp32 path, on fp32 master weights, ViT-H geometry.
Usage: python tools/precision_study.py [n_crops]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hamer_yolo_amd import synth
from oracle import hamer_ref as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.set_num_threads(len(os.sched_getaffinity(0)))
cfg = synth.HamerConfig()
sd = synth.hamer_state_dict(cfg, seed=0)
mp = synth.mano_params(seed=0)
img = synth.normalize_crops(synth.crops_u8(n, seed0=0))
with torch.no_grad():
    ref = R.hamer_forward(sd, mp, img, cfg)
    for emu in ("bf16", "fp16"):
        o = R.hamer_forward(sd, mp, img, cfg, emu=emu)
        rot = lambda d: torch.cat([d["global_orient"], d["hand_pose"]], 1)
        print(emu, " ".join(f"{k}={float((a - b).abs().max()):.2e}" for k, a, b in (
            ("pose6d", o["pose6d"], ref["pose6d"]), ("betas", o["betas"], ref["betas"]), ("cam", o["pred_cam"], ref["pred_cam"]),
            ("rotmats", rot(o), rot(ref)), ("verts", o["pred_vertices"], ref["pred_vertices"]),
            ("kp3d", o["pred_keypoints_3d"], ref["pred_keypoints_3d"]))), flush=True)
