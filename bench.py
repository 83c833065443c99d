#!/usr/bin/env python3
"""bench.py -- hands/sec of the HaMeR hot path on MI355X (BASELINE.json configs[1]).

A step = one forward of hm_hamer_forward (patch gather, ViT-H/16 backbone, transformer-decoder
MANO head, rot6d + MANO LBS + projection) over one batch of 64 synthetic 256x256 crops that are
already resident in HBM, 16-bit MFMA GEMMs (fp16 operands: the type that meets the 1e-3 parity bar
on fp32 master weights; --dtype bf16 selects the other), detector bypassed.  With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank processes its own 64-crop shard (weak
scaling); RCCL carries the one-off weight broadcast and the per-step gather of per-hand MANO
parameters to rank 0.

Prints ONE JSON line on rank 0 (see the field list in DESIGN.md section "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from hamer_yolo_amd import lib as L  # noqa: E402
from hamer_yolo_amd import shard, synth  # noqa: E402
from hamer_yolo_amd.engine import HamerEngine  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16/fp16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
EPI_NAMES = {0: "store", 1: "gelu", 2: "resid_f32", 3: "f32", 4: "silu", 5: "resid_ln", 6: "ln_store", 7: "ln_gelu",
             16: "fp8_store", 18: "fp8_resid_f32", 24: "fp8_gelu_mx8"}
PEAK_FP8_TFLOPS = 5000.0       # dense e4m3 on the block-scaled 16x16x128 MFMA (MI355X_MICROARCH.md)


def flops_per_hand(cfg: synth.HamerConfig) -> dict:
    v, d = cfg.vit, cfg.dec
    T, D, H = v.tokens, v.embed_dim, v.embed_dim * v.mlp_ratio
    blk = 2 * T * D * 3 * D + 2 * 2 * T * T * D + 2 * T * D * D + 2 * 2 * T * D * H
    patch = 2 * T * 3 * v.patch * v.patch * D
    kv = 2 * T * d.context_dim * d.depth * 2 * d.inner
    return {"vit": patch + v.depth * blk, "decoder_kv": kv, "total_mfma": patch + v.depth * blk + kv}


def host_threads() -> int:
    """Host cores this process may really use: HAMER_CPU_THREADS, else the cgroup CPU quota,
    else min(affinity, 16) (a one-GPU box's share of the host is 16 cores)."""
    if os.environ.get("HAMER_CPU_THREADS"):
        return max(1, int(os.environ["HAMER_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def cpu_baseline(cfg, sd_dev, mano_cpu, seconds: float = 12.0):
    """The oracle (CPU restatement of the reference path, fp32 PyTorch) timed on the host cores."""
    from oracle import hamer_ref as R
    threads = host_threads()
    torch.set_num_threads(threads)
    sd = {k: v.float().cpu() for k, v in sd_dev.items()}
    B = 8
    img = synth.normalize_crops(synth.crops_u8(B, seed0=0))
    with torch.no_grad():
        R.hamer_forward(sd, mano_cpu, img[:1], cfg)      # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            R.hamer_forward(sd, mano_cpu, img, cfg)
            n += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or n >= 6:
                break
    return {"value": round(n * B / dt, 3), "unit": "hands/s", "cores": threads, "kind": "port",
            "sample": f"{n} forwards of B={B} crops (ViT-H/16 + decoder + MANO), fp32 torch CPU oracle, {dt:.1f} s"}


def run_e2e(args, dev, dtype):
    """BASELINE configs[2]: seeded 1080p frames; every frame runs letterbox -> YOLOv7 -> decode -> NMS (timed in
    full), then 4 fixed boxes (2 left, 2 right) are substituted for its output (a random-weight detector finds
    nothing meaningful) and go through crop -> HaMeR -> MANO.  One step = `frames` frames."""
    import numpy as np
    from hamer_yolo_amd import ops
    from hamer_yolo_amd.yolo.engine import YoloEngine
    cfg = synth.HamerConfig()
    F = args.frames
    nfl = args.in_flight if args.in_flight > 0 else 3
    ysd = synth.yolo_state_dict(seed=0, nc=3)
    yolos = [YoloEngine(ysd, nc=3, device=dev) for _ in range(nfl)]      # one activation arena per batch in flight
    eng = HamerEngine(synth.hamer_state_dict(cfg, seed=0, device=dev), synth.mano_params(seed=0), cfg,
                      device=dev, dtype=dtype)
    frames = [synth.frame_u8(1080, 1920, seed=i).to(dev) for i in range(F)]
    boxes = [(400.0, 300.0, 220.0, True), (1500.0, 320.0, 180.0, False), (700.0, 800.0, 260.0, True), (1200.0, 760.0, 160.0, False)]
    rec = ops.crop_boxes([(cx, cy, s * 10.0 / 3.0, fl) for cx, cy, s, fl in boxes]).to(dev)
    mean = 255.0 * np.array([0.485, 0.456, 0.406]); std = 255.0 * np.array([0.229, 0.224, 0.225])
    imgs = [torch.empty(4 * F, 3, 256, 256, device=dev) for _ in range(nfl)]
    ctxs = eng.contexts(4 * F, nfl)
    eng.workspace(4 * F)
    torch.cuda.synchronize()
    nstep = [0]

    def step(k=None):
        k = nstep[0] % nfl if k is None else k
        nstep[0] += 1
        c, yolo, img = ctxs[k], yolos[k], imgs[k]
        with torch.cuda.stream(c.stream):                            # steps alternate between the contexts and overlap
            p = yolo.forward(frames)                                 # one batched pass over all frames of the step
            yolo.nms_enqueue(p, 0.25, 0.35, [0, 1, 2], True)         # box lists stay on the device: no host sync
            for i, fr in enumerate(frames):
                img[4 * i:4 * i + 4] = ops.crop_batch(fr, rec, mean, std)
            eng.forward(img, c.out, workspace=c.workspace)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    with L.profile(capacity=4096 * 4) as prof:
        step(0)
        torch.cuda.synchronize()
    by = {}
    for kind, epi, M, N, K, ms in prof.records:
        by[kind] = by.get(kind, 0.0) + ms
    print(json.dumps({"metric": "hands/sec end-to-end (YOLOv7 + crop + HaMeR + MANO), 1080p frames, 4 hands/frame",
                      "value": round(4 * F * args.steps / el, 2), "unit": "hands/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": round(1e3 * el / args.steps, 3), "frames_per_step": F, "frames_per_s": round(F * args.steps / el, 2),
                      "dtype": args.dtype + " (HaMeR) / fp16 (YOLOv7)", "data": "synthetic",
                      "config": {"workload": "BASELINE configs[2]: 1080p frames, YOLOv7 + 4 fixed boxes/frame + HaMeR",
                                 "batches_in_flight": nfl},
                      "gflop_per_frame": 61.9 + 4 * 251.03, "ms_per_step_by_kernel": {k: round(v, 3) for k, v in sorted(by.items(), key=lambda kv: -kv[1])}}), flush=True)


def mfma_busy_pmc(cfg):
    """MFMA-pipe busy fraction of the GEMM launches of one step from the committed PMC pass (rocprofv3 --pmc
    SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over tools/pmc_shapes.py; profiles/r01_pmc_mfma_busy.json), cycle-weighted."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_mfma_busy.json")
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    w = {"patch": 1, "qkv": cfg.vit.depth, "proj": cfg.vit.depth, "fc1": cfg.vit.depth, "fc2": cfg.vit.depth, "kv": 1}
    busy = sum(t[k]["SQ_VALU_MFMA_BUSY_CYCLES"] * n for k, n in w.items())
    cyc = sum(t[k]["shader_cycles"] * 1024 * n for k, n in w.items())
    return round(busy / cyc, 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="crops per GPU per step (BASELINE config: 64)")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16", "fp8"],
                    help="fp8: BASELINE configs[4] (qkv/fc1/fc2 on the fp8 MFMA, MXFP8 activations); use with --batch 256")
    ap.add_argument("--workload", default="crops", choices=["crops", "e2e"],
                    help="crops: BASELINE configs[1] (default, the contract line); e2e: configs[2], 1080p frames through "
                         "YOLOv7 + crop + HaMeR with 4 fixed boxes per frame (not a contract line, for DESIGN.md)")
    ap.add_argument("--frames", type=int, default=16, help="e2e: frames per step (hands per step = 4 x frames)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="batches in flight: consecutive steps alternate between this many HIP streams (own workspace and outputs "
                         "each), so one batch's HBM-bound phases overlap another's MFMA phases; 1 = strictly one after the other; "
                         "default 2 (crops) / 3 (e2e)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    rank, local, world = shard.init_distributed("nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = torch.distributed
    cfg = synth.HamerConfig()
    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    B = args.batch

    if args.workload == "e2e":
        return run_e2e(args, dev, dtype)

    # weights: rank 0 draws the synthetic checkpoint, RCCL broadcasts it
    sd0 = synth.hamer_state_dict(cfg, seed=0, device=dev) if rank == 0 else None
    if world > 1:
        meta =[{k: tuple(v.shape) for k, v in sd0.items()}] if rank == 0 else [None]
        dist.broadcast_object_list(meta, src=0)
        sd = shard.broadcast_state_dict(sd0, list(meta[0].keys()), meta[0], dev, src=0)
    else:
        sd = sd0
    mano_cpu = synth.mano_params(seed=0)
    eng = HamerEngine(sd, mano_cpu, cfg, device=dev, dtype=dtype, fp8=(args.dtype == "fp8"))

    # this rank's shard of the global crop set (seeds rank*B .. rank*B+B-1), resident in HBM
    img = synth.normalize_crops(synth.crops_u8(B, seed0=rank * B)).to(dev)
    out = eng.alloc_outputs(B)
    eng.workspace(B)
    ctxs = eng.contexts(B, args.in_flight if args.in_flight > 0 else 2)
    torch.cuda.synchronize()

    nstep = [0]

    def step():
        c = ctxs[nstep[0] % len(ctxs)]                   # every step is one whole batch; steps alternate between the contexts
        nstep[0] += 1
        with torch.cuda.stream(c.stream):
            eng.forward(img, c.out, workspace=c.workspace)
            if world > 1:
                shard.gather_mano(shard.pack_mano(c.out), dst=0)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    for c in ctxs:
        assert torch.isfinite(c.out["pred_vertices"]).all()

    res = None
    if rank == 0:
        hands = world * B * args.steps
        fl = flops_per_hand(cfg)
        res = {
            "metric": "hands/sec (HaMeR ViT-H/16 + decoder + MANO forward; detector bypassed)",
            "value": round(hands / elapsed, 2), "unit": "hands/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("BASELINE configs[4]: batch=%d synthetic 256x256 crops per GPU, HaMeR ViT-H/16 with qkv/fc1/fc2 "
                                    "in e4m3 on the block-scaled fp8 MFMA (MXFP8 activations), proj/attention bf16, fp32 residual, "
                                    "decoder and MANO, crops resident in HBM" % B) if args.dtype == "fp8" else
                                   "BASELINE configs[1]: batch=%d synthetic 256x256 crops per GPU, HaMeR ViT-H/16 "
                                   "+ 6-layer decoder + MANO, 16-bit MFMA, crops resident in HBM" % B,
                       "batch_per_gpu": B, "global_batch": world * B, "batches_in_flight": len(ctxs), "weights": "seeded random-init fp32 master weights, rounded to the operand type at load",
                       "parallelism": f"crop-shard x{world}", "mfma_gflop_per_hand": round(fl["total_mfma"] / 1e9, 2)},
            "model_mfma_frac": round(hands / elapsed * fl["total_mfma"] / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
        }

    # ---- roofline of the dominant kernel (the MFMA GEMM), HIP events on the launch stream
    if rank == 0 and not args.no_roofline:
        nprof = min(args.steps, 5)
        with L.profile(capacity=nprof * 512) as prof:
            for _ in range(nprof):
                eng.forward(img, out)                   # one stream: every launch timed alone, nothing overlapping it
            torch.cuda.synchronize()
        by_kind, gemm_fl, gemm_ms, per = {}, 0.0, 0.0, {}
        for kind, epi, M, N, K, ms in prof.records:
            by_kind[kind] = by_kind.get(kind, 0.0) + ms
            if kind == "gemm":
                gemm_fl += 2.0 * M * N * K
                gemm_ms += ms
                e = per.setdefault(EPI_NAMES[epi], [0, 0.0, 0.0])
                e[0] += 1; e[1] += ms; e[2] += 2.0 * M * N * K
        n_gemm = sum(v[0] for v in per.values())
        achieved = gemm_fl / (gemm_ms * 1e-3) / 1e12
        # HBM-side bytes per GEMM launch from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
        # runs over tools/pmc_shapes.py, gfx950 FETCH_SIZE x2 correction; tools/pmc_parse.py), weighted by launches per step
        traffic, traffic_src = None, os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(traffic_src) and B == 64:
            t = json.load(open(traffic_src))
            w = {"patch": 1, "qkv": cfg.vit.depth, "proj": cfg.vit.depth, "fc1": cfg.vit.depth, "fc2": cfg.vit.depth, "kv": 1}
            traffic = round(sum(t[k]["hbm_bytes"] * n for k, n in w.items()) / sum(w.values()))
        kernel_name, peak = "gemm_x3_kernel / gemm_tn_kernel (the 256x256 MFMA GEMM, all epilogues)", PEAK_BF16_TFLOPS
        if args.dtype == "fp8":      # dominant kernel = gemm_fp8_kernel: price its launches against the fp8 peak
            f8 = [(M_, N_, K_, ms_) for (kd, e_, M_, N_, K_, ms_) in prof.records if kd == "gemm" and e_ >= 16]
            achieved = sum(2.0 * a * b * c for a, b, c, _ in f8) / (sum(t for *_, t in f8) * 1e-3) / 1e12
            kernel_name, peak, traffic = "gemm_fp8_kernel (store / gelu_mx8 / resid_f32)", PEAK_FP8_TFLOPS, None
        res["roofline"] = {
            "kernel": kernel_name, "bound": "mfma", "achieved": round(achieved, 2),
            "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
            "traffic_note": "bytes per launch beyond L2 (FETCH_SIZE*2 + WRITE_SIZE, PMC pass committed under profiles/; Infinity-Cache hits included), "
                            "algorithmic operand+result bytes per launch: %d" % round(sum(
                                (2 * (M_ * K_ + N_ * K_) + M_ * N_ * {2: 8, 5: 10}.get(e_, 2)) for (_, e_, M_, N_, K_, _) in
                                [r for r in prof.records if r[0] == "gemm"]) / n_gemm),
            "mfma_busy_pmc": mfma_busy_pmc(cfg) if (B == 64 and args.dtype != "fp8") else None,
            "launches_per_step": n_gemm // nprof, "avg_launch_ms": round(gemm_ms / n_gemm, 5),
            "flop_per_launch": round(gemm_fl / n_gemm),
            "per_epilogue": {k: {"launches_per_step": v[0] // nprof, "avg_ms": round(v[1] / v[0], 5),
                                 "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 2)} for k, v in per.items()},
            "ms_per_step_by_kernel": {k: round(v / nprof, 4) for k, v in sorted(by_kind.items(), key=lambda kv: -kv[1])},
            "timing": f"hipEvent pairs around every launch, separate pass of {nprof} steps after the timed region",
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(cfg, sd, mano_cpu)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
