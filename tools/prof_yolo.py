"""Per-layer table of the YOLOv7 convolutions (hipEvent pairs per launch): for F frames of 1080p in one batched pass, every
conv launch in network order with M, N, K, tile rows / columns, time and TFLOP/s, then the totals by kernel family.
Env: FUSE=0 (no E-ELAN pair fusion), SPLITK_WS=0 (no split-K scratch), CONV_TILE=1..6 (force a tile), CONV_SPLITK=1 (never split), CONV_DIRECT=1 / 2 (no direct kernels / the stem's only).
Usage: python tools/prof_yolo.py [frames=16] [reps=3]      (rocprofv3 --kernel-trace --stats -- python3 tools/prof_yolo.py 16)"""
import os, sys, collections, torch
sys.path.insert(0, ".")
from hamer_yolo_amd import synth, lib as L
if os.environ.get("ABLATION_LIB"):     # experiments library (python -m hamer_yolo_amd.build --ablations): CONV_DIRECT=4 / 5 = the two stem layers
    L.LIB_PATH = L.LIB_PATH.replace(".so", "_abl.so")   # without their activation / without their stores (WRONG results, bound diagnosis only)
from hamer_yolo_amd.yolo.engine import YoloEngine
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__))))
from runlog import banner
banner()
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = YoloEngine(synth.yolo_state_dict(seed=0, nc=3), nc=3, device="cuda")
eng.fuse_pairs = os.environ.get("FUSE", "1") == "1"
eng.split_k = os.environ.get("SPLITK_WS", "1") == "1"
eng.fuse_stem = os.environ.get("FUSE_STEM", "1") == "1"     # 0: Conv 0 and Conv 1 as two launches
L.check(L.load().hm_set_option(L.HM_OPT_CONV_TILE, int(os.environ.get("CONV_TILE", 0))))       # 1..6: force one tile for every layer
L.check(L.load().hm_set_option(L.HM_OPT_CONV_SPLITK, int(os.environ.get("CONV_SPLITK", 0))))   # 1: never split
L.check(L.load().hm_set_option(L.HM_OPT_CONV_KGROUPS, int(os.environ.get("CONV_KGROUPS", 0))))   # 1: no K groups inside a workgroup
L.check(L.load().hm_set_option(L.HM_OPT_CONV_GENERAL_LOADER, int(os.environ.get("CONV_GENERAL", 0))))   # 1: the general implicit-GEMM loader everywhere
L.check(L.load().hm_set_option(L.HM_OPT_CONV_DIRECT, int(os.environ.get("CONV_DIRECT", 0))))   # 1: implicit GEMM everywhere, 2: direct stem only
_chunk = torch.stack([synth.frame_u8(1080, 1920, seed=i) for i in range(F)]).cuda()      # as the folder drivers upload a chunk: slices of one
frames = [_chunk[i] for i in range(F)]                                                    # tensor -> ONE letterbox launch per pass
for _ in range(3):
    eng.forward(frames)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    eng.forward(frames)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 20 * 1e3
with L.profile(capacity=4096) as prof:
    for _ in range(REPS):
        eng.forward(frames)
    torch.cuda.synchronize()
recs = prof.records
per_pass = len(recs) // REPS
tot = collections.defaultdict(float)
for kind, epi, M, N, K, ms in recs:
    tot[kind] += ms / REPS
print(f"frames={F}: ms per pass by kernel family:", {k: round(v, 3) for k, v in tot.items()})
convs = [(i, r) for i, r in enumerate(recs[:per_pass]) if r[0] == "conv"]
print(f"{'#':>3} {'k/s':>4} {'M':>8} {'N':>5} {'K':>5} {'us':>8} {'TF/s':>7}   (average of {REPS} passes)")
cfl = cms = 0.0
for j, (i, (kind, epi, M, N, K, ms)) in enumerate(convs):
    t = sum(recs[i + p * per_pass][5] for p in range(REPS)) / REPS
    fl = 2.0 * M * N * K
    cfl += fl; cms += t
    print(f"{j:3d} k{epi//10}s{epi%10} {M:8d} {N:5d} {K:5d} {t*1e3:8.1f} {fl/t/1e9:7.1f}")
print(f"whole pass (letterbox + {len(convs)} convolutions + pools + decode), wall clock, no events: {wall:.3f} ms for {F} frame(s)")
print(f"conv stack: {cms:.3f} ms per pass, {cfl/1e9:.1f} GFLOP, {cfl/cms/1e9:.1f} TFLOP/s ({cfl/cms/1e9/2500*100:.1f} % of the 2.5 PF 16-bit MFMA peak)")
