"""Per-shape time of the YOLOv7 convolutions for a batch of 1080p frames (hipEvent pairs per launch)."""
import sys, collections, torch
sys.path.insert(0, ".")
from hamer_yolo_amd import synth, lib as L
from hamer_yolo_amd.yolo.engine import YoloEngine
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
eng = YoloEngine(synth.yolo_state_dict(seed=0, nc=3), nc=3, device="cuda")
frames = [synth.frame_u8(1080, 1920, seed=i).cuda() for i in range(F)]
for _ in range(3):
    eng.forward(frames)
torch.cuda.synchronize()
with L.profile(capacity=4096) as prof:
    for _ in range(3):
        eng.forward(frames)
    torch.cuda.synchronize()
by = collections.defaultdict(lambda: [0, 0.0])
tot = collections.defaultdict(float)
for kind, epi, M, N, K, ms in prof.records:
    tot[kind] += ms / 3
    if kind == "conv":
        by[(epi, M, N, K)][0] += 1
        by[(epi, M, N, K)][1] += ms / 3
print({k: round(v, 3) for k, v in tot.items()})
rows = sorted(by.items(), key=lambda kv: -kv[1][1])
for (epi, M, N, K), (n, ms) in rows[:24]:
    fl = 2.0 * M * N * K * (n / 3)
    byts = (M * K / (9 if epi // 10 == 3 else 1) + M * N) * 2 * (n / 3)
    print(f"k{epi//10}s{epi%10} M={M:7d} N={N:5d} K={K:5d} x{n//3:2d}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF  ~{byts/ms/1e9:6.2f} TB/s act")
