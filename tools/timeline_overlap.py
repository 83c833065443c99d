#!/usr/bin/env python3
"""What the two batches in flight buy, from a rocprofv3 --kernel-trace CSV of `bench.py`: for the timed steps (the last
`--tail` fraction of the trace) the union of kernel intervals (device busy), the time two or more kernels overlap, idle
gaps, and per kernel family the time it runs alone / beside another kernel.
--forwards a:b restricts the window to [end of the a-th forward, end of the b-th] (a forward ends with its MANO kernel):
the timed steps of `bench.py --steps 20 --warmup 5` are forwards 6..25.
Usage: python tools/timeline_overlap.py gpurun_out/prof_r02/r02_kernel_trace.csv [--forwards 6:24 | --tail 0.7]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
tail = float(sys.argv[sys.argv.index("--tail") + 1]) if "--tail" in sys.argv else 0.7


def family(n):
    for key, fam in (("gemm_px", "gemm_px"), ("gemm_x3", "gemm_x3"), ("gemm_fp8", "gemm_fp8"), ("gemm_tn", "gemm_tn"), ("gemm", "gemm_other"),
                     ("layernorm", "layernorm"), ("attention", "attention"), ("linear_f32", "linear_f32"), ("mano", "mano"),
                     ("im2col", "im2col"), ("broadcast", "broadcast")):
        if key in n:
            return fam
    return "other"


rows = []
with open(path, newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), family(r["Kernel_Name"]), r["Queue_Id"]))
rows.sort()
if "--forwards" in sys.argv:
    a, b = (int(v) for v in sys.argv[sys.argv.index("--forwards") + 1].split(":"))
    ends = sorted(e for s, e, fam, q in rows if fam == "mano")
    t_lo, t_hi = ends[a - 1], ends[b - 1]
    rows = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
    print(f"window: forwards {a + 1}..{b} = {(t_hi - t_lo) / 1e6 / (b - a):.3f} ms per forward")
else:
    t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1.0 - tail)
    rows = [r for r in rows if r[0] >= t_lo]
ev = []
for s, e, fam, q in rows:
    ev.append((s, 1, fam)); ev.append((e, -1, fam))
ev.sort()
active = defaultdict(int)
n_active = 0
last = ev[0][0]
busy = over = 0
alone = defaultdict(int); beside = defaultdict(int)
for t, d, fam in ev:
    dt = t - last
    if n_active >= 1:
        busy += dt
        if n_active >= 2:
            over += dt
        fams = [f for f, c in active.items() if c > 0]
        for f in fams:
            (beside if n_active >= 2 else alone)[f] += dt
    active[fam] += d; n_active += d; last = t
span = rows[-1][1] - rows[0][0]
tot = defaultdict(int)
for s, e, fam, q in rows:
    tot[fam] += e - s
print(f"span {span/1e6:.2f} ms, kernels {len(rows)}, device busy {busy/span*100:.1f} %, >=2 kernels at once {over/span*100:.1f} %, idle {100-busy/span*100:.1f} %")
print(f"sum of kernel durations {sum(tot.values())/1e6:.2f} ms = {sum(tot.values())/span:.3f} x span")
print(f"{'family':12s} {'sum ms':>9s} {'alone ms':>9s} {'beside ms':>9s}")
for fam in sorted(tot, key=lambda k: -tot[k]):
    print(f"{fam:12s} {tot[fam]/1e6:9.2f} {alone[fam]/1e6:9.2f} {beside[fam]/1e6:9.2f}")
