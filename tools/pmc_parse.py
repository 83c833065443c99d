#!/usr/bin/env python3
"""Parse the rocprofv3 --pmc CSVs of tools/pmc_shapes.py into per-launch traffic beyond L2 (bytes) and MFMA-pipe busy
fractions.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
stream, so it is doubled (MI355X_MICROARCH.md, HBM section).  SQ_VALU_MFMA_BUSY_CYCLES is summed over all 1024 SIMDs,
GRBM_GUI_ACTIVE over the 8 XCDs: shader cycles of a launch = GRBM_GUI_ACTIVE / 8, busy = MFMA_BUSY / (cycles * 1024).
Usage: pmc_parse.py fetch.csv write.csv mfma.csv traffic.json mfma_busy.json"""
import csv
import json
import sys

sys.path.insert(0, "tools")
from pmc_shapes import KERNELS, M, SHAPES  # noqa: E402


def per_kernel(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    ours = [r for r in rows if any(k in r["Kernel_Name"] for k in KERNELS)]
    return [(float(r["Counter_Value"]), r["Kernel_Name"]) for r in ours]


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
busy = per_kernel(sys.argv[3], "SQ_VALU_MFMA_BUSY_CYCLES")
gui = per_kernel(sys.argv[3], "GRBM_GUI_ACTIVE")
names = [s[0] for s in SHAPES] + ["layernorm", "attention"]
n = len(names)
assert len(fetch) == 2 * n and len(write) == 2 * n and len(busy) == 2 * n and len(gui) == 2 * n, (len(fetch), len(write), len(busy), len(gui))
traffic, mfma = {}, {"_note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over tools/pmc_shapes.py (B=64 shapes, fp16, second "
                                   "repetition); shader cycles = GRBM_GUI_ACTIVE / 8, mfma_busy_frac = MFMA_BUSY / (shader cycles * 1024)"}
for i, nm in enumerate(names):
    f = fetch[n + i][0] * 1024 * 2      # second repetition; gfx950 correction x2
    w = write[n + i][0] * 1024
    alg = None
    if i < len(SHAPES):
        _, K, N, epi = SHAPES[i]
        osz = 4 if epi == 2 else 2
        alg = M * K * 2 + N * K * 2 + M * N * osz + (M * N * 4 if epi == 2 else 0)
    traffic[nm] = {"kernel": fetch[n + i][1][:90], "fetch_bytes": f, "write_bytes": w, "hbm_bytes": f + w, "algorithmic_bytes": alg}
    cyc = gui[n + i][0] / 8.0
    mfma[nm] = {"kernel": busy[n + i][1][:90], "SQ_VALU_MFMA_BUSY_CYCLES": busy[n + i][0], "GRBM_GUI_ACTIVE": gui[n + i][0],
                "shader_cycles": cyc, "mfma_busy_frac": round(busy[n + i][0] / (cyc * 1024), 4)}
json.dump(traffic, open(sys.argv[4], "w"), indent=1)
json.dump(mfma, open(sys.argv[5], "w"), indent=1)
print(json.dumps({k: {"hbm_bytes": v["hbm_bytes"], "algorithmic_bytes": v["algorithmic_bytes"]} for k, v in traffic.items()}, indent=1))
print(json.dumps({k: v["mfma_busy_frac"] for k, v in mfma.items() if k != "_note"}))
