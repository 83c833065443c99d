#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter CSVs by kernel: mean counter value per dispatch and the number of dispatches, for the
kernels whose name contains one of the given substrings.  FETCH_SIZE / WRITE_SIZE (KiB) are converted to bytes, FETCH_SIZE
doubled as MI355X_MICROARCH.md prescribes for gfx950; with SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE in one file the MFMA
busy fraction = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024) is added.
Usage: pmc_by_kernel.py out.json name1,name2,... file.csv [file.csv ...]"""
import collections
import csv
import json
import sys

out, names, files = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in files:
    for r in csv.DictReader(open(path)):
        k = next((n for n in names if n in r["Kernel_Name"]), None)
        if k is not None:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"_note": "mean per dispatch over the whole run; FETCH_SIZE x 1024 x 2 (gfx950), WRITE_SIZE x 1024 -> bytes; "
                "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)"}
for k, cs in agg.items():
    e = {"dispatches": max(len(v) for v in cs.values())}
    if "FETCH_SIZE" in cs:
        e["fetch_bytes"] = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024 * 2
    if "WRITE_SIZE" in cs:
        e["write_bytes"] = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024
    if "fetch_bytes" in e and "write_bytes" in e:
        e["hbm_bytes"] = e["fetch_bytes"] + e["write_bytes"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
        busy, gui = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(cs["GRBM_GUI_ACTIVE"])
        e["mfma_busy_frac"] = busy / (gui / 8.0 * 1024.0)
        e["shader_cycles_per_dispatch"] = gui / 8.0 / len(cs["GRBM_GUI_ACTIVE"])
    res[k] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
