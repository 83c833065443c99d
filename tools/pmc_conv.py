#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes over tools/prof_yolo.py for the convolution kernels: per kernel name, launches, and the mean
per launch of every counter found.  Derived: LDS bank-conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; MFMA-pipe busy =
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024); traffic beyond L2 = 2 x FETCH_SIZE KiB (gfx950 correction,
MI355X_MICROARCH.md) + WRITE_SIZE KiB.  Usage: pmc_conv.py out.json pass1.csv [pass2.csv ...]"""
import collections, csv, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        key = None
        if "conv3x3_c64_kernel" in name: key = "conv3x3_c64_kernel"
        elif "conv3x3_direct_kernel" in name: key = "conv3x3_direct_kernel"
        elif "gemm_tn_kernel" in name: key = "gemm_tn_kernel (all convolution tiles)"
        if key: acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    o = {"launches": max(len(v) for v in c.values())}
    for n, v in c.items(): o[n + "_per_launch"] = sum(v) / len(v)
    if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c: o["lds_bank_conflict_share"] = round(sum(c["SQ_LDS_BANK_CONFLICT"]) / max(1.0, sum(c["SQ_LDS_IDX_ACTIVE"])), 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c: o["mfma_busy_frac"] = round(sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(c["GRBM_GUI_ACTIVE"]) / 8 * 1024), 4)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c: o["hbm_bytes_per_launch"] = (2 * sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) + sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])) * 1024
    out[k] = o
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
