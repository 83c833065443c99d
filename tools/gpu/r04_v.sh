#!/bin/bash
# round 4: group_m sweep (tile walk: M-tiles per super-row) on the ViT-H shapes, default kernels, interleaved
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04v; mkdir -p $O
VARIANTS=108,102,104,112,116,124,148 ROUNDS=5 REPS=5 timeout -k 10 400 python3 tools/bench_gemm_ab.py > $O/group_m.log 2>&1 || { tail -20 $O/group_m.log; exit 1; }
tail -38 $O/group_m.log
