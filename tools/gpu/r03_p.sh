#!/bin/bash
# round 3: 3x3 64 -> 64 halo kernel -- parity with the implicit GEMM, then the per-layer table with and without it
set -o pipefail
O=gpurun_out/r03p; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_yolo.py -x -q -m gpu -k "64_to_64 or direct_stem" > $O/t.log 2>&1; rc=$?; tail -5 $O/t.log; [ $rc -eq 0 ] || exit $rc
CONV_DIRECT=2 timeout -k 10 200 python tools/prof_yolo.py 16 > $O/yolo16_gemm.txt 2>&1 || { tail $O/yolo16_gemm.txt; exit 1; }
timeout -k 10 200 python tools/prof_yolo.py 16 > $O/yolo16_c64.txt 2>&1 || { tail $O/yolo16_c64.txt; exit 1; }
grep -E "  64   576 |whole pass|conv stack" $O/yolo16_gemm.txt; echo ---; grep -E "  64   576 |whole pass|conv stack" $O/yolo16_c64.txt
