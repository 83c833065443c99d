#!/bin/bash
# round 3: stride-2 32 -> 64 halo kernel -- parity with the implicit GEMM, then the per-layer table with and without it
set -o pipefail
O=gpurun_out/r03ag; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_yolo.py -x -q -m gpu -k "stride2_32 or 64_to_64" > $O/t.log 2>&1; rc=$?; tail -5 $O/t.log; [ $rc -eq 0 ] || exit $rc
CONV_DIRECT=2 timeout -k 10 200 python tools/prof_yolo.py 16 > $O/yolo16_off.txt 2>&1 || { tail $O/yolo16_off.txt; exit 1; }
timeout -k 10 200 python tools/prof_yolo.py 16 > $O/yolo16_on.txt 2>&1 || { tail $O/yolo16_on.txt; exit 1; }
grep -E "^  [0-3] k|whole pass|conv stack" $O/yolo16_off.txt; echo ---; grep -E "^  [0-3] k|whole pass|conv stack" $O/yolo16_on.txt
