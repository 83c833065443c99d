#!/bin/bash
# round 3: 1x1 weights-in-registers kernel -- parity with the implicit GEMM, then the per-layer table with and without it
set -o pipefail
O=gpurun_out/r03ah; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_yolo.py -x -q -m gpu -k "1x1_weights or stride2_32 or 64_to_64 or direct_stem" > $O/t.log 2>&1; rc=$?; tail -5 $O/t.log; [ $rc -eq 0 ] || exit $rc
CONV_DIRECT=2 timeout -k 10 200 python tools/prof_yolo.py 16 > $O/yolo16_off.txt 2>&1 || { tail $O/yolo16_off.txt; exit 1; }
timeout -k 10 200 python tools/prof_yolo.py 16 > $O/yolo16_on.txt 2>&1 || { tail $O/yolo16_on.txt; exit 1; }
CONV_DIRECT=3 timeout -k 10 200 python tools/prof_yolo.py 16 > $O/yolo16_all.txt 2>&1 || { tail $O/yolo16_all.txt; exit 1; }
for f in off on all; do echo "== $f"; grep -E "k1s1 +(245760|61440) +(128|256) +(128|256) |whole pass|conv stack" $O/yolo16_$f.txt; done
