#!/bin/bash
set -o pipefail
O=gpurun_out/r03aj; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_yolo.py -x -q -m gpu -k "stride2_32" > $O/t.log 2>&1; rc=$?; tail -5 $O/t.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/prof_yolo.py 16 > $O/yolo16.txt 2>&1 || { tail $O/yolo16.txt; exit 1; }
grep -E "^  [0-3] k|whole pass|conv stack" $O/yolo16.txt
