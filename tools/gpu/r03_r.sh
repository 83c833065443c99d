#!/bin/bash
# round 3: detector + chain suites with the 64 -> 64 halo kernel in place, then the 16-frame and 1-frame tables
set -o pipefail
O=gpurun_out/r03r; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_yolo.py tests/test_gpu_chain.py tests/test_gpu_api.py -x -q -m gpu > $O/t.log 2>&1; rc=$?; tail -4 $O/t.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/prof_yolo.py 16 > $O/yolo16.txt 2>&1 || { tail $O/yolo16.txt; exit 1; }
timeout -k 10 200 python tools/prof_yolo.py 1 > $O/yolo1.txt 2>&1 || exit 1
grep -E "whole pass|conv stack" $O/yolo16.txt $O/yolo1.txt
