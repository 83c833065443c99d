#!/bin/bash
set -o pipefail
O=gpurun_out/r03ai; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_yolo.py tests/test_gpu_chain.py tests/test_gpu_rootnet.py -x -q -m gpu > $O/t.log 2>&1; rc=$?; tail -3 $O/t.log; [ $rc -eq 0 ] || exit $rc
for f in 16 1; do
timeout -k 10 200 python tools/prof_yolo.py $f > $O/yolo${f}.txt 2>&1 || { tail $O/yolo${f}.txt; exit 1; }
grep -E "whole pass|conv stack" $O/yolo${f}.txt
done
