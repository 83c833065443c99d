#!/bin/bash
set -o pipefail
O=gpurun_out/r03ac; mkdir -p $O
for f in 1 16; do
timeout -k 10 200 python tools/prof_yolo.py $f > $O/yolo${f}.txt 2>&1 || { tail $O/yolo${f}.txt; exit 1; }
grep -E "whole pass|conv stack" $O/yolo${f}.txt
done
timeout -k 10 600 python -m pytest tests/test_gpu_yolo.py tests/test_gpu_chain.py -x -q -m gpu > $O/t.log 2>&1; tail -2 $O/t.log
