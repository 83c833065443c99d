#!/bin/bash
# round 3: K groups inside the convolution workgroup -- tests, then the detector pass with and without them
set -o pipefail
O=gpurun_out/r03v; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_yolo.py -x -q -m gpu -k "k_groups or every_tile or split_k" > $O/t.log 2>&1; rc=$?; tail -5 $O/t.log; [ $rc -eq 0 ] || exit $rc
for f in 16 1; do
CONV_KGROUPS=1 timeout -k 10 200 python tools/prof_yolo.py $f > $O/yolo${f}_plain.txt 2>&1 || { tail $O/yolo${f}_plain.txt; exit 1; }
timeout -k 10 200 python tools/prof_yolo.py $f > $O/yolo${f}_k2.txt 2>&1 || { tail $O/yolo${f}_k2.txt; exit 1; }
grep -E "whole pass|conv stack" $O/yolo${f}_plain.txt $O/yolo${f}_k2.txt
done
