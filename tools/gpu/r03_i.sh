#!/bin/bash
# round 3, GPU call I: direct stem convolution: tests, per-layer tables
set -o pipefail
O=gpurun_out/r03i; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_yolo.py -x -q -m gpu > $O/t_yolo.log 2>&1; rc=$?; tail -5 $O/t_yolo.log; [ $rc -le 1 ] || exit $rc
for F in 16 1; do timeout -k 10 200 python tools/prof_yolo.py $F 3 > $O/yolo_layers_$F.log 2>&1 || exit 1; sed -n 5,9p $O/yolo_layers_$F.log; tail -2 $O/yolo_layers_$F.log; done
