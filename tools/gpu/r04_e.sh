#!/bin/bash
# round 4: serial K ranges for the split-K layers of big detector passes -- tests, then the per-layer table at 48 / 16 frames
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04e; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_yolo.py -x -q -m gpu > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -3 $O/t.log
timeout -k 10 300 python3 tools/prof_yolo.py 48 > $O/yolo48.log 2>&1 || { tail -20 $O/yolo48.log; exit 1; }
tail -2 $O/yolo48.log
timeout -k 10 300 python3 tools/prof_yolo.py 16 > $O/yolo16.log 2>&1 || { tail -20 $O/yolo16.log; exit 1; }
tail -2 $O/yolo16.log
