#!/bin/bash
# round 4: YOLOv7 per-layer table at the detector-pass sizes of the new folder driver (48 and 64 frames), and without split-K
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
timeout -k 10 300 python3 tools/prof_yolo.py 48 > $O/yolo48.log 2>&1 || { tail -20 $O/yolo48.log; exit 1; }
tail -3 $O/yolo48.log
CONV_SPLITK=1 timeout -k 10 300 python3 tools/prof_yolo.py 48 > $O/yolo48_nosplit.log 2>&1 || { tail -20 $O/yolo48_nosplit.log; exit 1; }
tail -2 $O/yolo48_nosplit.log
timeout -k 10 300 python3 tools/prof_yolo.py 64 > $O/yolo64.log 2>&1 || { tail -20 $O/yolo64.log; exit 1; }
tail -2 $O/yolo64.log
