#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04t; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_yolo.py -x -q -m gpu > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -3 $O/t.log
for F in 48 16 1; do
  timeout -k 10 300 python3 tools/prof_yolo.py $F > $O/yolo$F.log 2>&1 || { tail -20 $O/yolo$F.log; exit 1; }
  echo "frames $F: $(tail -1 $O/yolo$F.log)"
done
