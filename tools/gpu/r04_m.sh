#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04m; mkdir -p $O
for F in 64 96; do
  timeout -k 10 300 python3 tools/prof_yolo.py $F > $O/yolo$F.log 2>&1 || { tail -20 $O/yolo$F.log; exit 1; }
  echo "frames $F: $(tail -2 $O/yolo$F.log | tr '\n' ' ')"
done
