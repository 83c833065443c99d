#!/bin/bash
# round 4, closing run: the whole GPU suite, then the YOLO kernel stats with the final kernels (48 / 16 / 1 frames)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04final; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -4 $O/t.log
for F in 48 16 1; do
  mkdir -p $O/yolo$F
  rocprofv3 --kernel-trace --stats -d $O/yolo$F -o r04_yolo$F --output-format csv -- python3 tools/prof_yolo.py $F 3 > $O/yolo$F/prof_yolo.log 2>&1 || exit 1
  grep "conv stack" $O/yolo$F/prof_yolo.log
  rm -f $O/yolo$F/*kernel_trace.csv
done
