#!/bin/bash
# round 4: what bounds the two stem layers of YOLOv7 at 48 frames (experiments library: no activation / no stores), and a finer
# host timeline of the first milliseconds of a 64-frame pass
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04ab; mkdir -p $O
for CD in 0 4 5; do
  ABLATION_LIB=1 CONV_DIRECT=$CD timeout -k 10 300 python3 tools/prof_yolo.py 48 > $O/yolo48_cd$CD.log 2>&1 || { tail -20 $O/yolo48_cd$CD.log; exit 1; }
  echo "CONV_DIRECT=$CD: $(grep -E '^ +0 k3s1' $O/yolo48_cd$CD.log) | $(grep -E '^ +1 k3s2' $O/yolo48_cd$CD.log)"
done
timeout -k 10 300 python3 tools/probes/e2e_trace.py 64 > $O/trace64.log 2>&1 || { tail -30 $O/trace64.log; exit 1; }
cat $O/trace64.log | tail -40
