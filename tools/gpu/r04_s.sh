#!/bin/bash
# round 4: per-layer tile sweep at 48 frames with the lean loader (every tile forced in turn; layers a tile cannot serve keep their own)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04s; mkdir -p $O
for T in 0 1 2 4 5 6; do
  CONV_TILE=$T timeout -k 10 300 python3 tools/prof_yolo.py 48 > $O/yolo48_tile$T.log 2>&1 || { tail -20 $O/yolo48_tile$T.log; exit 1; }
  echo "tile $T: $(tail -1 $O/yolo48_tile$T.log)"
done
