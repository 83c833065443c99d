#!/bin/bash
# round 4: lean vs general implicit-GEMM loader, same box, same process state: per-layer tables at 48 / 16 / 1 frames
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04l; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_yolo.py -x -q -m gpu -k "lean_loader or serial_k or batched_frames or k_groups" > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -3 $O/t.log
for F in 48 16 1; do
  for G in 0 1 0 1; do
    CONV_GENERAL=$G timeout -k 10 300 python3 tools/prof_yolo.py $F > $O/yolo${F}_general$G.log 2>&1 || { tail -20 $O/yolo${F}_general$G.log; exit 1; }
    echo "frames $F general_loader=$G: $(tail -1 $O/yolo${F}_general$G.log)"
  done
done
