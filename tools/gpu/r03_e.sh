#!/bin/bash
# round 3, GPU call E2: conv tests, deep-ring tiles forced + automatic choice with / without split-K, 16 frames and 1 frame
set -o pipefail
O=gpurun_out/r03e2; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_yolo.py tests/test_gpu_rootnet.py -x -q -m gpu > $O/t_yolo.log 2>&1; rc=$?; tail -5 $O/t_yolo.log; [ $rc -le 1 ] || exit $rc
for F in 16 1; do
  for T in 7 8 9 0; do
    CONV_TILE=$T CONV_SPLITK=1 timeout -k 10 200 python tools/prof_yolo.py $F 3 > $O/layers_f${F}_t${T}.log 2>&1 || { tail -5 $O/layers_f${F}_t${T}.log; exit 1; }
    echo "F=$F tile=$T nosplit: $(tail -1 $O/layers_f${F}_t${T}.log)"
  done
  timeout -k 10 200 python tools/prof_yolo.py $F 3 > $O/layers_f${F}_auto_split.log 2>&1 || exit 1
  echo "F=$F auto split: $(tail -1 $O/layers_f${F}_auto_split.log)"
  FUSE=0 timeout -k 10 200 python tools/prof_yolo.py $F 3 > $O/layers_f${F}_auto_split_nofuse.log 2>&1 || exit 1
  echo "F=$F auto split nofuse: $(tail -1 $O/layers_f${F}_auto_split_nofuse.log)"
done
