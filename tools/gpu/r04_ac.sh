#!/bin/bash
# round 4: Conv 0 + Conv 1 of the detector as one launch (hm_conv2d_stem_pair): the detector tests, then the 48-frame layer table
# with and without it
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04ac; mkdir -p $O
timeout -k 10 300 python3 tools/probes/stem_pair_debug.py > $O/debug.log 2>&1 || { tail -40 $O/debug.log; exit 1; }
grep differ $O/debug.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_yolo.py -x -q -m gpu -k "stem" > $O/t_stem.log 2>&1 || { tail -40 $O/t_stem.log; exit 1; }
tail -2 $O/t_stem.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_yolo.py -x -q -m gpu > $O/t_yolo.log 2>&1 || { tail -40 $O/t_yolo.log; exit 1; }
tail -2 $O/t_yolo.log
for FS in 0 1; do
  FUSE_STEM=$FS timeout -k 10 300 python3 tools/prof_yolo.py 48 > $O/yolo48_fs$FS.log 2>&1 || { tail -20 $O/yolo48_fs$FS.log; exit 1; }
  echo "FUSE_STEM=$FS: $(sed -n 5,6p $O/yolo48_fs$FS.log | tr '\n' '|') $(tail -1 $O/yolo48_fs$FS.log)"
done
FUSE_STEM=1 timeout -k 10 300 python3 tools/prof_yolo.py 16 > $O/yolo16_fs1.log 2>&1 || exit 1
echo "16 frames: $(sed -n 5p $O/yolo16_fs1.log) $(tail -1 $O/yolo16_fs1.log)"
FUSE_STEM=1 timeout -k 10 300 python3 tools/prof_yolo.py 1 > $O/yolo1_fs1.log 2>&1 || exit 1
echo "1 frame: $(sed -n 5p $O/yolo1_fs1.log) $(tail -2 $O/yolo1_fs1.log | tr '\n' '|')"
