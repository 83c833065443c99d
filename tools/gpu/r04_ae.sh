#!/bin/bash
# round 4: the stem pair kernel after a change: bit-equality probe + its tests, 48-frame table
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04ae; mkdir -p $O
timeout -k 10 300 python3 tools/probes/stem_pair_debug.py > $O/debug.log 2>&1 || { tail -40 $O/debug.log; exit 1; }
grep differ $O/debug.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_yolo.py -x -q -m gpu -k "stem" > $O/t_stem.log 2>&1 || { tail -40 $O/t_stem.log; exit 1; }
tail -1 $O/t_stem.log
FUSE_STEM=1 timeout -k 10 300 python3 tools/prof_yolo.py 48 > $O/yolo48.log 2>&1 || { tail -20 $O/yolo48.log; exit 1; }
echo "48 frames: $(sed -n 5,6p $O/yolo48.log | tr '\n' '|') $(tail -1 $O/yolo48.log)"
