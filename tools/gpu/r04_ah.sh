#!/bin/bash
# round 4: whole GPU suite after the detector changes (packed SiLU, stem pair, persistent 1x1), then same-box layer tables
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04ah; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -3 $O/t.log
timeout -k 10 300 python3 tools/prof_yolo.py 48 > $O/yolo48.log 2>&1 || exit 1
tail -1 $O/yolo48.log
FUSE_STEM=0 timeout -k 10 300 python3 tools/prof_yolo.py 48 > $O/yolo48_two_stem_launches.log 2>&1 || exit 1
tail -1 $O/yolo48_two_stem_launches.log
timeout -k 10 300 python3 tools/prof_yolo.py 64 > $O/yolo64.log 2>&1 || exit 1
tail -1 $O/yolo64.log
timeout -k 10 300 python3 tools/prof_yolo.py 16 > $O/yolo16.log 2>&1 || exit 1
tail -1 $O/yolo16.log
timeout -k 10 300 python3 tools/prof_yolo.py 1 > $O/yolo1.log 2>&1 || exit 1
tail -2 $O/yolo1.log
