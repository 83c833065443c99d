#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04ad; mkdir -p $O
timeout -k 10 300 python3 tools/probes/stem_pair_debug.py > $O/debug.log 2>&1 || { tail -40 $O/debug.log; exit 1; }
cat $O/debug.log
