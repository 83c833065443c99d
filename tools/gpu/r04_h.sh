#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04h; mkdir -p $O
timeout -k 10 300 python3 tools/probes/e2e_trace.py 64 > $O/trace64.log 2>&1 || { tail -30 $O/trace64.log; exit 1; }
cat $O/trace64.log | tail -30
