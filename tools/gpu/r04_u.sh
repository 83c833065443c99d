#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04u; mkdir -p $O
timeout -k 10 300 python3 tools/bench_attention.py > $O/att.log 2>&1 || { tail -20 $O/att.log; exit 1; }
cat $O/att.log | tail -8
