#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04o; mkdir -p $O
timeout -k 10 300 python3 tools/probes/decode_rate.py > $O/decode_rate.log 2>&1 || { tail -20 $O/decode_rate.log; exit 1; }
cat $O/decode_rate.log
