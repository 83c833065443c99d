#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04r; mkdir -p $O
timeout -k 10 300 python3 tools/probes/gemm_clock.py > $O/gemm_clock.log 2>&1 || { tail -20 $O/gemm_clock.log; exit 1; }
cat $O/gemm_clock.log
