#!/bin/bash
set -o pipefail
O=gpurun_out/r03t; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_hamer.py -x -q -m gpu > $O/t.log 2>&1; rc=$?; tail -4 $O/t.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 700 python tools/probes/forward_vs_batch.py > $O/forward_vs_batch.log 2>&1 || { tail $O/forward_vs_batch.log; exit 1; }
grep "^B=" $O/forward_vs_batch.log
