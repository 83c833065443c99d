#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04ap; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_chain.py -x -q -m gpu -k "ring" > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -2 $O/t.log
