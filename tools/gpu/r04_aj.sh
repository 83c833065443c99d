#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04aj; mkdir -p $O
timeout -k 10 600 python3 tools/probes/e2e_policy.py > $O/policy.log 2>&1 || { tail -30 $O/policy.log; exit 1; }
grep frames $O/policy.log
