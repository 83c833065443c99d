#!/bin/bash
set -o pipefail
O=gpurun_out/r03j; mkdir -p $O
timeout -k 10 300 python tools/probes/shard_job_gap.py > $O/shard_job_gap.log 2>&1 || { tail $O/shard_job_gap.log; exit 1; }
grep -v "^#\|amdgpu" $O/shard_job_gap.log
