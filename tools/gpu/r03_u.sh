#!/bin/bash
set -o pipefail
O=gpurun_out/r03u; mkdir -p $O
BATCHES=24,32,40,64,72,76,80 timeout -k 10 400 python tools/probes/forward_vs_batch.py > $O/forward_vs_batch.log 2>&1 || { tail $O/forward_vs_batch.log; exit 1; }
grep "^B=" $O/forward_vs_batch.log
timeout -k 10 300 python bench.py --workload e2e --hands4 --chunks 4 --steps 4 --warmup 1 > $O/e2e.log 2>&1 || exit 1
python -c "
import json; d=json.loads(open('$O/e2e.log').read().strip().splitlines()[-1]); print('e2e:', d['value'],'hands/s', d['frames_per_s'],'frames/s', d['ms_per_step'],'ms per pass')"
