#!/bin/bash
set -o pipefail
O=gpurun_out/r03y; mkdir -p $O
SECONDS=0; HAMER_BENCH_SIDE=e2e timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.log 2>$O/bench.err || { tail $O/bench.err; exit 1; }
echo "wall $SECONDS s"
python -c "
import json; d=json.loads(open('$O/bench.log').read().strip().splitlines()[-1]); print(d['value']); print(d['side_configs'])"
