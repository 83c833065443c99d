#!/bin/bash
set -o pipefail
O=gpurun_out/r03n; mkdir -p $O
for S in e2e shard,e2e fp8,e2e; do HAMER_BENCH_SIDE=$S timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline > $O/b_$S.log 2>&1 || { tail $O/b_$S.log; exit 1; }; python -c "
import json; d=json.loads(open('$O/b_$S.log').read().strip().splitlines()[-1]); print('$S:', d['value'], {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d['side_configs'].items()})"; done
