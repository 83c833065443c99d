#!/bin/bash
set -o pipefail
O=gpurun_out/r03af; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_api.py tests/test_gpu_rootnet.py -x -q -m gpu > $O/t.log 2>&1; rc=$?; tail -2 $O/t.log; [ $rc -eq 0 ] || exit $rc
HAMER_BENCH_SIDE=e2e timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.log 2>$O/bench.err || { tail $O/bench.err; exit 1; }
python -c "
import json; d=json.loads(open('$O/bench.log').read().strip().splitlines()[-1]); print(d['value']); e=d['side_configs']['configs[2] e2e 1080p, ~4 hands/frame']; print(e['value'], e['ms_per_step'], e['long_pass'])"
timeout -k 10 400 python bench.py --workload e2e-depth --hands4 --chunks 4 --steps 3 --warmup 1 > $O/e2e_depth.log 2>&1 || exit 1
python -c "
import json; d=json.loads(open('$O/e2e_depth.log').read().strip().splitlines()[-1]); print('e2e-depth:', d['value'],'hands/s', d['ms_per_step'],'ms per pass')"
