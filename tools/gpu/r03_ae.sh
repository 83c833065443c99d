#!/bin/bash
set -o pipefail
O=gpurun_out/r03ae; mkdir -p $O
timeout -k 10 400 python bench.py --workload e2e-depth --hands4 --chunks 4 --steps 3 --warmup 1 > $O/e2e_depth.log 2>&1 || { tail $O/e2e_depth.log; exit 1; }
python -c "
import json; d=json.loads(open('$O/e2e_depth.log').read().strip().splitlines()[-1]); print('e2e-depth:', d['value'],'hands/s', d['ms_per_step'],'ms per pass')"
timeout -k 10 400 python bench.py --workload e2e --hands4 --chunks 4 --steps 3 --warmup 1 > $O/e2e.log 2>&1 || exit 1
python -c "
import json; d=json.loads(open('$O/e2e.log').read().strip().splitlines()[-1]); print('e2e:', d['value'],'hands/s', d['ms_per_step'],'ms per pass')"
timeout -k 10 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_rootnet.py tests/test_gpu_api.py -x -q -m gpu > $O/t.log 2>&1; tail -2 $O/t.log
