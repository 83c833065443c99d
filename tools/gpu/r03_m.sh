#!/bin/bash
set -o pipefail
O=gpurun_out/r03m; mkdir -p $O
for C in 4 6 8 12; do timeout -k 10 300 python bench.py --workload e2e --hands4 --chunks $C --steps 3 --warmup 1 > $O/e2e_c$C.log 2>&1 || { tail $O/e2e_c$C.log; exit 1; }; python -c "
import json,sys; d=json.loads(open('$O/e2e_c$C.log').read().strip().splitlines()[-1]); print('chunks $C:', d['value'],'hands/s', d['frames_per_s'],'frames/s', d['ms_per_step'],'ms per pass', round(d['ms_per_step']/$C,1),'ms per chunk')"; done
timeout -k 10 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_rootnet.py tests/test_gpu_api.py -x -q -m gpu > $O/t.log 2>&1; rc=$?; tail -3 $O/t.log; [ $rc -le 1 ] || exit $rc
