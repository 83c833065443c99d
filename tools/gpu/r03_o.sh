#!/bin/bash
set -o pipefail
O=gpurun_out/r03o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_yolo.py tests/test_gpu_chain.py -x -q -m gpu > $O/t.log 2>&1; rc=$?; tail -3 $O/t.log; [ $rc -le 1 ] || exit $rc
timeout -k 10 300 python bench.py --workload e2e --hands4 --chunks 4 --steps 4 --warmup 1 > $O/e2e.log 2>&1 || exit 1
python -c "
import json; d=json.loads(open('$O/e2e.log').read().strip().splitlines()[-1]); print('e2e:', d['value'],'hands/s', d['frames_per_s'],'frames/s', d['ms_per_step'],'ms per pass')"
