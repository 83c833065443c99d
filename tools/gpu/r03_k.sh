#!/bin/bash
set -o pipefail
O=gpurun_out/r03k; mkdir -p $O
for C in 4 8 16; do timeout -k 10 300 python bench.py --workload e2e --hands4 --chunks $C --steps 3 --warmup 1 > $O/e2e_c$C.log 2>&1 || { tail $O/e2e_c$C.log; exit 1; }; python -c "
import json,sys; d=json.loads(open('$O/e2e_c$C.log').read().strip().splitlines()[-1]); print('chunks $C:', d['value'],'hands/s', d['frames_per_s'],'frames/s', d['ms_per_step'],'ms per pass', d['frames_per_pass'],'frames')"; done
timeout -k 10 300 python bench.py --workload shard1024 --steps 5 --warmup 2 --no-roofline --no-cpu-baseline > $O/shard1024.log 2>&1 || { tail $O/shard1024.log; exit 1; }
python -c "
import json; d=json.loads(open('$O/shard1024.log').read().strip().splitlines()[-1]); print('shard1024:', d['value'], d['ms_per_step'])"
timeout -k 10 300 python bench.py --no-side --no-roofline --no-cpu-baseline > $O/crops.log 2>&1 || exit 1
python -c "
import json; d=json.loads(open('$O/crops.log').read().strip().splitlines()[-1]); print('crops:', d['value'], d['ms_per_step'])"
