#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04ak; mkdir -p $O
timeout -k 10 600 python3 tools/probes/e2e_policy.py > $O/policy.log 2>&1 || { tail -30 $O/policy.log; exit 1; }
grep frames $O/policy.log
HAMER_BENCH_SIDE=e2e timeout -k 10 600 python3 bench.py --no-cpu-baseline > $O/bench.json.log 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
r=json.loads(open('gpurun_out/r04ak/bench.json.log').read().strip().splitlines()[-1])
print('contract', r['value'], r['ms_per_step'])
e=r['side_configs']['configs[2] e2e 1080p, ~4 hands/frame']
print('e2e64', e['value'], e['ms_per_step'], e['vs_contract_line'], e['pipeline'], 'long', e['long_pass'])
PY
