#!/bin/bash
# round 4: raw BMP rows + device-side flip -- the driver tests that read files, the host timeline, the e2e side entry
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04z; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_chain.py tests/test_gpu_api.py tests/test_gpu_rootnet.py -x -q -m gpu > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -3 $O/t.log
timeout -k 10 300 python3 tools/probes/e2e_trace.py 64 > $O/trace64.log 2>&1 || { tail -30 $O/trace64.log; exit 1; }
tail -15 $O/trace64.log
HAMER_BENCH_SIDE=e2e timeout -k 10 600 python3 bench.py --no-cpu-baseline > $O/bench.json.log 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
r=json.loads(open('gpurun_out/r04z/bench.json.log').read().strip().splitlines()[-1])
print('contract', r['value'], r['ms_per_step'], r['model_mfma_frac'], r['roofline']['frac'])
e=r['side_configs']['configs[2] e2e 1080p, ~4 hands/frame']
print('e2e64', e['value'], e['ms_per_step'], e['vs_contract_line'], 'long', e['long_pass'])
PY
