#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04ao; mkdir -p $O
timeout -k 10 600 python3 bench.py > $O/bench.json.log 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
r=json.loads(open('gpurun_out/r04ao/bench.json.log').read().strip().splitlines()[-1])
print('contract', r['value'], r['ms_per_step'], 'model', r['model_mfma_frac'], 'roofline', r['roofline']['frac'], r['roofline']['achieved'])
s=r['side_configs']
print('shard', s['configs[3] shard1024, N=1']['value'], 'fp8', s['configs[4] fp8 ViT-H, B=256']['value'], s['configs[4] fp8 ViT-H, B=256']['roofline']['frac'])
e=s['configs[2] e2e 1080p, ~4 hands/frame']
print('e2e64', e['value'], e['ms_per_step'], e['vs_contract_line'], e['pipeline'], 'long', e['long_pass']['value'], e['long_pass']['vs_contract_line'], 'conv', e['roofline']['conv']['achieved'])
print('cpu', r['cpu_baseline']['value'], r['cpu_baseline']['cores'], 'det cpu', e['cpu_baseline_detector']['value'], 'side seconds', s['seconds'])
PY
