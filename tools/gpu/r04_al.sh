#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04al; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_yolo.py -x -q -m gpu -k "maxpool" > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -1 $O/t.log
timeout -k 10 600 python3 bench.py > $O/bench.json.log 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
r=json.loads(open('gpurun_out/r04al/bench.json.log').read().strip().splitlines()[-1])
print('contract', r['value'], r['ms_per_step'], 'model', r['model_mfma_frac'], 'roofline', r['roofline']['frac'], r['roofline']['achieved'])
s=r['side_configs']
print('shard', s['configs[3] shard1024, N=1']['value'], 'fp8', s['configs[4] fp8 ViT-H, B=256']['value'], s['configs[4] fp8 ViT-H, B=256']['roofline']['frac'])
e=s['configs[2] e2e 1080p, ~4 hands/frame']
print('e2e64', e['value'], e['ms_per_step'], e['vs_contract_line'], 'long', e['long_pass']['value'], e['long_pass']['vs_contract_line'], 'conv', e['roofline']['conv']['achieved'], e['roofline'].get('other_ms_per_pass'))
print('cpu', r['cpu_baseline']['value'], r['cpu_baseline']['cores'], 'side seconds', s['seconds'])
PY
timeout -k 10 300 python3 tools/prof_yolo.py 48 > $O/yolo48.log 2>&1 || exit 1
sed -n 3p $O/yolo48.log; tail -2 $O/yolo48.log
