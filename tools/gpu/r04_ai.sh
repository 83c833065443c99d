#!/bin/bash
# round 4: bench line + side configurations, host timeline, rocprofv3 kernel stats of a 48-frame detector pass (new kernels)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04ai; mkdir -p $O
timeout -k 10 600 python3 bench.py > $O/bench.json.log 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
r=json.loads(open('gpurun_out/r04ai/bench.json.log').read().strip().splitlines()[-1])
print('contract', r['value'], r['ms_per_step'], 'model', r['model_mfma_frac'], 'roofline', r['roofline']['frac'], r['roofline']['achieved'])
s=r['side_configs']
print('shard', s['configs[3] shard1024, N=1']['value'], 'fp8', s['configs[4] fp8 ViT-H, B=256']['value'], s['configs[4] fp8 ViT-H, B=256']['roofline']['frac'])
e=s['configs[2] e2e 1080p, ~4 hands/frame']
print('e2e64', e['value'], e['vs_contract_line'], 'long', e['long_pass']['value'], e['long_pass']['vs_contract_line'], 'conv', e['roofline']['conv'], 'det cpu', e['cpu_baseline_detector']['value'])
print('cpu', r['cpu_baseline']['value'], r['cpu_baseline']['cores'], 'side seconds', s['seconds'])
PY
timeout -k 10 300 python3 tools/probes/e2e_trace.py 64 > $O/trace64.log 2>&1 || { tail -30 $O/trace64.log; exit 1; }
grep -v "decoded file" $O/trace64.log | tail -16
mkdir -p $O/prof_yolo48
rocprofv3 --kernel-trace --stats -d $O/prof_yolo48 -o r04_yolo48b --output-format csv -- python3 tools/prof_yolo.py 48 3 > $O/prof_yolo48/prof_yolo.log 2>&1 || exit 1
tail -1 $O/prof_yolo48/prof_yolo.log
find $O/prof_yolo48 -name '*kernel_stats.csv' | head -2
