#!/bin/bash
# round 4, closing run on the final tree: whole GPU suite, smoke, N = 2 rehearsal, default bench
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04final3; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -3 $O/t.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
HAMER_BENCH_REHEARSAL=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload e2e --hands4 --chunks 2 --steps 2 --warmup 1 --no-roofline > $O/e2e_2ranks.log 2>&1 || { tail -30 $O/e2e_2ranks.log; exit 1; }
tail -c 300 $O/e2e_2ranks.log; echo
timeout -k 10 600 python3 bench.py > $O/bench.json.log 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
r=json.loads(open('gpurun_out/r04final3/bench.json.log').read().strip().splitlines()[-1])
print('contract', r['value'], r['ms_per_step'], 'model', r['model_mfma_frac'], 'roofline', r['roofline']['frac'], r['roofline']['achieved'])
s=r['side_configs']
print('shard', s['configs[3] shard1024, N=1']['value'], 'fp8', s['configs[4] fp8 ViT-H, B=256']['value'], s['configs[4] fp8 ViT-H, B=256']['roofline']['frac'])
e=s['configs[2] e2e 1080p, ~4 hands/frame']
print('e2e64', e['value'], e['vs_contract_line'], 'long', e['long_pass']['value'], e['long_pass']['vs_contract_line'], 'conv', e['roofline']['conv']['achieved'], 'det cpu', e['cpu_baseline_detector']['value'], e['cpu_baseline_detector']['sample'][:60])
print('cpu', r['cpu_baseline']['value'], r['cpu_baseline']['cores'], 'side seconds', s['seconds'])
PY
