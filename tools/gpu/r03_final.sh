#!/bin/bash
# round 3: final verification -- smoke, full GPU suite, default bench, profile collection
set -o pipefail
O=gpurun_out/r03final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 1100 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; rc=$?; tail -4 $O/t_all.log; [ $rc -le 1 ] || exit $rc
SECONDS=0; timeout -k 10 900 python bench.py > $O/bench.json.log 2>$O/bench.err || { tail -20 $O/bench.err; exit 1; }
echo "bench.py wall: $SECONDS s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03final/bench.json.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['model_mfma_frac'], d['roofline']['frac'], d['roofline']['per_epilogue'])
print({k:(v.get('value') if isinstance(v,dict) else v) for k,v in d['side_configs'].items()}, d['cpu_baseline']['value'])
PY
bash tools/collect_profiles.sh r03 > $O/collect.log 2>&1 || { tail -20 $O/collect.log; exit 1; }
tail -4 $O/collect.log
