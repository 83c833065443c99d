#!/bin/bash
# round 3, GPU call G: full GPU suite, default bench, yolo per-layer after the invariant split rule
set -o pipefail
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; rc=$?; tail -6 $O/t_all.log; [ $rc -le 1 ] || exit $rc
for F in 16 1; do timeout -k 10 200 python tools/prof_yolo.py $F 3 > $O/yolo_layers_$F.log 2>&1 || exit 1; tail -2 $O/yolo_layers_$F.log; done
SECONDS=0; timeout -k 10 900 python bench.py > $O/bench.json.log 2>$O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03g/bench.json.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['model_mfma_frac'], d['roofline']['frac'], d['roofline']['per_epilogue'])
print(json.dumps(d['side_configs'], indent=1))
PY
echo "bench.py wall: $SECONDS s"
