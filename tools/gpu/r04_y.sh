#!/bin/bash
# round 4: X-fragments-first read order adopted in every MFMA sub-step (px default, x3, x3r, tn / conv): full suite, bench twice, YOLO 48
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04y; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -3 $O/t.log
for i in 1 2; do
  timeout -k 10 300 python3 bench.py --no-side --no-cpu-baseline > $O/bench$i.log 2>&1 || { tail -20 $O/bench$i.log; exit 1; }
  python3 -c "
import json; r=json.loads(open('$O/bench$i.log').read().strip().splitlines()[-1]); print('bench', r['value'], r['ms_per_step'], r['model_mfma_frac'], r['roofline']['frac'], {k:(v['avg_ms'],v['tflops']) for k,v in r['roofline']['per_epilogue'].items()})"
done
timeout -k 10 300 python3 tools/prof_yolo.py 48 > $O/yolo48.log 2>&1 || exit 1
tail -1 $O/yolo48.log; sed -n 5p $O/yolo48.log
CONV_GENERAL=0 timeout -k 10 300 python3 tools/prof_yolo.py 16 > $O/yolo16.log 2>&1 || exit 1
tail -1 $O/yolo16.log
