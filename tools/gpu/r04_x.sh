#!/bin/bash
# round 4: lean copy issue adopted (px default, x3, x3r, helpers without M0 save/restore): full kernel + model tests, A/B against the
# old px form (variant 36 of the experiments library), whole-model bench twice
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04x; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -3 $O/t.log
ABLATION_LIB=1 VARIANTS=26,36 ROUNDS=8 REPS=5 timeout -k 10 300 python3 tools/bench_gemm_ab.py > $O/ab.log 2>&1 || { tail -20 $O/ab.log; exit 1; }
tail -11 $O/ab.log
for i in 1 2; do
  timeout -k 10 300 python3 bench.py --no-side --no-cpu-baseline > $O/bench$i.log 2>&1 || { tail -20 $O/bench$i.log; exit 1; }
  python3 -c "
import json; r=json.loads(open('$O/bench$i.log').read().strip().splitlines()[-1]); print('bench', r['value'], r['ms_per_step'], r['model_mfma_frac'], r['roofline']['frac'], r['roofline']['per_epilogue'])"
done
