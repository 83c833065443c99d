#!/bin/bash
# round 3, GPU call B: full GPU suite, whole-model A/B of the residual fetch, default bench with side configurations, detector calibration
set -o pipefail
O=gpurun_out/r03b; mkdir -p $O
step() { echo "== $*" | tee -a $O/steps.log; }
step model_ab; ROUNDS=5 STEPS=20 timeout -k 10 600 python tools/bench_model_ab.py > $O/model_ab.log 2>&1 || { tail -5 $O/model_ab.log; exit 1; }
cat $O/model_ab.log
step bench; SECONDS=0; timeout -k 10 900 python bench.py > $O/bench.json.log 2>$O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json.log; echo "bench.py wall: $SECONDS s"
step tests; timeout -k 10 1100 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; rc=$?; tail -8 $O/t_all.log; [ $rc -le 1 ] || exit $rc
