#!/bin/bash
# round 3, GPU call A: new-kernel tests, residual A/B, contract bench, YOLO per-layer table, detector calibration sweep
set -o pipefail
O=gpurun_out/r03a; mkdir -p $O
step() { echo "== $*" | tee -a $O/steps.log; }
step tests; timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_shard.py -x -q -m gpu > $O/t_kernels.log 2>&1; rc=$?; tail -3 $O/t_kernels.log; [ $rc -le 1 ] || exit $rc
step ab_resid; VARIANTS=200,201 SHAPES=proj,fc2 ROUNDS=8 REPS=5 timeout -k 10 300 python tools/bench_gemm_ab.py > $O/ab_resid_warm.log 2>&1 || exit 1
COLD_RESID=1 VARIANTS=200,201 SHAPES=proj,fc2 ROUNDS=8 REPS=5 timeout -k 10 300 python tools/bench_gemm_ab.py > $O/ab_resid_cold.log 2>&1 || exit 1
cat $O/ab_resid_warm.log $O/ab_resid_cold.log
step bench; timeout -k 10 600 python bench.py > $O/bench.json.log 2>$O/bench.err || exit 1
cat $O/bench.json.log
step hamer_tests; timeout -k 10 900 python -m pytest tests/test_gpu_hamer.py -x -q -m gpu > $O/t_hamer.log 2>&1; rc=$?; tail -3 $O/t_hamer.log; [ $rc -le 1 ] || exit $rc
step yolo16; timeout -k 10 300 python tools/prof_yolo.py 16 > $O/yolo_layers_16.log 2>&1 || exit 1
step yolo1; timeout -k 10 300 python tools/prof_yolo.py 1 > $O/yolo_layers_1.log 2>&1 || exit 1
tail -3 $O/yolo_layers_16.log; tail -2 $O/yolo_layers_1.log
step hands; timeout -k 10 300 python tools/probes/yolo_hands_per_frame.py > $O/hands_per_frame.log 2>&1 || exit 1
cat $O/hands_per_frame.log
