#!/bin/bash
# round 4: gemm_px_kernel with lean copy issue (variant 36, experiments library) against the production kernel
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04w; mkdir -p $O
HM_TEST_EXPERIMENTS=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pipelined_persistent_experiment_exact and (39 or 40)" > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -3 $O/t.log
ABLATION_LIB=1 VARIANTS=26,39,40 ROUNDS=8 REPS=5 EPI_STORE=1 timeout -k 10 300 python3 tools/bench_gemm_ab.py > $O/ab_store.log 2>&1 || { tail -20 $O/ab_store.log; exit 1; }
tail -17 $O/ab_store.log
