#!/bin/bash
# round 3, GPU call H: lane-swap epilogue of the persistent GEMM: kernel tests, per-shape A/B, whole-model A/B
set -o pipefail
O=gpurun_out/r03h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $O/t_kernels.log 2>&1; rc=$?; tail -3 $O/t_kernels.log; [ $rc -le 1 ] || exit $rc
VARIANTS=201,202 SHAPES=qkv,fc1,kv ROUNDS=8 REPS=5 timeout -k 10 300 python tools/bench_gemm_ab.py > $O/ab_px_epilogue.log 2>&1 || { tail $O/ab_px_epilogue.log; exit 1; }
grep -v "^#\|amdgpu" $O/ab_px_epilogue.log
CONFIGS="serial_lds:1:px_lds_epi=1;serial_swap:1:;two_lds:2:px_lds_epi=1;two_swap:2:" ROUNDS=5 timeout -k 10 600 python tools/bench_model_ab.py > $O/model_ab_px_epilogue.log 2>&1 || { tail $O/model_ab_px_epilogue.log; exit 1; }
grep -v "^#\|amdgpu" $O/model_ab_px_epilogue.log
