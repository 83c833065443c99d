#!/bin/bash
# round 4: the software-pipelined persistent GEMM (variant 27) -- exactness, then interleaved A/B against 26 / 24
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "persistent_kernel_exact or tile_variants_exact or set_variant" > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -3 $O/t.log
VARIANTS=26,27 ROUNDS=6 REPS=5 EPI_STORE=1 timeout -k 10 300 python3 tools/bench_gemm_ab.py > $O/ab_store.log 2>&1 || { tail -20 $O/ab_store.log; exit 1; }
tail -12 $O/ab_store.log
VARIANTS=26,27 ROUNDS=6 REPS=5 timeout -k 10 300 python3 tools/bench_gemm_ab.py > $O/ab.log 2>&1 || { tail -20 $O/ab.log; exit 1; }
tail -12 $O/ab.log
