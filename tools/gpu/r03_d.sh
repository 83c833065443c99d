#!/bin/bash
# round 3, GPU call D: persistent-GEMM grid at model level (interleaved A/B), per-shape A/B of the grid
set -o pipefail
O=gpurun_out/r03d; mkdir -p $O
CONFIGS="serial_256:1:;serial_240:1:px_grid=240;serial_248:1:px_grid=248;two_256:2:;two_240:2:px_grid=240;two_248:2:px_grid=248;two_232:2:px_grid=232" ROUNDS=5 STEPS=20 timeout -k 10 600 python tools/bench_model_ab.py > $O/model_ab_grid.log 2>&1 || { tail -5 $O/model_ab_grid.log; exit 1; }
cat $O/model_ab_grid.log
