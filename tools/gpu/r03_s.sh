#!/bin/bash
# round 3: 128x128 tile (variant 0) vs the 256x256 kernels (variant 26) per ViT-H GEMM shape and batch size
set -o pipefail
O=gpurun_out/r03s; mkdir -p $O
for b in 16 24 32 40 48 56 64 68 72 80 96; do
BATCH=$b VARIANTS=0,26 ROUNDS=4 REPS=5 SHAPES=qkv,proj,fc1,fc2 timeout -k 10 120 python tools/bench_gemm_ab.py > $O/b$b.log 2>&1 || { tail $O/b$b.log; exit 1; }
echo "== B=$b"; grep -v "^#\|amdgpu.ids" $O/b$b.log
done
