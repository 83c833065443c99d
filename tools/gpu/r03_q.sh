#!/bin/bash
set -o pipefail
O=gpurun_out/r03q; mkdir -p $O
for a in 0 16 32 64 48 80 112; do
CONV_DIRECT=$a timeout -k 10 200 python tools/prof_yolo.py 16 2 > $O/y_$a.txt 2>&1 || { tail $O/y_$a.txt; exit 1; }
echo "abl $a: $(grep -E '^  2 k3s1' $O/y_$a.txt) | $(grep -E '^  5 k3s1' $O/y_$a.txt)"
done
