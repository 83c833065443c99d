#!/bin/bash
# round 3, GPU call C: CU-partition feasibility probe (the CU-mask bit map is profiles/r03_cu_mask_map.txt)
set -o pipefail
O=gpurun_out/r03c; mkdir -p $O
for m in 16 32 8; do
MCU=$m timeout -k 10 300 python tools/probes/cu_partition.py > $O/cu_partition_$m.log 2>&1 || { tail -20 $O/cu_partition_$m.log; exit 1; }
grep -v "^#\|amdgpu.ids" $O/cu_partition_$m.log
done
