#!/bin/bash
# round 4, final tree: the round's evidence collection once more (contract command and its serial twin under rocprofv3, PMC over
# the step's GEMM shapes, YOLO 16 / 1) -- the GEMM loop and the detector changed after tools/gpu/r04_collect.sh had run
set -o pipefail
export TMPDIR=/tmp
bash tools/collect_profiles.sh r04f > gpurun_out/collect_r04f.log 2>&1 || { tail -30 gpurun_out/collect_r04f.log; exit 1; }
tail -8 gpurun_out/collect_r04f.log
