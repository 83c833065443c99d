#!/bin/bash
# round 4: the opt-in tests of the experiments library (libhamer_hip_abl.so) on the final tree
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04am; mkdir -p $O
HM_TEST_EXPERIMENTS=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $O/t_exp.log 2>&1 || { tail -40 $O/t_exp.log; exit 1; }
tail -2 $O/t_exp.log
