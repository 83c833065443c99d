#!/bin/bash
set -o pipefail
O=gpurun_out/r03l; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_hamer.py -x -q -m gpu -k "head_major or qkv_head" > $O/t.log 2>&1; rc=$?; tail -5 $O/t.log; [ $rc -le 1 ] || exit $rc
CONFIGS="serial_tm:1:qkv_tm=1;serial_hm:1:;two_tm:2:qkv_tm=1;two_hm:2:" ROUNDS=5 timeout -k 10 600 python tools/bench_model_ab.py > $O/model_ab_qkv_layout.log 2>&1 || { tail $O/model_ab_qkv_layout.log; exit 1; }
grep -v "^#\|amdgpu" $O/model_ab_qkv_layout.log
timeout -k 10 300 python bench.py --no-side --no-cpu-baseline --in-flight 1 --steps 10 > $O/bench_serial.log 2>&1 || exit 1
python -c "
import json; d=json.loads(open('$O/bench_serial.log').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['ms_per_step_by_kernel'], d['roofline']['per_epilogue'])"
