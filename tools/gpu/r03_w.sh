#!/bin/bash
set -o pipefail
O=gpurun_out/r03w; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_yolo.py tests/test_gpu_chain.py tests/test_gpu_api.py tests/test_gpu_loaders.py -x -q -m gpu > $O/t.log 2>&1; rc=$?; tail -6 $O/t.log; [ $rc -eq 0 ] || exit $rc
