#!/bin/bash
# round 3, GPU call F: ToMe fp32 metric tests, detector agreement, yolo tests
set -o pipefail
O=gpurun_out/r03f; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_hamer.py tests/test_gpu_chain.py tests/test_gpu_yolo.py tests/test_gpu_loaders.py -q -m gpu > $O/t.log 2>&1; rc=$?; tail -15 $O/t.log; [ $rc -le 1 ] || exit $rc
tail -12 gpurun_out/parity_report.jsonl
