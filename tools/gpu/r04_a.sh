#!/bin/bash
# round 4, first GPU call: the new folder pipeline (chain / api / rootnet tests), RCCL at world size 1, then the e2e bench lines
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_shard.py tests/test_gpu_chain.py tests/test_gpu_api.py tests/test_gpu_rootnet.py -x -q -m gpu -s > $O/t1.log 2>&1 || { tail -40 $O/t1.log; exit 1; }
tail -5 $O/t1.log
timeout -k 10 300 python3 bench.py --workload e2e --hands4 --chunks 4 --steps 3 --warmup 1 > $O/e2e64.log 2>&1 || { tail -30 $O/e2e64.log; exit 1; }
tail -c 1500 $O/e2e64.log; echo
timeout -k 10 300 python3 bench.py --workload e2e --hands4 --chunks 12 --steps 2 --warmup 1 --no-roofline > $O/e2e192.log 2>&1 || { tail -30 $O/e2e192.log; exit 1; }
tail -c 900 $O/e2e192.log; echo
