#!/bin/bash
# round 4: rehearsal of the N = 2 code paths on the one-GPU box (both ranks on the one card, gloo): e2e folder sharding, then the contract workload
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04j; mkdir -p $O
HAMER_BENCH_REHEARSAL=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload e2e --hands4 --chunks 2 --steps 2 --warmup 1 --no-roofline > $O/e2e_2ranks.log 2>&1 || { tail -30 $O/e2e_2ranks.log; exit 1; }
tail -c 1200 $O/e2e_2ranks.log; echo
HAMER_BENCH_REHEARSAL=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $O/crops_2ranks.log 2>&1 || { tail -30 $O/crops_2ranks.log; exit 1; }
tail -c 700 $O/crops_2ranks.log; echo
