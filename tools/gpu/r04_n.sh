#!/bin/bash
# round 4: e2e long pass (192 frames) with detector passes of 48 / 64 / 96 frames, same box
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04n; mkdir -p $O
for D in 48 64 96 64 48; do
  timeout -k 10 300 python3 bench.py --workload e2e --hands4 --chunks 12 --steps 3 --warmup 1 --no-roofline --det-frames $D > $O/e2e192_d$D.log 2>&1 || { tail -30 $O/e2e192_d$D.log; exit 1; }
  python3 -c "
import json,sys
r=json.loads(open('$O/e2e192_d$D.log').read().strip().splitlines()[-1]); print('det_frames $D: 192-frame pass', r['value'], 'hands/s', r['ms_per_step'], 'ms', r['config']['detector_passes_per_pass_rank0'], 'passes')"
done
timeout -k 10 300 python3 bench.py --workload e2e --hands4 --chunks 4 --steps 3 --warmup 1 --no-roofline > $O/e2e64.log 2>&1 || exit 1
tail -c 300 $O/e2e64.log
