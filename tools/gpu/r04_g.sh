#!/bin/bash
# round 4: the default bench line (contract + side configs)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04g; mkdir -p $O
T0=$(date +%s); timeout -k 10 600 python3 bench.py > $O/bench.json.log 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
tail -c 6000 $O/bench.json.log
echo; echo "wall seconds: $(( $(date +%s) - T0 ))"
