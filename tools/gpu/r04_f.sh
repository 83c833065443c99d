#!/bin/bash
# round 4: the whole GPU suite after the prescale / serial-range / folder-driver changes
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04f; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu --durations=15 > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -25 $O/t.log
