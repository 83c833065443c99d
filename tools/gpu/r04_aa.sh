#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04aa; mkdir -p $O
for SI in 0.005 0.0005 0.0001; do
  timeout -k 10 300 python3 -c "
import sys; sys.setswitchinterval($SI); sys.argv=['e2e_trace.py','64']; exec(open('tools/probes/e2e_trace.py').read())" > $O/trace_$SI.log 2>&1 || { tail -30 $O/trace_$SI.log; exit 1; }
  echo "switch interval $SI: $(grep '^pass' $O/trace_$SI.log) | $(grep -m1 det_enqueued $O/trace_$SI.log) | $(grep -m1 batch_enqueued $O/trace_$SI.log)"
done
