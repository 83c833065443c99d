#!/bin/bash
# round 4: what the per-image K-group rule costs / buys per layer at 48 frames and at one frame (CONV_KGROUPS=1: never)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04aq; mkdir -p $O
for F in 48 1; do
  for KG in 0 1; do
    CONV_KGROUPS=$KG timeout -k 10 300 python3 tools/prof_yolo.py $F > $O/yolo${F}_kg$KG.log 2>&1 || { tail -20 $O/yolo${F}_kg$KG.log; exit 1; }
    echo "frames $F CONV_KGROUPS=$KG: $(tail -2 $O/yolo${F}_kg$KG.log | tr '\n' '|')"
  done
done
python3 - <<'PY'
import re
def rows(f):
    out = {}
    for l in open(f):
        m = re.match(r'\s*(\d+) (k\ds\d)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)', l)
        if m: out[int(m.group(1))] = (m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(5)), float(m.group(6)))
    return out
for F in (48, 1):
    a, b = rows(f'gpurun_out/r04aq/yolo{F}_kg0.log'), rows(f'gpurun_out/r04aq/yolo{F}_kg1.log')
    for i in sorted(a):
        if abs(a[i][4] - b[i][4]) > 0.04 * a[i][4] and a[i][4] > 5:
            print(F, i, a[i][:4], 'rule', a[i][4], 'never', b[i][4])
PY
