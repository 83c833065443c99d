#!/bin/bash
# round 4: 1x1 layers on the persistent GEMM kernel: tests, then the 48-frame table with it (automatic) and without (tile rule only)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04ag; mkdir -p $O
timeout -k 10 300 python3 tools/probes/stem_pair_debug.py > $O/debug.log 2>&1 || { tail -40 $O/debug.log; exit 1; }
grep differ $O/debug.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_yolo.py -x -q -m gpu -k "persistent or stem" > $O/t1.log 2>&1 || { tail -40 $O/t1.log; exit 1; }
tail -1 $O/t1.log
timeout -k 10 300 python3 tools/prof_yolo.py 48 > $O/yolo48_auto.log 2>&1 || { tail -20 $O/yolo48_auto.log; exit 1; }
CONV_TILE=16 timeout -k 10 300 python3 tools/prof_yolo.py 48 > $O/yolo48_px_forced.log 2>&1 || { tail -20 $O/yolo48_px_forced.log; exit 1; }
CONV_TILE=0 timeout -k 10 300 python3 tools/prof_yolo.py 16 > $O/yolo16_auto.log 2>&1 || { tail -20 $O/yolo16_auto.log; exit 1; }
python3 - <<'PY'
import re
def rows(f):
    out = {}
    for l in open(f):
        m = re.match(r'\s*(\d+) (k\ds\d)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)', l)
        if m: out[int(m.group(1))] = (m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(5)), float(m.group(6)))
    return out
a, b = rows('gpurun_out/r04ag/yolo48_auto.log'), rows('gpurun_out/r04ag/yolo48_px_forced.log')
for i in sorted(a):
    if a[i][0] == 'k1s1' and a[i][2] % 256 == 0 and a[i][3] >= 256:
        print(i, a[i][:4], 'auto', a[i][4], 'px forced', b[i][4])
for f in ('yolo48_auto', 'yolo48_px_forced', 'yolo16_auto'):
    print(f, open(f'gpurun_out/r04ag/{f}.log').read().strip().splitlines()[-1])
PY
