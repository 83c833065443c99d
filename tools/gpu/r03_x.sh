#!/bin/bash
# round 3: kernel trace of the e2e driver -- where is the GPU idle?
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03x; mkdir -p $O
rocprofv3 --kernel-trace -d $O/trace -o e2e --output-format csv -- python3 bench.py --workload e2e --hands4 --chunks 4 --steps 3 --warmup 1 > $O/e2e.log 2>&1 || { tail $O/e2e.log; exit 1; }
tail -c 400 $O/e2e.log
F=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
ev=sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r.get('Queue_Id','')) for r in rows)
t_end=ev[-1][1]
# last ~330 ms = three timed passes; analyse the last 300 ms
lo=t_end-300_000_000
ev=[e for e in ev if e[1]>lo]
# union busy time
busy=0; cur_s=cur_e=None; gaps=[]
for s,e,n,q in ev:
    if cur_e is None: cur_s,cur_e=s,e; continue
    if s>cur_e:
        busy+=cur_e-cur_s; gaps.append((s-cur_e,cur_e,n)); cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
span=ev[-1][1]-ev[0][0]
print('span ms',span/1e6,'busy ms',busy/1e6,'busy frac',busy/span)
gaps.sort(reverse=True)
print('largest gaps (us, before kernel):')
for g,t,n in gaps[:15]: print('  %8.1f  at %8.2f ms  %s'%(g/1e3,(t-ev[0][0])/1e6,n))
import collections
tot=collections.Counter()
for g,t,n in gaps: tot['%s'%( '>1ms' if g>1e6 else ('100us-1ms' if g>1e5 else ('10-100us' if g>1e4 else '<10us')))]+=g
print({k:round(v/1e6,2) for k,v in tot.items()})
PY
