#!/bin/bash
# round 4: PMC counters of the stem pair kernel (separate passes, --pmc only)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r04af; mkdir -p $O
timeout -k 10 120 rocprofv3 -L > $O/avail.txt 2>&1 || true
grep -o "SQ_[A-Z0-9_]*" $O/avail.txt | sort -u | tr '\n' ' ' > $O/sq_names.txt; wc -w $O/sq_names.txt
timeout -k 10 200 python3 tools/probes/stem_pair_run.py 48 6 > $O/plain.log 2>&1 || { tail $O/plain.log; exit 1; }
tail -1 $O/plain.log
i=0
for PMC in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $PMC -d $O/p$i -o p$i --output-format csv -- python3 tools/probes/stem_pair_run.py 48 4 > $O/p$i.log 2>&1 || { echo "pass $i ($PMC) failed: $(tail -2 $O/p$i.log | tr '\n' ' ')"; continue; }
  echo "pass $i ($PMC) ok"
done
python3 tools/pmc_by_kernel.py $O/stem_pair_pmc.json conv_stem_pair_kernel $(find $O -name '*counter_collection.csv') > $O/summary.log 2>&1 || { tail $O/summary.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('gpurun_out/r04af/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_stem_pair_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(f"{k:28s} {sum(v)/len(v):16.1f}  ({len(v)} dispatches)")
PY
