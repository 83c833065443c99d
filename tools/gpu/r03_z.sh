#!/bin/bash
# round 3: PMC passes over the detector pass (16 frames): LDS conflicts, MFMA busy, traffic of the convolution kernels
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r03z; mkdir -p $O
i=0
for PMC in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC -d $O/p$i -o p$i --output-format csv -- python3 tools/prof_yolo.py 16 1 > $O/p$i.log 2>&1 || { tail $O/p$i.log; exit 1; }
  echo "pass $i ($PMC) done"
done
python3 tools/pmc_conv.py $O/r03_pmc_conv.json $(find $O -name '*counter_collection.csv' | sort) > $O/summary.txt
cat $O/summary.txt
