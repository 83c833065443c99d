#!/bin/bash
# round 4 evidence: tools/collect_profiles.sh (contract command, serial twin, PMC over the step's GEMM shapes, YOLO 16 / 1), plus
# the YOLO pass at the new driver's 48 frames and the fp8 configuration (configs[4], B = 256): kernel stats + three PMC passes
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
bash tools/collect_profiles.sh r04 > $OUT/collect_r04.log 2>&1 || { tail -30 $OUT/collect_r04.log; exit 1; }
tail -5 $OUT/collect_r04.log
mkdir -p $OUT/prof_r04_yolo48
rocprofv3 --kernel-trace --stats -d $OUT/prof_r04_yolo48 -o r04_yolo48 --output-format csv -- python3 tools/prof_yolo.py 48 3 > $OUT/prof_r04_yolo48/prof_yolo.log 2>&1 || exit 1
tail -2 $OUT/prof_r04_yolo48/prof_yolo.log
mkdir -p $OUT/prof_r04_fp8 $OUT/pmc_r04_fp8
rocprofv3 --kernel-trace --stats -d $OUT/prof_r04_fp8 -o r04_fp8 --output-format csv -- python3 bench.py --dtype fp8 --batch 256 --in-flight 1 --steps 6 --warmup 2 --no-cpu-baseline --no-side > $OUT/prof_r04_fp8/bench_under_rocprof.log 2>&1 || exit 1
tail -c 400 $OUT/prof_r04_fp8/bench_under_rocprof.log; echo
for pass in fetch write mfma; do
  case $pass in
    fetch) PMC="FETCH_SIZE" ;;
    write) PMC="WRITE_SIZE" ;;
    mfma) PMC="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" ;;
  esac
  rocprofv3 --pmc $PMC -d $OUT/pmc_r04_fp8/$pass -o $pass --output-format csv -- python3 bench.py --dtype fp8 --batch 256 --in-flight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-side --no-roofline > $OUT/pmc_r04_fp8/$pass.log 2>&1 || exit 1
  echo "fp8 pmc pass $pass done"
done
python3 tools/pmc_by_kernel.py $OUT/pmc_r04_fp8/r04_pmc_fp8_b256.json gemm_fp8p_kernel,gemm_fp8_kernel,layernorm_mx8_kernel,vit_attention_kernel $(find $OUT/pmc_r04_fp8 -name '*counter_collection.csv') > $OUT/pmc_r04_fp8/summary.log 2>&1 || { tail $OUT/pmc_r04_fp8/summary.log; exit 1; }
tail -40 $OUT/pmc_r04_fp8/summary.log
