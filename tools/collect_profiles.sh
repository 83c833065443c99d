#!/bin/bash
# Round profile collection on the GPU box (run from the repo root through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the contract bench command      -> gpurun_out/prof_$TAG/
#   2. three separate PMC passes over tools/pmc_shapes.py (one counter set each; never combined with trace domains)
#      FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE   -> gpurun_out/pmc_$TAG/{fetch,write,mfma}
#   3. tools/pmc_parse.py                                                   -> gpurun_out/pmc_$TAG/*.json
# Copy what should be judged into profiles/ afterwards (gpurun_out/ is scratch).
set -o pipefail
TAG=${1:-r03}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT/prof_$TAG $OUT/pmc_$TAG
rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o ${TAG} --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side > $OUT/prof_$TAG/bench_under_rocprof.log 2>&1 || exit 1
tail -c 300 $OUT/prof_$TAG/bench_under_rocprof.log
# the same bench with ONE batch in flight: per-kernel durations free of the other stream (what bench.py's event-timed pass
# measures; under two batches in flight a kernel's start-to-end time includes waiting for CUs the other stream holds)
mkdir -p $OUT/prof_${TAG}_serial
rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_serial -o ${TAG}_serial --output-format csv -- python3 bench.py --steps 10 --warmup 3 --in-flight 1 --no-cpu-baseline --no-side > $OUT/prof_${TAG}_serial/bench_under_rocprof.log 2>&1 || exit 1
tail -c 300 $OUT/prof_${TAG}_serial/bench_under_rocprof.log
for pass in fetch write mfma; do
  case $pass in
    fetch) PMC="FETCH_SIZE" ;;
    write) PMC="WRITE_SIZE" ;;
    mfma) PMC="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" ;;
  esac
  rocprofv3 --pmc $PMC -d $OUT/pmc_$TAG/$pass -o $pass --output-format csv -- python3 tools/pmc_shapes.py > $OUT/pmc_$TAG/$pass.log 2>&1 || exit 1
  echo "pmc pass $pass done"
done
F=$(find $OUT/pmc_$TAG/fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/pmc_$TAG/write -name '*counter_collection.csv' | head -1)
Mf=$(find $OUT/pmc_$TAG/mfma -name '*counter_collection.csv' | head -1)
python3 tools/pmc_parse.py "$F" "$W" "$Mf" $OUT/pmc_$TAG/${TAG}_pmc_traffic.json $OUT/pmc_$TAG/${TAG}_pmc_mfma_busy.json
cp "$F" $OUT/pmc_$TAG/fetch_size_counter_collection.csv; cp "$W" $OUT/pmc_$TAG/write_size_counter_collection.csv; cp "$Mf" $OUT/pmc_$TAG/mfma_busy_counter_collection.csv

# 4. (round 3) the YOLOv7 detector pass: kernel trace of tools/prof_yolo.py for 16 frames of 1080p in one pass, and for one frame
for F in 16 1; do  # (round 4 adds 48 frames in tools/gpu/r04_collect.sh)
  mkdir -p $OUT/prof_${TAG}_yolo$F
  rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_yolo$F -o ${TAG}_yolo$F --output-format csv -- python3 tools/prof_yolo.py $F 3 > $OUT/prof_${TAG}_yolo$F/prof_yolo.log 2>&1 || exit 1
  tail -2 $OUT/prof_${TAG}_yolo$F/prof_yolo.log
done
