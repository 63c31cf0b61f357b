#!/bin/bash
# PMC passes (HBM-side bytes) for the SAD kernels, each counter in its own rocprofv3 run with
# --kernel-trace only (MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do
# not fit one pass).  Usage: gpurun -- 'bash tools/gpu_pmc.sh <tag> <workload>'
set -u
TAG=${1:-r01}; WL=${2:-sad16x16_modeA_4k_8bit}
OUT=gpurun_out/$TAG/pmc_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$N -o pmc -- \
      python bench.py --steps 5 --warmup 1 --workload $WL --others "" --no-cpu-baseline > $OUT/$N.json 2> $OUT/$N.err
  ls $OUT/$N | head -5
done
f=$(find $OUT/FETCH_SIZE -name '*counter_collection.csv' | head -1); [ -n "$f" ] && head -5 "$f"
