#!/bin/bash
# HBM-side bytes of sad_sb_kernel (FETCH_SIZE / WRITE_SIZE, each in its own rocprofv3 pass, --kernel-trace only) for
# the three Mode-A workloads; writes profiles/<tag>_pmc_sb_<workload>.json and profiles/traffic.json["<workload>:sb"].
# Usage: gpurun -- 'bash tools/gpu_pmc_sb.sh <tag>'
set -u
TAG=${1:-r01s}
export TMPDIR=/tmp
for WL in sad16x16_modeA_1080p_8bit sad16x16_modeA_4k_8bit sad16x16_modeA_4k_10bit; do
  OUT=gpurun_out/$TAG/pmcsb_$WL
  mkdir -p $OUT
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
    N=$(echo $C | tr ' ' '_')
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$N -o pmc -- \
        python bench.py --steps 3 --warmup 1 --workload $WL --others "" --no-cpu-baseline > $OUT/$N.json 2> $OUT/$N.err
  done
  python tools/pmc_traffic_sb.py $TAG $WL
done
