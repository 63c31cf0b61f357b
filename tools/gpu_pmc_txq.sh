#!/bin/bash
# HBM-side bytes of the transform + quantise kernels (FETCH_SIZE and WRITE_SIZE, each in its own rocprofv3 pass, --kernel-trace
# only) with in-run calibration launches of known traffic (bench.py pmc_calibration_ops); writes profiles/<tag>_pmc_txq.json and
# profiles/traffic.json["txq_<size>"].   Usage: gpurun -- 'bash tools/gpu_pmc_txq.sh <tag>'
set -u
TAG=${1:-r02}
WL=${2:-txq_1080p_8bit}
export TMPDIR=/tmp
export AOMHIP_PMC_CALIB=1
OUT=gpurun_out/$TAG/pmctxq
[ "$WL" != "txq_1080p_8bit" ] && OUT=gpurun_out/$TAG/pmc$WL
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o pmc -- \
      python3 bench.py --steps 3 --warmup 1 --workload $WL --no-cpu-baseline > $OUT/$C.json 2> $OUT/$C.err
done
python3 tools/pmc_traffic_txq.py $TAG $WL
