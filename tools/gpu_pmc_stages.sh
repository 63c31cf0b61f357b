#!/bin/bash
# Instruction counters of the inner loop's kernels (verdict r03 item 6: a VALU roofline per stage).  Two separate --pmc passes (no trace domains
# beside the kernel trace), then tools/gpu_pmc_stages.py writes profiles/r04_inner_loop_pmc.json.
# Usage: gpurun -- '[WL=<bench workload> TAG=_<name>] bash tools/gpu_pmc_stages.sh'
set -u
WL=${WL:-inner_loop_4k_10bit}
OUT=gpurun_out/gpu_pmc_stages${TAG:-}; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- python3 bench.py --steps 3 --warmup 1 --workload $WL --others "" --no-cpu-baseline > $OUT/g$i.json 2> $OUT/g$i.err
done
python3 tools/gpu_pmc_stages.py $OUT
