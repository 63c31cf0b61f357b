#!/bin/bash
# PMC diagnostics for the kernels of one bench workload.  Usage: gpurun -- 'bash tools/gpu_pmc_kernel.sh <tag> <workload> <kernel-substring>'
set -u
TAG=${1:-r01}; WL=${2:-inner_loop_4k_10bit}; KS=${3:-cdef_luma}
OUT=gpurun_out/$TAG/pmc_$(echo "$KS" | tr -c "A-Za-z0-9_" "_")
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o k -- python bench.py --steps 4 --warmup 1 --workload $WL --others "" --no-cpu-baseline > $OUT/trace.json 2> $OUT/trace.err
grep -h "$KS" $OUT/trace/*kernel_stats.csv | cut -c1-60,200-400 | head -3
i=0
while read -r C; do
  [ -z "$C" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- \
      python bench.py --steps 2 --warmup 1 --workload $WL --others "" --no-cpu-baseline > $OUT/g$i.json 2> $OUT/g$i.err
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_IFETCH SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC
LIST
python - "$OUT" "$KS" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/g*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c,v in sorted(acc.items()): print("  %-28s %.6g  (n=%d)"%(c,sum(v)/len(v),len(v)))
PY
