#!/bin/bash
# Texture-address / L1 counters for the kernels of one bench workload.  Usage: gpurun -- 'bash tools/gpu_pmc_ta.sh <tag> <workload> <kernel-substring>'
set -u
TAG=${1:-r02s}; WL=${2:-inner_loop_4k_10bit}; KS=${3:-fullpel_diamond}
OUT=gpurun_out/$TAG/ta_$KS
mkdir -p $OUT
export TMPDIR=/tmp
i=0
while read -r C; do
  [ -z "$C" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- \
      python3 bench.py --steps 2 --warmup 1 --workload $WL --others "" --no-cpu-baseline > $OUT/g$i.json 2> $OUT/g$i.err
done <<'LIST'
GRBM_GUI_ACTIVE GRBM_TA_BUSY
TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
LIST
python3 - "$OUT" "$KS" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/g*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c,v in sorted(acc.items()): print("  %-44s %.6g  (n=%d)"%(c,sum(v)/len(v),len(v)))
PY
