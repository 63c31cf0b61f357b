#!/bin/bash
# PMC diagnostics of the strip SAD kernel through tools/gpu_ab_sadsb.py.  Usage: gpurun -- 'bash tools/gpu_pmc_ab.sh <tag> <lib or ""> <ab args...>'
set -u
TAG=$1; LIB=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ -n "$LIB" ] && export AOMHIP_LIB=$LIB
i=0
while read -r C; do
  [ -z "$C" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- python3 tools/gpu_ab_sadsb.py "$@" > $OUT/g$i.json 2> $OUT/g$i.err
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC
LIST
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/g*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sad_strip" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c,v in sorted(acc.items()): print("  %-28s %.6g  (n=%d)"%(c,sum(v)/len(v),len(v)))
PY
