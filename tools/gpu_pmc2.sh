#!/bin/bash
# Diagnostic PMC groups for one workload. Usage: gpurun -- 'bash tools/gpu_pmc2.sh <tag> <workload> [ENV=VAL]'
set -u
TAG=${1:-r01}; WL=${2:-sad16x16_modeA_4k_8bit}; EV=${3:-AOMHIP_NOP=1}
OUT=gpurun_out/$TAG/diag_$WL
mkdir -p $OUT
export TMPDIR=/tmp
export $EV
i=0
while read -r C; do
  [ -z "$C" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- \
      python bench.py --steps 3 --warmup 1 --workload $WL --others "" --no-cpu-baseline > $OUT/g$i.json 2> $OUT/g$i.err
  tail -2 $OUT/g$i.err
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU
SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL GRBM_GUI_ACTIVE GRBM_TA_BUSY
TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
TCC_REQ_sum TCC_BUSY_avr TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum
SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITE_sum TCC_NORMAL_WRITEBACK_sum
LIST
python - "$OUT" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/g*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "sad_" not in k and "xform_quant" not in k: continue
        import re
        m=re.search(r"xform_quant\w*<(\d+), (\d+)", k)
        short=("xq%sx%s"%m.groups()) if m else ("x4d" if "x4d" in k else "cand")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    print("==",k)
    for c,v in sorted(d.items()): print("  %-44s %.6g"%(c,sum(v)/len(v)))
PY
