#!/bin/bash
# PMC passes for the superblock-bucketed SAD kernel. Usage: gpurun -- 'bash tools/gpu_pmc3.sh <tag> <workload>'
set -u
TAG=${1:-r01s}; WL=${2:-sad16x16_modeA_4k_8bit}
OUT=gpurun_out/$TAG/sb_$WL
mkdir -p $OUT
export TMPDIR=/tmp
i=0
while read -r C; do
  [ -z "$C" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- \
      python bench.py --steps 3 --warmup 1 --workload $WL --others "" --no-cpu-baseline > $OUT/g$i.json 2> $OUT/g$i.err
  tail -2 $OUT/g$i.err
done <<'LIST'
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_TA_BUSY
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU
SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD
LIST
python - "$OUT" "$TAG" "$WL" <<'PY'
import csv,glob,sys,collections,json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/g*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "sad_" not in k: continue
        short="sad_sb" if "sad_sb" in k else "sad_x4d" if "x4d" in k else "sad_cand"
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={k:{c:sum(v)/len(v) for c,v in d.items()} for k,d in acc.items()}
json.dump(out, open("gpurun_out/%s/pmc_sb_%s.json"%(sys.argv[2],sys.argv[3]),"w"), indent=1, sort_keys=True)
for k,d in out.items():
    print("==",k)
    for c,v in sorted(d.items()): print("  %-36s %.6g"%(c,v))
PY
