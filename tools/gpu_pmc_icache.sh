set -u; export TMPDIR=/tmp
OUT=gpurun_out/r06p/ic; mkdir -p $OUT
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/g$i -o pmc -- python bench.py --steps 2 --warmup 1 --workload tf_motion_search_4k_10bit --others "" --no-cpu-baseline > $OUT/g$i.json 2> $OUT/g$i.err || tail -3 $OUT/g$i.err
done
python - $OUT <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/g*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "subpel_bilinear" in k or "full_pixel_search" in k:
            acc[k[:75]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    print(k)
    for c,v in sorted(d.items()): print("   %-30s %.5g (n=%d)"%(c,sum(v)/len(v),len(v)))
PY
