export TMPDIR=/tmp
for M in "DIAMOND 4" "NSTEP 3"; do
  set -- $M
  O=gpurun_out/r02t/fps_$1
  mkdir -p $O
  for C in "TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_INSTS_VALU"; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/g1 -o pmc -- python3 tools/gpu_fps_methods.py $1 $2 > $O/g1.out 2> $O/g1.err
  done
  python3 - $O <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/g*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k="general" if "full_pixel_search_kernel" in r["Kernel_Name"] else ("diamond" if "fullpel_diamond_kernel" in r["Kernel_Name"] else None)
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items():
    m={c:sum(v)/len(v) for c,v in d.items()}
    print(sys.argv[1], k, {c:round(v) for c,v in m.items()}, "lookups/load %.1f" % (m["TCP_TOTAL_CACHE_ACCESSES_sum"]/m["SQ_INSTS_VMEM_RD"]))
PY
done
