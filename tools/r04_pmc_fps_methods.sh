#!/bin/bash
# instruction counts of the lean diamond kernel and of the general kernel per method (tools/r04_fps_methods.py under rocprofv3 --pmc)
export TMPDIR=/tmp
OUT=gpurun_out/r04_pmc_fps; mkdir -p $OUT
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/g1 -o pmc -- python3 tools/r04_fps_methods.py > $OUT/g1.txt 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/g2 -o pmc -- python3 tools/r04_fps_methods.py > $OUT/g2.txt 2>&1
python3 - <<'PY'
import csv,glob,collections
for g in ("g1","g2"):
    f=glob.glob("gpurun_out/r04_pmc_fps/%s/*counter_collection.csv"%g)[0]
    by=collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k=(int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-60:])
        by.setdefault(k,{})[r["Counter_Name"]]=float(r["Counter_Value"])
    # consecutive dispatches of the search kernels: group runs of identical kernel names in launch order (each config = 3 warm + 10 timed + ...)
    runs=[]
    for (d,k),c in sorted(by.items()):
        if "search" not in k and "diamond" not in k: continue
        w=c.get("SQ_WAVES",0)
        if w<=0: continue
        e={n:v/w for n,v in c.items() if n!="SQ_WAVES"}
        if runs and runs[-1][0]==k and d-runs[-1][2]<=2: runs[-1][1].append(e); runs[-1][2]=d
        else: runs.append([k,[e],d])
    # split long runs of the general kernel into configs of equal length is not possible here: print per-run averages and lengths
    for k,es,_ in runs:
        avg={n:sum(e[n] for e in es)/len(es) for n in es[0]}
        print(g, k[-45:], len(es), {n.replace("SQ_INSTS_",""):round(v,1) for n,v in avg.items()})
PY
