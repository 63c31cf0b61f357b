# round 3 A/B on one box: LIBS="path ..." x DBGS="0 4096" x workloads, REPS rounds; then the phase clocks of the profiling build
mkdir -p gpurun_out/${OUT:-r03d}
for R in $(seq 1 ${REPS:-2}); do
IFS=';' read -ra WORKS <<< "${WORK:-4k 8 64 320,48;1080p 8 64 240,64;4k 10 32 160,32}"
for W in "${WORKS[@]}"; do
  for LIBP in ${LIBS:-build/exp_base/libaomhip_exp.so build/exp/libaomhip_exp.so}; do
  for CFG in ${CFGS:-wide}; do
  for D in ${DBGS:-0}; do echo "args=$W lib=$LIBP cfg=$CFG dbg=$D"; AOMHIP_SB_CFG=$CFG AOMHIP_LIB=$LIBP AOMHIP_SB_DBG=$D timeout 300 python tools/gpu_ab_sadsb.py $W 2>&1 | grep -E '^\{"cell|rror' | cut -c1-400; done
  done
done; done; done > gpurun_out/${OUT:-r03d}/ab.log 2>&1
python3 - <<'PY'
import re,os
cur=None
for l in open('gpurun_out/%s/ab.log' % os.environ.get('OUT','r03d')):
    if l.startswith('args='): cur=l.strip()
    elif l.startswith('{'):
        m=re.search(r'"ms": ([0-9.]+)',l); i=re.search(r'"identical": (\w+)',l); f=re.search(r'"frac_of_8TBs": ([0-9.]+)',l)
        c=re.search(r'"cell": "([0-9,]+)"',l); rg=re.search(r'"range": (\d+)',l)
        print(cur, 'cell', c.group(1) if c else None, 'range', rg.group(1) if rg else None, 'ms', round(float(m.group(1)),4) if m else l[:100], 'frac', round(float(f.group(1)),3) if f else None, 'identical', i.group(1) if i else None)
    else: print(l.strip()[:200])
PY
if [ -n "$PROF" ]; then for D in $PROF; do AOMHIP_SB_DBG=$D AOMHIP_LIB=build/exp/libaomhip_exp_prof.so python tools/gpu_sb_prof.py ${PROFARGS:-4k 8 64 320,48} 2>&1 | tail -1 | tee -a gpurun_out/${OUT:-r03d}/prof.log; done; fi
