#!/bin/bash
# round 6 A/B on one box: experiment builds of sad_strip_kernel (explib/libsadsb_<name>.so, tools/sb_build_exp.sh) x workloads x rounds,
# then the phase clocks of the _prof builds.   OUT=<dir under gpurun_out> NAMES="base pd1 ..." REPS=3 PROF="base pd1" bash tools/sb_ab.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
OUT=gpurun_out/${OUT:-r06_sad}; mkdir -p $OUT
export AB_REPS=${AB_REPS:-30}
IFS=';' read -ra WORKS <<< "${WORK:-4k 8 64 320,48;1080p 8 64 240,64;4k 10 32 160,32}"
for R in $(seq 1 ${REPS:-3}); do
  for W in "${WORKS[@]}"; do
    for N in ${NAMES:-base}; do
      for D in ${DBGS:-0}; do
        echo "args=$W lib=$N dbg=$D"
        AOMHIP_SB_LIB=explib/libsadsb_$N.so AOMHIP_SB_DBG=$D timeout 300 python tools/gpu_ab_sadsb.py $W 2>&1 | grep -E '^\{"cell|rror' | cut -c1-400
      done
    done
  done
done > $OUT/ab.log 2>&1
python3 - "$OUT" <<'PY'
import re, sys, collections
cur = None
tab = collections.OrderedDict()
for l in open(sys.argv[1] + "/ab.log"):
    if l.startswith("args="):
        cur = l.strip()
    elif l.startswith("{"):
        m = re.search(r'"ms": ([0-9.]+)', l); i = re.search(r'"identical": (\w+)', l); f = re.search(r'"frac_of_8TBs": ([0-9.]+)', l)
        if m:
            tab.setdefault(cur, []).append((float(m.group(1)), float(f.group(1)) if f else None, i.group(1) if i else None))
        else:
            print(cur, l.strip()[:200])
    else:
        print(cur, l.strip()[:200])
for k, v in tab.items():
    print(k, "ms", " / ".join("%.4f" % x[0] for x in v), "frac", " / ".join("%.3f" % x[1] for x in v), "identical", ",".join(str(x[2]) for x in v))
PY
for N in ${PROF:-}; do
  for D in ${PROFDBG:-0}; do
    IFS=';' read -ra PW <<< "${PROFARGS:-4k 8 64 320,48}"
    for W in "${PW[@]}"; do
      echo "prof lib=$N dbg=$D args=$W"
      AOMHIP_SB_DBG=$D AOMHIP_SB_LIB=explib/libsadsb_${N}_prof.so timeout 300 python tools/gpu_sb_prof.py $W 2>&1 | tail -1
    done
  done
done | tee $OUT/prof.log
