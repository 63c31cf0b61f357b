# 8-bit against 10-bit: the search / filter workloads of bench.py at both plane depths on the same box (bench.py --bit-depth), parity samples included.
# An 8-bit plane is half the bytes and its kernels are the same templates: a stage that is SLOWER at 8 bits points at an instantiation that compiled
# badly (round 6: the 8-bit sub-pel kernels kept a lambda as a real call and lived in scratch memory -- the temporal filter's 8-bit pass 8.4 vs 5.3 ms).
#   bash tools/bd_audit.sh [out_dir]   ->  <out_dir>/<workload>_bd<depth>.json + a table on stdout
set -u; export TMPDIR=/tmp
OUT=${1:-gpurun_out/bd_audit}; mkdir -p $OUT
for w in search_4k_10bit default_search_4k_10bit first_pass_4k_10bit inner_loop_4k_10bit cdef_search_4k_10bit compound_search_4k_10bit; do
  for bd in 10 8; do
    python bench.py --workload $w --bit-depth $bd --steps 12 --warmup 2 --no-cpu-baseline > $OUT/${w}_bd$bd.json 2> $OUT/${w}_bd$bd.err || echo "$w bd$bd: rc $?"
  done
done
python - $OUT <<'PY'
import json, sys, glob, os
def flat(o, p=""):
    if isinstance(o, dict):
        for k, v in o.items():
            yield from flat(v, p + "/" + k)
    elif isinstance(o, (int, float)) and not isinstance(o, bool) and ("ms" in p.split("/")[-1] or "_us" in p or p.endswith("us")):
        yield p, float(o)
out = sys.argv[1]
for f10 in sorted(glob.glob(out + "/*_bd10.json")):
    f8 = f10.replace("_bd10", "_bd8")
    try:
        a = dict(flat(json.loads([l for l in open(f10) if l.startswith("{")][-1])))
        b = dict(flat(json.loads([l for l in open(f8) if l.startswith("{")][-1])))
    except Exception as e:
        print(os.path.basename(f10), "unreadable:", e); continue
    print("==", os.path.basename(f10)[:-10])
    for k in a:
        if k in b and a[k] > 0:
            print("  %-90s 10-bit %10.4f   8-bit %10.4f   %s" % (k[-90:], a[k], b[k], "8-BIT SLOWER" if b[k] > 1.05 * a[k] else ""))
PY
