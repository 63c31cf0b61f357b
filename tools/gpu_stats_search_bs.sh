# per-kernel rocprofv3 times of tools/gpu_ab_search_bs.py: bash tools/gpu_stats_search_bs.sh "<bs> <bd>" ...
export TMPDIR=/tmp
for A in "$@"; do
  D=gpurun_out/bs_tmp; rm -rf $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o k -- python3 tools/gpu_ab_search_bs.py $A > /dev/null 2>&1
  python3 - "$A" $(find $D -name "*kernel_stats.csv" | head -1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[2])))[:2]:
    print(sys.argv[1], r["Name"][13:60], "avg %.4f ms" % (float(r["AverageNs"]) / 1e6))
PY
  rm -rf $D
done
