#!/bin/bash
# Per-workload rocprofv3 kernel stats (one bench.py run per workload so that kernel averages match the bench line).
# Usage: gpurun -- 'bash tools/gpu_profiles.sh <tag>'
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG/stats
mkdir -p $OUT
export TMPDIR=/tmp
for WL in ${WLS:-sad16x16_modeA_1080p_8bit sad16x16_modeA_4k_8bit sad16x16_modeA_4k_10bit txq_1080p_8bit txq_4k_10bit search_4k_10bit inner_loop_4k_10bit cdef_search_4k_10bit wiener_stats_4k default_search_4k_10bit tf_motion_search_4k_10bit first_pass_4k_10bit compound_search_4k_10bit warp_error_4k int_pro_4k_8bit variance16x16_modeA_1080p_8bit sub_pixel_variance16x16_modeA_1080p_8bit variance16x16_modeA_4k_10bit sub_pixel_variance16x16_modeA_4k_10bit filters_ring_4k_10bit}; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$WL -o k -- \
      python3 bench.py --steps 20 --warmup 3 --workload $WL --others "" --no-cpu-baseline > $OUT/$WL.json 2> $OUT/$WL.err
  f=$(find $OUT/$WL -name '*kernel_stats.csv' | head -1)
  echo "== $WL"; [ -n "$f" ] && cut -c1-230 "$f" | head -6
  python - "$OUT/$WL.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read())
r=d.get("roofline")
print("   bench: value %.4g %s" % (d["value"], d["unit"]) + (("; dominant %s avg_launch_ms %.4f" % (r["kernel"], r["avg_launch_ms"])) if r and "kernel" in r else ""))
PY
done
