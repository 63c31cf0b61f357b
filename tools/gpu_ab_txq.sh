#!/bin/bash
# A/B of xform_quant kernel variants. Usage: gpurun -- 'bash tools/gpu_ab_txq.sh "0 1"'
VALS=${1:-"0 1"}
python -m pytest tests/test_gpu_xform_quant.py -x -q -m gpu 2>&1 | tail -3
for round in 1 2; do
for v in $VALS; do
  echo "== AOMHIP_XQ_VARIANT=$v round $round"
  AOMHIP_XQ_VARIANT=$v python - <<'PY'
import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import bench, aom_av1_psy_amd as pkg, pyoracle as orc
ctx = pkg.capi.Context(0)
r = bench.run_txq(pkg, ctx, orc, 10, 2, False)
print("value %.4g blocks/s parity %s" % (r["value"], r["parity_frame0_16x16"]))
for k, v in r["per_size"].items():
    print("   %-6s %.4f ms  %.3f TB/s  frac %.3f" % (k, v["avg_launch_ms"], v["achieved_GBs"] / 1e3, v["frac"]))
PY
done
done
