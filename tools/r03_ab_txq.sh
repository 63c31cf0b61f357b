#!/bin/bash
# round 3 A/B of the transform + quantise kernels across library builds ("" = shipped): bash tools/r03_ab_txq.sh "build/ab/libaomhip_fastbtf.so"
LIBS=${1:-""}
mkdir -p gpurun_out/${OUT:-r03p}
for round in 1 2; do
for v in "" $LIBS; do
for w in txq_1080p_8bit txq_4k_10bit; do
  echo "== lib=${v:-shipped} $w round $round"
  AOMHIP_LIB=$v python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.4g blocks/s' % d['value'], {k: (round(v['avg_launch_ms'], 4), round(v['frac'], 3)) for k, v in d['per_size'].items()})"
done; done; done 2>&1 | tee gpurun_out/${OUT:-r03p}/ab_txq.log
