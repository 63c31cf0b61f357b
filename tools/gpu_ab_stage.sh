#!/bin/bash
# A/B of one inner-loop stage across library builds: gpurun -- 'bash tools/gpu_ab_stage.sh <stage> "lib1.so lib2.so"'
STAGE=$1; LIBS=${2:-""}
for r in 1 2; do
for v in "" $LIBS; do
AOMHIP_LIB=$v python bench.py --workload inner_loop_4k_10bit --steps 20 --warmup 3 --others "" --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lib=%s  %s ms %.4f   frame fps %.1f' % ('${v:-shipped}', '$STAGE', d['stages']['$STAGE']['ms'], d['value']))"
done; done
