#!/bin/bash
run() {
python bench.py --workload inner_loop_4k_10bit --steps 20 --warmup 3 --others "" --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fps %.1f ' % d['value'], {k: round(v['ms'], 4) for k, v in d['stages'].items() if 'fullpel' in k})"
}
for lib in w4 ; do for w in 2 4 8; do for r in 16 12 8; do for h in 1 0; do
echo -n "== lib=$lib waves=$w R=$r hint=$h: "; AOMHIP_BENCH_GRID_HINT=$h AOMHIP_SEARCH_CELL_WAVES=$w AOMHIP_SEARCH_CELL_R=$r AOMHIP_LIB=${lib:+build/exp/libaomhip_$lib.so} bash -c "$(declare -f run); run"
done; done; done; done
