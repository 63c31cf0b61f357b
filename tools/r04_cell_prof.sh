#!/bin/bash
python -m pytest tests/test_gpu_mcomp.py tests/test_gpu_full_size.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -1
AOMHIP_SEARCH_CELL=0 python -m pytest tests/test_gpu_mcomp.py tests/test_gpu_full_size.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -1
for lib in prof4; do
for env in "A=1" "AOMHIP_SEARCH_CELL_R=32" "AOMHIP_SEARCH_CELL_WAVES=8"; do
echo "== $lib $env"
env $env AOMHIP_LIB=build/exp/libaomhip_$lib.so python tools/r04_cell_prof.py 2>&1 | tail -4
done; done
run() {
python bench.py --workload inner_loop_4k_10bit --steps 20 --warmup 3 --others "" --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('inner loop: fps %.1f ' % d['value'], {k: round(v['ms'], 4) for k, v in d['stages'].items() if 'pel' in k})"
}
echo "== default lib, cell off"; AOMHIP_SEARCH_CELL=0 run
echo "== default lib (min8)"; run
for lib in mw4; do for env in "A=1" "AOMHIP_SEARCH_CELL_R=32" "AOMHIP_SEARCH_CELL_WAVES=8" "AOMHIP_SEARCH_CELL_WAVES=8 AOMHIP_SEARCH_CELL_R=32"; do
echo "== $lib $env"; env $env AOMHIP_LIB=build/exp/libaomhip_$lib.so bash -c "$(declare -f run); run"
done; done
