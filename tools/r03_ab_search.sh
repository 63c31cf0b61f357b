#!/bin/bash
# round 3 A/B of the search kernels: phase planes on / off in the same library (AOMHIP_SEARCH_PHASE_PLANES), optionally other libraries too
mkdir -p gpurun_out/${OUT:-r03o}
python -m pytest tests/test_gpu_mcomp.py tests/test_gpu_full_pixel_search.py tests/test_gpu_tf.py tests/test_gpu_goldens.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2; do
for v in 1 0; do
echo "== phase planes=$v"
export AOMHIP_SEARCH_PHASE_PLANES=$v
python bench.py --workload inner_loop_4k_10bit --steps 20 --warmup 3 --others "" --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('inner loop: fps %.1f ' % d['value'], {k: round(v['ms'], 4) for k, v in d['stages'].items()})"
python bench.py --workload default_search_4k_10bit --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('default search: NSTEP ms %.4f  subpel 8-tap ms %.4f' % (d['full_pixel_search_NSTEP_ms_per_frame'], d['subpel_tree_8tap_ms_per_frame']))"
python bench.py --workload tf_motion_search_4k_10bit --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tf: 10-bit ms/frame %.3f   8-bit %.3f' % (d['q30_mesh_pruned_when_close']['ms_per_filtered_frame'], d['same_pass_8bit']['q30_mesh_pruned_when_close']['ms_per_filtered_frame']))"
done; done 2>&1 | tee gpurun_out/${OUT:-r03o}/ab_search.log
