#!/bin/bash
# A/B of full_pixel_search_kernel occupancy targets (AOMHIP_FPS_CELL_WAVES): NSTEP, TF, first pass per library
run() {
python bench.py --workload default_search_4k_10bit --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  NSTEP ms %.4f parity %s' % (d['full_pixel_search_NSTEP_ms_per_frame'], d['parity_sample_slot0']))"
python bench.py --workload tf_motion_search_4k_10bit --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  tf: 10-bit ms/frame %.3f   8-bit %.3f' % (d['q30_mesh_pruned_when_close']['ms_per_filtered_frame'], d['same_pass_8bit']['q30_mesh_pruned_when_close']['ms_per_filtered_frame']))"
python bench.py --workload first_pass_4k_10bit --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  first pass ms %.3f (zero legs %.3f)' % (d['ms_per_frame'], d['ms_zero_mv_legs_only']))"
}
for lib in "" fpsw5 fpsw6; do echo "== lib=${lib:-default(4)}"; AOMHIP_LIB=${lib:+build/exp/libaomhip_$lib.so} bash -c "$(declare -f run); run"; done
