#!/bin/bash
# parity of everything that runs through the search kernels, then the search workloads' timings
python -m pytest tests/test_gpu_mcomp.py tests/test_gpu_full_pixel_search.py tests/test_gpu_tf.py tests/test_gpu_fp.py tests/test_gpu_fp_frame.py tests/test_gpu_me.py tests/test_gpu_simple_motion.py tests/test_gpu_single_motion.py tests/test_gpu_goldens.py tests/test_gpu_full_size.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -3
python bench.py --workload default_search_4k_10bit --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('default search: NSTEP ms %.4f  subpel 8-tap ms %.4f parity %s' % (d['full_pixel_search_NSTEP_ms_per_frame'], d['subpel_tree_8tap_ms_per_frame'], d['parity_sample_slot0']))"
python bench.py --workload tf_motion_search_4k_10bit --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tf: 10-bit ms/frame %.3f   8-bit %.3f' % (d['q30_mesh_pruned_when_close']['ms_per_filtered_frame'], d['same_pass_8bit']['q30_mesh_pruned_when_close']['ms_per_filtered_frame']))"
python bench.py --workload first_pass_4k_10bit --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('first pass:', {k: v for k, v in d.items() if 'ms' in k})"
