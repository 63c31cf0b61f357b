# A/B on one box: the lean (diamond / n-step only) instantiation of full_pixel_search_kernel vs the general one (AOMHIP_FPS_LEAN=0)
set -u
python -m pytest tests/test_gpu_full_pixel_search.py tests/test_gpu_mcomp.py tests/test_gpu_tf.py tests/test_gpu_fp.py tests/test_gpu_fp_frame.py tests/test_gpu_composites.py tests/test_gpu_me.py tests/test_gpu_single_caller.py -q 2>&1 | tail -3
for R in 1 2; do for L in 1 0; do
  echo "== AOMHIP_FPS_LEAN=$L round $R"
  AOMHIP_FPS_LEAN=$L python bench.py --workload default_search_4k_10bit --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().splitlines()[-1]);print({k:round(v,4) for k,v in d.items() if k.endswith('ms_per_frame')})"
  AOMHIP_FPS_LEAN=$L python bench.py --workload tf_motion_search_4k_10bit --steps 6 --warmup 1 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().splitlines()[-1]);print('tf', d['q30_mesh_pruned_when_close'], d['q12_mesh_always'], d['same_pass_8bit']['q30_mesh_pruned_when_close'])"
done; done
python bench.py --workload first_pass_4k_10bit --steps 4 --warmup 1 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().splitlines()[-1]);print('first pass', d['ms_per_frame'])"
