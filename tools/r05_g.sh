#!/bin/bash
# round 5: quick A/B loop for the sub-pel kernels -- parity of the searches, then the compound, inner-loop, default-search and TF bench legs
mkdir -p gpurun_out/r05g
timeout 900 python -m pytest tests/test_gpu_compound_subpel.py tests/test_gpu_joint_search.py tests/test_gpu_composites.py tests/test_gpu_single_caller.py tests/test_gpu_mcomp.py tests/test_gpu_tf.py tests/test_gpu_full_pixel_search.py tests/test_gpu_fp_frame.py tests/test_gpu_single_motion.py tests/test_gpu_tpl_inter.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
AOMHIP_BENCH_COMPOUND_BS=16 timeout 600 python bench.py --workload compound_search_4k_10bit --steps 10 --warmup 3 > gpurun_out/r05g/bench_compound.json 2> gpurun_out/r05g/bench_compound.err
python3 -c "
import json; d=json.load(open('gpurun_out/r05g/bench_compound.json')); print(json.dumps(d['by_block_size']))"
for WL in inner_loop_4k_10bit default_search_4k_10bit tf_motion_search_4k_10bit first_pass_4k_10bit; do
timeout 600 python bench.py --workload $WL --steps 40 --warmup 5 > gpurun_out/r05g/bench_$WL.json 2> gpurun_out/r05g/bench_$WL.err
python3 -c "
import json; d=json.load(open('gpurun_out/r05g/bench_$WL.json')); print('$WL', d.get('value'), d.get('ms_per_step'))"
done
