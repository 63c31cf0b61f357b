#!/bin/bash
# round 5: parity of the compound / OBMC searches, per-kernel times of their bench legs at one block size (rocprofv3 --kernel-trace --stats),
# instruction counters of the 16x16 legs (tools/r04_pmc_stages.sh), then the bench legs by size
export TMPDIR=/tmp
mkdir -p gpurun_out/r05e
timeout 900 python -m pytest tests/test_gpu_compound_search.py tests/test_gpu_compound_fullpel.py tests/test_gpu_compound_subpel.py tests/test_gpu_joint_search.py tests/test_gpu_composites.py tests/test_gpu_single_caller.py tests/test_gpu_single_motion.py tests/test_gpu_compound.py tests/test_gpu_obmc_subpel.py tests/test_gpu_mcomp.py tests/test_gpu_tf.py tests/test_gpu_tpl_inter.py tests/test_gpu_full_pixel_search.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -20 > gpurun_out/r05e/pytest.log
cat gpurun_out/r05e/pytest.log
for bs in 8 16; do
  export AOMHIP_BENCH_COMPOUND_BS=$bs
  mkdir -p gpurun_out/r05e/bs$bs
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bs$bs -o cs -- python3 bench.py --workload compound_search_4k_10bit --steps 6 --warmup 2 > gpurun_out/r05e/bs$bs/bench.json 2> gpurun_out/r05e/bs$bs/err.log
  f=$(find /tmp/prof_bs$bs -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" gpurun_out/r05e/bs$bs/kernel_stats.csv; head -8 "$f" | cut -c1-60,300-420; fi
done
export AOMHIP_BENCH_COMPOUND_BS=16
WL=compound_search_4k_10bit TAG=_compound timeout 900 bash tools/r04_pmc_stages.sh > gpurun_out/r05e/pmc.log 2>&1
tail -12 gpurun_out/r05e/pmc.log | cut -c1-600
rm -rf gpurun_out/r04_pmc_stages_compound/g*/
unset AOMHIP_BENCH_COMPOUND_BS
timeout 600 python bench.py --workload compound_search_4k_10bit --steps 10 --warmup 3 > gpurun_out/r05e/bench_compound.json 2> gpurun_out/r05e/bench_compound.err
python3 -c "
import json; d=json.load(open('gpurun_out/r05e/bench_compound.json')); print(json.dumps(d['by_block_size'], indent=0))"
timeout 600 python bench.py --workload inner_loop_4k_10bit --steps 60 --warmup 5 > gpurun_out/r05e/bench_inner.json 2> gpurun_out/r05e/bench_inner.err
python3 -c "
import json; d=json.load(open('gpurun_out/r05e/bench_inner.json')); print('inner loop', d.get('value'), d.get('ms_per_step')); print(json.dumps(d.get('stages_ms', d.get('stages', {})))[:1500])"
timeout 600 python bench.py --workload tf_motion_search_4k_10bit --steps 10 --warmup 3 > gpurun_out/r05e/bench_tf.json 2> gpurun_out/r05e/bench_tf.err
python3 -c "
import json; d=json.load(open('gpurun_out/r05e/bench_tf.json')); print('tf', d.get('value'), d.get('ms_per_step'))"
