#!/bin/bash
# round 5: compound / OBMC full-pel kernels with blocks of up to 1024 px in registers -- parity, then the per-size bench legs
mkdir -p gpurun_out/r05d
timeout 1500 python -m pytest tests/test_gpu_compound_search.py tests/test_gpu_compound_fullpel.py tests/test_gpu_joint_search.py tests/test_gpu_composites.py tests/test_gpu_single_caller.py tests/test_gpu_single_motion.py tests/test_gpu_compound.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -20 > gpurun_out/r05d/pytest.log
cat gpurun_out/r05d/pytest.log
timeout 900 python bench.py --workload compound_search_4k_10bit --steps 10 --warmup 3 > gpurun_out/r05d/bench_compound.json 2> gpurun_out/r05d/bench_compound.err
tail -c 3000 gpurun_out/r05d/bench_compound.json
