#!/bin/bash
# round 5: the projection-based motion estimation on the device -- parity, bench, kernel trace
mkdir -p gpurun_out/r05k
timeout 600 python -m pytest tests/test_gpu_intpro.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --workload int_pro_4k_8bit --steps 10 --warmup 2 > gpurun_out/r05k/bench_int_pro.json 2> gpurun_out/r05k/bench_int_pro.err
tail -c 1500 gpurun_out/r05k/bench_int_pro.json; tail -3 gpurun_out/r05k/bench_int_pro.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05k/prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload int_pro_4k_8bit --steps 10 --warmup 2 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r05k/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -4 "$f" | cut -c1-300; cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r05k/int_pro_kernel_stats.csv; fi
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r05k/prof
