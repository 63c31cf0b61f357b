#!/bin/bash
# round 5: the global-motion model error on the device -- parity, bench, kernel trace
mkdir -p gpurun_out/r05j
timeout 600 python -m pytest tests/test_gpu_warp_error.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --workload warp_error_4k --steps 10 --warmup 2 > gpurun_out/r05j/bench_warp_error.json 2> gpurun_out/r05j/bench_warp_error.err
tail -c 1500 gpurun_out/r05j/bench_warp_error.json; tail -3 gpurun_out/r05j/bench_warp_error.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05j/prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload warp_error_4k --steps 10 --warmup 2 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r05j/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -6 "$f"; cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r05j/warp_error_kernel_stats.csv; fi
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r05j/prof
