#!/bin/bash
# round 5: per-kernel times of the compound / OBMC search legs at one block size (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for bs in 8 64; do
  export AOMHIP_BENCH_COMPOUND_BS=$bs
  mkdir -p $R/gpurun_out/r05e/bs$bs
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bs$bs -o cs -- python3 $R/bench.py --workload compound_search_4k_10bit --steps 6 --warmup 2 > $R/gpurun_out/r05e/bs$bs/bench.json 2> $R/gpurun_out/r05e/bs$bs/err.log
  f=$(find /tmp/prof_bs$bs -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/r05e/bs$bs/kernel_stats.csv
  head -25 $f | cut -c1-200
done
