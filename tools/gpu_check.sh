#!/bin/bash
# Run on the GPU box via gpurun: parity tests, smoke, bench, rocprof kernel stats.
# Usage: gpurun --timeout 1500 -- 'bash tools/gpu_check.sh [tag]'
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo" ; rocminfo 2>/dev/null | grep -E 'Marketing|gfx|Compute Unit|L2|L3' | head -12
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.txt
echo "== bench"
timeout 900 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
tail -c 6000 $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprof kernel stats"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o sad -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof.err
find $OUT/prof -name '*kernel_stats*' | head; f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f"
