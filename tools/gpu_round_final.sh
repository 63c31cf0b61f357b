# final measurement set of the tree (tag = the round, e.g. r06z): GPU test suite, per-workload rocprofv3 kernel stats (tools/gpu_profiles.sh), the default bench line
set -u; export TMPDIR=/tmp
T=${1:-r06z}; OUT=gpurun_out/$T; mkdir -p $OUT
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/pytest_gpu.log; cat $OUT/pytest_gpu.log
bash tools/gpu_profiles.sh $T > $OUT/profiles.log 2>&1
find $OUT/stats -name '*kernel_trace.csv' -delete; find $OUT/stats -name '*agent_info.csv' -delete; find $OUT/stats -name '*domain_stats.csv' -delete
grep -E "^== |bench:" $OUT/profiles.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; wc -c $OUT/bench_default.json; cp gpurun_out/bench_full.json $OUT/bench_full.json
du -sh $OUT
bash tools/bd_audit.sh $OUT/bd_audit > $OUT/bd_audit.txt 2>&1; grep -c "8-BIT SLOWER" $OUT/bd_audit.txt
