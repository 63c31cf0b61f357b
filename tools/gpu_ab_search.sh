mkdir -p gpurun_out/r01u
python -m pytest tests/test_gpu_mcomp.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2; do
for v in "" aom-av1-psy_amd/lib/libaomhip_nolds.so; do
echo "== lib=$v"
AOMHIP_LIB=$v python bench.py --workload search_4k_10bit --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
done; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01u/search -o k -- python bench.py --workload search_4k_10bit --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r01u/search.json 2>gpurun_out/r01u/search.err
cut -c1-150 gpurun_out/r01u/search/*kernel_stats.csv | head -5
