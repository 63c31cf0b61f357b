set -u; TAG=r06t; export TMPDIR=/tmp; mkdir -p gpurun_out/$TAG
bash tools/gpu_pmc_txq.sh $TAG txq_1080p_8bit > gpurun_out/$TAG/txq_1080p.log 2>&1
bash tools/gpu_pmc_txq.sh $TAG txq_4k_10bit > gpurun_out/$TAG/txq_4k.log 2>&1
for WL in variance16x16_modeA_1080p_8bit variance16x16_modeA_4k_10bit sub_pixel_variance16x16_modeA_1080p_8bit sub_pixel_variance16x16_modeA_4k_10bit filters_ring_4k_10bit; do
  OUT=gpurun_out/$TAG/pmc_$WL; mkdir -p $OUT
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o pmc -- python3 bench.py --steps 3 --warmup 1 --workload $WL --others "" --no-cpu-baseline > $OUT/$C.json 2> $OUT/$C.err
  done
  python3 tools/pmc_traffic_generic.py $TAG $WL
done > gpurun_out/$TAG/generic.log 2>&1
cp profiles/traffic.json gpurun_out/$TAG/traffic.json
cp profiles/${TAG}_pmc_*.json gpurun_out/$TAG/ 2>/dev/null
find gpurun_out/$TAG -name '*kernel_trace.csv' -delete; find gpurun_out/$TAG -name '*agent_info.csv' -delete
for f in txq_1080p txq_4k generic; do tail -n 2 gpurun_out/$TAG/$f.log | cut -c1-300; done
