#!/bin/bash
# round 5: the configs[4] frame after the DPP minimum in fullpel_diamond_kernel and with the ring's frames in one graph
mkdir -p gpurun_out/r05f
timeout 900 python -m pytest tests/test_gpu_mcomp.py tests/test_gpu_bench_schema.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for i in 1 2; do
timeout 600 python bench.py --workload inner_loop_4k_10bit --steps 60 --warmup 5 > gpurun_out/r05f/bench_inner$i.json 2> gpurun_out/r05f/bench_inner$i.err
python3 -c "
import json; d=json.load(open('gpurun_out/r05f/bench_inner$i.json')); print('inner loop', d.get('value'), d.get('ms_per_step')); print(json.dumps(d.get('config'))[:600]); print(json.dumps(d.get('without_graph', d.get('launch')))[:600])"
done
AOMHIP_BENCH_GRAPH=frame timeout 600 python bench.py --workload inner_loop_4k_10bit --steps 60 --warmup 5 > gpurun_out/r05f/bench_inner_frame.json 2> gpurun_out/r05f/bench_inner_frame.err
python3 -c "
import json; d=json.load(open('gpurun_out/r05f/bench_inner_frame.json')); print('inner loop (graph per frame)', d.get('value'), d.get('ms_per_step'))"
