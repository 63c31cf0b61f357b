#!/bin/bash
# round 5: the whole GPU suite + the default bench of the final tree
mkdir -p gpurun_out/r05i
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r05i/pytest_gpu.log; cat gpurun_out/r05i/pytest_gpu.log
python bench.py > gpurun_out/r05i/bench_default.json 2> gpurun_out/r05i/bench_default.err; wc -c gpurun_out/r05i/bench_default.json; cp gpurun_out/bench_full.json gpurun_out/r05i/bench_full.json
python3 -c "
import json; d=json.load(open('gpurun_out/r05i/bench_default.json')); print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
