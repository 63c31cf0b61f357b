#!/bin/bash
# A/B of kernel variants selected by env var. Usage: gpurun -- 'bash tools/gpu_ab.sh VAR "v0 v1 ..." [workload]'
VAR=$1; VALS=$2; WL=${3:-sad16x16_modeA_4k_8bit}
mkdir -p gpurun_out/ab
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for round in 1 2; do
for v in $VALS; do
  echo "== $VAR=$v round $round"
  env $VAR=$v python bench.py --steps 20 --warmup 3 --workload $WL --others "" --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.4g cand/s  ms/step %.4f  x4d_ms %.4f cand_ms %.4f parity %s' % (d['value'], d['ms_per_step'], d['kernels']['sad_x4d_kernel_avg_ms'], d['kernels']['sad_cand_kernel_avg_ms'], d['parity_frame0']))"
done
done
