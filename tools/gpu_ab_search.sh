#!/bin/bash
# A/B of motion-search kernel builds: gpurun -- 'bash tools/gpu_ab_search.sh "lib1.so lib2.so"'  ("" = the shipped library)
LIBS=${1:-""}
python -m pytest tests/test_gpu_mcomp.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2; do
for v in "" $LIBS; do
echo "== lib=${v:-shipped}"
AOMHIP_LIB=$v python bench.py --workload search_4k_10bit --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/frame %.4f  blocks/s %.4g parity %s' % (d['ms_per_step'], d['value'], d.get('parity_sample_slot0')))"
done; done
