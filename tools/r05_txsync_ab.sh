# A/B on one box: transform kernels with wavefront-level LDS ordering (product) vs workgroup barriers (explib/libaomhip_txsync0.so)
set -u; OUT=gpurun_out/r05g; mkdir -p $OUT
python -m pytest tests/test_gpu_xform_quant.py tests/test_gpu_inv_txfm.py tests/test_gpu_encode_block.py tests/test_gpu_full_size.py tests/test_gpu_goldens.py -x -q 2>&1 | tail -3
for R in 1 2; do
for L in "" explib/libaomhip_txsync0.so; do
  for WL in txq_1080p_8bit txq_4k_10bit; do
    AOMHIP_LIB=$L python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().splitlines()[-1]);print('lib=${L:-product}', '$WL', 'value %.4g' % d['value'], {k:round(v['frac'],3) for k,v in d['per_size'].items()})"
  done
  AOMHIP_LIB=$L python bench.py --workload inner_loop_4k_10bit --steps 60 --warmup 3 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().splitlines()[-1]);print('lib=${L:-product}', 'inner_loop', round(d['value'],1), {k:[round(x*1e3,1) for x in v['ms_by_slot']] for k,v in d['stages'].items() if 'xform' in k or 'inv' in k or 'encode' in k})"
done; done
