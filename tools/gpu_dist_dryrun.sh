#!/bin/bash
# Dry run of bench.py's N > 1 code path on a ONE-GPU box.  RCCL refuses two ranks on one device, so the path runs with a
# process group and an RCCL communicator of ONE rank: torch.distributed set-up, the id broadcast, aomhip_comm_init,
# aomhip_allgather_recon in front of every search step, the reductions, the single-GPU comparison leg; the headline stays the weak-scaling SAD metric.
# (Peer traffic itself: the loop-back case of tests/test_gpu_exchange.py on one GPU, the multi-process case with >= 2.)
mkdir -p gpurun_out/dryrun
AOMHIP_BENCH_FORCE_DIST=1 python bench.py --steps 5 --warmup 1 "$@" 2> gpurun_out/dryrun/forced.err | grep '^{' > gpurun_out/dryrun/forced.json
python - gpurun_out/dryrun/forced.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print({k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "scaling", "ms_per_step", "parity_frame0_and_last_slot")})
assert d["metric"] == "SAD-candidates/s" and d["scaling"] == "weak"  # the same headline at every N
s = d["strong_scaling_search"]
print("strong_scaling_search:", {k: s.get(k) for k in ("value", "unit", "speedup_over_single_gpu", "rccl_ranks_in_communicator", "parity_sample_slot0", "exchange", "tile_columns_px", "blocks_max_rank_over_mean")})
assert s["rccl_ranks_in_communicator"] == d["n_gpus"]
PY
tail -5 gpurun_out/dryrun/forced.err
