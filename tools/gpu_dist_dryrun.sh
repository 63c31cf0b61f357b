#!/bin/bash
# Dry run of bench.py's N > 1 code path on a ONE-GPU box: 2 and 4 ranks share device 0, gloo carries the
# barrier / reductions (RCCL cannot put two ranks on one GPU).  Checks sharding, ring growth, timing reduction.
mkdir -p gpurun_out/dryrun
for N in 2 4 8; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus $N --steps 5 --warmup 1 --dist-backend gloo --frames-per-gpu 8 "$@" 2> gpurun_out/dryrun/n$N.err | grep '^{' > gpurun_out/dryrun/n$N.json
  python - gpurun_out/dryrun/n$N.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print("world", d["n_gpus"], "value %.4g" % d["value"], d["unit"], "parity", d.get("parity_frame0", d.get("parity_sample_slot0")),
      "kernels", d.get("kernels"), "cfg", {k: d["config"].get(k) for k in ("workload", "candidates_per_step")})
PY
done
