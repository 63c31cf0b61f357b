#!/bin/bash
# Dry run of bench.py's N > 1 code path on a ONE-GPU box: 2 and 4 ranks share device 0, gloo carries the
# barrier / reductions (RCCL cannot put two ranks on one GPU).  Checks sharding, ring growth, timing reduction.
for N in 2 4; do
  echo "== world $N (gloo, shared GPU)"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus $N --steps 5 --warmup 1 --dist-backend gloo --frames-per-gpu 8 2>&1 | tail -3 | cut -c1-900
done
