#!/bin/bash
# All-reduce time against size with the library's default FIFO layout and chunk rule (8 ranks sharing one GPU);
# set RCCL_AMD_NSLOTS / RCCL_AMD_SLOT_BYTES / RCCL_AMD_ROUNDS_PER_CHANNEL in the environment to compare.
for B in 134217728 268435456 1073741824; do
  timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 8 --steps 8 --warmup 2 --bytes $B 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('8 ranks bytes $B:', d['ms_per_step'], d['config']['plan'])" || exit 1
done
