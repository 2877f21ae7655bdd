#!/bin/bash
# Two ranks sharing ONE GPU: ring (one workgroup per channel walks scatter, reduce and gather itself) against the default
# schedule, all-reduce over the mid sizes, and reduce-scatter / all-gather.
set -o pipefail
run() {  # algo coll bytes
  local out
  out=$(NCCL_ALGO=$1 timeout -k 10 250 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus 2 --steps 30 --warmup 5 --no-cpu-baseline --coll $2 --bytes $3 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('${1:-default} $2 $3 B:', d['ms_per_step'], 'ms  plan', d['config'].get('plan'))" "$out"
}
for b in 8388608 16777216 25165824 33554432 50331648 67108864 134217728; do run Ring allreduce $b && run "" allreduce $b || exit 1; done
for b in 16777216 67108864 268435456; do for c in reducescatter allgather; do run Ring $c $b && run "" $c $b || exit 1; done; done
