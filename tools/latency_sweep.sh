#!/bin/bash
# usage: tools/latency_sweep.sh <ranks> : small-message time per operation, low-latency protocol against the FIFO protocol
N=${1:-2}
for COLL in ${COLLS:-allreduce reducescatter allgather broadcast reduce}; do
  for B in 1024 65536 262144; do
    for PROTO in LL Simple; do
      NCCL_PROTO=$PROTO timeout -k 10 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
        --master-port 29517 bench.py --gpus $N --steps 300 --warmup 20 --coll $COLL --bytes $B 2>/dev/null | grep "^{" | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ranks $N $COLL bytes/rank $B NCCL_PROTO=$PROTO us/op', round(d['ms_per_step']*1000,2), 'proto', d['config']['plan']['proto'])" || exit 1
    done
  done
done
