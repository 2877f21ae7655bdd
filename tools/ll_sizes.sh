#!/bin/bash
# usage: tools/ll_sizes.sh RANKS [COLL] : time per operation across the low-latency range
N=${1:-2}; COLL=${2:-allreduce}
for B in 1024 65536 131072 262144 524288 1048576; do
  timeout -k 10 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29961 \
    bench.py --gpus $N --steps 300 --warmup 20 --coll $COLL --bytes $B 2>/dev/null | grep "^{" | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ranks $N $COLL bytes/rank $B us/op', round(d['ms_per_step']*1000,2), 'proto', d['config']['plan']['proto'])" || exit 1
done
