#!/bin/bash
# usage: tools/midsize_rounds.sh RANKS [SIZES...] : all-reduce time against the rounds a channel cuts its share into
N=${1:-2}; shift
SIZES=${@:-2097152 4194304 8388608 16777216 33554432 67108864 134217728 268435456}
for B in $SIZES; do
  for R in ${ROUNDS:-1 2 4 0}; do
    RCCL_AMD_ROUNDS_PER_CHANNEL=$R timeout -k 10 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29972 \
      bench.py --gpus $N --steps 100 --warmup 10 --bytes $B 2>/dev/null | grep "^{" | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['config']['plan']; print('ranks $N bytes $B rounds $R us', round(d['ms_per_step']*1000,1), 'ch', p['nChannels'], 'chunkElems', p['chunkElems'])" || exit 1
  done
done
