#!/bin/bash
# usage: tools/sizes3.sh RANKS COLL SIZES... : time per operation
N=$1; COLL=$2; shift; shift
for B in "$@"; do
  timeout -k 10 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29973 \
    bench.py --gpus $N --steps 100 --warmup 10 --coll $COLL --bytes $B 2>/dev/null | grep "^{" | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['config']['plan']; print('ranks $N $COLL bytes/rank $B us', round(d['ms_per_step']*1000,1), 'ch', p['nChannels'], 'chunkElems', p['chunkElems'])" || exit 1
done
