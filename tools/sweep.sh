#!/bin/bash
# Size sweep of one collective through torch.distributed.run (loop-back when ranks share a GPU).
#   tools/sweep.sh <ranks> <coll> <dtype> [extra env assignments...]
N=$1; COLL=$2; DT=$3; shift 3
for B in 1048576 4194304 16777216 67108864 268435456 1073741824; do
  env "$@" timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
      --master-port 29511 bench.py --gpus $N --coll $COLL --dtype $DT --bytes $B --steps 10 --warmup 3 2>/dev/null | grep "^{" |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['config']['plan']; print('ranks',$N,'$COLL','$DT','bytes',$B,'ms',d['ms_per_step'],'busbw',d['value'],'algbw',d['config']['algbw_GBps'],'ch',p.get('nChannels'),'chunk',p.get('chunkElems'),'algo',p.get('algo'))" || exit 1
done
