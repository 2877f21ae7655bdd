#!/bin/bash
# usage: tools/oneshot_limits.sh RANKS SIZES... : time per all-reduce with the one-exchange protocol off and forced
N=$1; shift
for B in "$@"; do
  for V in 0 67108864; do
    RCCL_AMD_ONESHOT_MAX_BYTES=$V timeout -k 10 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29973 \
      bench.py --gpus $N --steps 100 --warmup 10 --bytes $B --no-cpu-baseline 2>/dev/null | grep "^{" | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['config']['plan']; print('ranks $N bytes $B oneshot_max $V us', round(d['ms_per_step']*1000,1), 'proto', p.get('proto'), 'ch', p['nChannels'], 'ok', d.get('verified_against_oracle'))" || exit 1
  done
done
