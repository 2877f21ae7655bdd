#!/bin/bash
# usage: tools/two_exchange_limits.sh RANKS SIZES... : time per all-reduce on the three-phase FIFO schedule and in two exchanges
N=$1; shift
for B in "$@"; do
  for V in 0 134217728; do
    RCCL_AMD_ONESHOT_MAX_BYTES=0 RCCL_AMD_TWO_EXCHANGE_MAX_BYTES=$V timeout -k 10 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29974 \
      bench.py --gpus $N --steps 50 --warmup 5 --bytes $B --no-cpu-baseline 2>/dev/null | grep "^{" | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['config']['plan']; print('ranks $N bytes $B two_exchange_max $V us', round(d['ms_per_step']*1000,1), 'proto', p.get('proto'), 'ch', p['nChannels'], 'ok', d.get('verified_against_oracle'))" || exit 1
  done
done
