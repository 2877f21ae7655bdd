#!/bin/bash
# FIFO slot size against time: all-reduce with 2, 4 and 8 ranks sharing ONE GPU at 1 GiB, and 128 / 256 / 512 MiB with two ranks.
one() { # ranks bytes slot
  RCCL_AMD_SLOT_BYTES=$3 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((29700+$1)) bench.py --gpus $1 --steps 10 --warmup 3 --no-cpu-baseline --bytes $2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('ranks $1 bytes $2 slot $3:', d['ms_per_step'], 'ms, frac', d['roofline']['frac'], d['config'].get('plan'), d.get('verified_against_oracle'))"
}
for slot in ${SLOTS:-524288 786432}; do
  one 2 1073741824 $slot
  one 8 1073741824 $slot
  one 4 1073741824 $slot
  one 2 134217728 $slot
  one 2 268435456 $slot
  one 2 536870912 $slot
  one 8 268435456 $slot
done
