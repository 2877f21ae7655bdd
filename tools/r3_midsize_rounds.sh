#!/bin/bash
# All-pairs FIFO schedule just above the two-exchange limit, ranks sharing ONE GPU: rounds per channel against time.
set -o pipefail
run() {  # ranks bytes rounds
  local out
  out=$(RCCL_AMD_ROUNDS_PER_CHANNEL=$3 timeout -k 10 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 \
        --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus $1 --steps 30 --warmup 5 --no-cpu-baseline --bytes $2 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('ranks $1 bytes $2 rounds $3:', round(d['ms_per_step']*1000,1), 'us  plan', d['config'].get('plan'))" "$out"
}
for n in 8 4; do for b in 134217728 268435456; do for r in 0 6 8 12 16 24; do run $n $b $r || exit 1; done; done; done
