#!/bin/bash
# 1 GiB all-reduce with ranks sharing ONE GPU, FIFO path against registered receive buffers (bench.py --register:
# receive buffer from ncclMemAlloc), for the all-pairs schedule, the ring and the chain.  ms per all-reduce.
set -o pipefail
run() {  # ranks algo extra-flag
  local out
  out=$(NCCL_ALGO=$2 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 \
        --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus $1 --steps 10 --warmup 3 --no-cpu-baseline $3 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('ranks $1 $2 ${3:-fifo}:', d['ms_per_step'], 'ms  verified', d['verified_against_oracle'], 'pieces', d['config']['gather_phase_pieces'], 'hbm frac', d['roofline']['frac'])" "$out"
}
for n in 8 2; do for a in Ring Tree Direct; do run $n $a "" && run $n $a --register || exit 1; done; done
