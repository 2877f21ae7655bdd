#!/bin/bash
# Does the stride between FIFO slots matter (all workgroups start their steps at slot bases that are multiples of 768 KiB
# apart)?  1 GiB all-reduce with ranks sharing ONE GPU, slot sizes a little above 768 KiB (the steps stay 768 KiB).
set -o pipefail
run() {  # ranks slotbytes
  local out
  out=$(RCCL_AMD_SLOT_BYTES=$2 timeout -k 10 250 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 \
        --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus $1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('ranks $1 slot $2:', d['ms_per_step'], 'ms  verified', d['verified_against_oracle'], d['config']['plan'])" "$out"
}
for n in 8 2; do for s in 786432 790528 802816 819200 851968; do run $n $s || exit 1; done; done
