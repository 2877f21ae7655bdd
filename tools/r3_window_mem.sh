#!/bin/bash
# 1 GiB all-reduce with ranks sharing ONE GPU, by the kind of memory the window is made of (RCCL_AMD_WINDOW_MEM).
set -o pipefail
run() {  # ranks kind
  local out
  out=$(RCCL_AMD_WINDOW_MEM=$2 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 \
        --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus $1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('ranks $1 window $2:', d['ms_per_step'], 'ms  verified', d['verified_against_oracle'])" "$out"
}
for n in 2 8; do for k in uncached finegrained coarse; do run $n $k || exit 1; done; done
