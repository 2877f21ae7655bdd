#!/bin/bash
# 1 GiB all-reduce with ranks sharing ONE GPU against the number of workgroups in flight (3 per channel and rank):
# does the time follow the workgroup count (each compute unit moves what it can) or stay (HBM is the bound)?
set -o pipefail
run() {  # ranks channels
  local out
  out=$(RCCL_AMD_NCHANNELS=$2 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 \
        --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus $1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('ranks $1 channels $2 workgroups', $1*$2*3, ':', d['ms_per_step'], 'ms  verified', d['verified_against_oracle'])" "$out"
}
run 8 4 && run 8 6 && run 8 8 && run 8 10 && run 4 10 && run 4 16 && run 4 20 && run 4 21
