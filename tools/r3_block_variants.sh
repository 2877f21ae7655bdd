#!/bin/bash
# 1 GiB all-reduce with ranks sharing ONE GPU: 1024-thread workgroups against 512-thread ones on twice the channels
# (two workgroups per compute unit: one drains its stores while the other issues).  ms per all-reduce.
set -o pipefail
run() {  # ranks threads channels
  local out
  out=$(RCCL_AMD_NTHREADS=$2 RCCL_AMD_NCHANNELS=$3 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 \
        --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus $1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('ranks $1 threads $2 channels $3:', d['ms_per_step'], 'ms  verified', d['verified_against_oracle'], d['config']['plan'])" "$out"
}
run 4 1024 20 && run 4 512 40 && run 4 512 20 && run 8 1024 10 && run 8 512 20 && run 2 1024 64 && run 2 512 64
