#!/bin/bash
# 1 GiB all-reduce, ranks sharing ONE GPU: channel counts just below the number that fills all 256 compute units.
set -o pipefail
run() {  # ranks channels
  local out
  out=$(RCCL_AMD_NCHANNELS=$2 timeout -k 10 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 \
        --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus $1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('ranks $1 channels $2 workgroups', $1*$2*3, ':', d['ms_per_step'], 'ms')" "$out"
}
run 3 24 && run 3 26 && run 3 27 && run 3 28 && run 5 15 && run 5 16 && run 5 17 && run 6 13 && run 6 14 && run 7 11 && run 7 12
