#!/bin/bash
# 8 ranks sharing ONE GPU, all-reduce of 64 / 128 / 256 MiB: the default (two exchanges up to 64 MiB, then the three-role FIFO
# schedule) against the FIFO schedule with one workgroup per channel (RCCL_AMD_FUSED_ROLES=1, two exchanges off).
set -o pipefail
run() {  # bytes fused twoexmax channels
  local out
  out=$(RCCL_AMD_FUSED_ROLES=$2 RCCL_AMD_TWO_EXCHANGE_MAX_BYTES=$3 RCCL_AMD_NCHANNELS=$4 timeout -k 10 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus 8 --steps 30 --warmup 5 --no-cpu-baseline --bytes $1 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('bytes $1 one-workgroup $2 two-exchange limit $3:', round(d['ms_per_step']*1000,1), 'us  plan', d['config'].get('plan'))" "$out"
}
for b in 67108864 134217728 268435456; do run $b 0 67108864 10 && run $b 1 0 32 || exit 1; done
