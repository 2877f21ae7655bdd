#!/bin/bash
# Reduce-scatter and all-gather of 256 MiB per rank, 8 ranks sharing ONE GPU, against the channel count (two workgroups per
# channel and rank: 10 channels = 160 workgroups in flight, 16 = 256), with the per-role step timers of rank 0.
set -o pipefail
run() {  # coll channels
  local out
  out=$(RCCL_AMD_TUNE_FLAGS=32 RCCL_AMD_NCHANNELS=$2 timeout -k 10 250 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus 8 --steps 10 --warmup 3 --no-cpu-baseline --coll $1 --bytes 268435456 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('$1 channels $2:', d['ms_per_step'], 'ms  busbw', d['value'], 'frac', d['roofline']['frac'], d['config'].get('step_timers_rank0'))" "$out"
}
for c in 10 12 14 16; do run reducescatter $c && run allgather $c || exit 1; done
