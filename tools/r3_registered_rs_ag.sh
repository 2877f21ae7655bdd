#!/bin/bash
# Reduce-scatter and all-gather of BYTES (default 256 MiB) per rank with 8 ranks sharing ONE GPU (the reduce-scatter + all-gather benchmark configuration):
# FIFO path against registered buffers (reduce-scatter: send buffers read directly; all-gather: blocks written directly).
set -o pipefail
run() {  # coll algo flag
  local out
  out=$(NCCL_ALGO=$2 timeout -k 10 250 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus 8 --steps 10 --warmup 3 --no-cpu-baseline --coll $1 --bytes ${BYTES:-268435456} $3 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('$1 $2 ${3:-fifo}:', d['ms_per_step'], 'ms  pieces', d['config']['gather_phase_pieces'], 'busbw', d['value'])" "$out"
}
run reducescatter Direct "" && run reducescatter Direct --register && run allgather Direct "" && run allgather Direct --register && run allgather Ring "" && run allgather Ring --register
