#!/bin/bash
# The benchmark configurations with every rank on ONE GPU (loop-back: HBM-bound, no link is used):
#   ncclAllReduce sum fp32, 1 MiB .. 1 GiB, 2 ranks, ring and default schedule
#   ncclReduceScatter + ncclAllGather fp32, 256 MiB per rank, 8 ranks, ring and default schedule
#   ncclAllReduce sum bf16 / fp16, 1 GiB, 8 ranks, tree (chain) and default schedule
# One line per run: ms per operation, bus bandwidth, fraction of HBM peak on the design's own bytes, verification.
set -o pipefail
run() {  # ranks algo coll dtype bytes
  local out
  out=$(NCCL_ALGO=$2 timeout -k 10 250 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 \
        --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus $1 --steps 20 --warmup 5 --no-cpu-baseline --coll $3 --dtype $4 --bytes $5 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); v=d.get('verification') or {}; print('ranks $1 ${2:-default} $3 $4 $5 B:', d['ms_per_step'], 'ms  busbw', d['value'], 'GB/s  hbm frac', d['roofline']['frac'], ' verified', d.get('verified_against_oracle'), 'mismatches', v.get('whole_buffer_mismatches_all_ranks'), 'plan', d['config'].get('plan'))" "$out"
}
for b in 1048576 4194304 16777216 67108864 268435456 1073741824; do run 2 Ring allreduce fp32 $b && run 2 "" allreduce fp32 $b || exit 1; done
for a in Ring ""; do run 8 "$a" reducescatter fp32 268435456 && run 8 "$a" allgather fp32 268435456 || exit 1; done
for t in bf16 fp16; do run 8 Tree allreduce $t 1073741824 && run 8 "" allreduce $t 1073741824 || exit 1; done
