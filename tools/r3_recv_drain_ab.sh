#!/bin/bash
# Receive-only steps with their drain against without it (one of the two is RCCL_AMD_TUNE_FLAGS=128; see Ctx::step): 1 GiB all-reduce at 2 / 4 / 8 ranks
# and reduce-scatter / all-gather of 128 MiB per rank at 8 ranks, ranks sharing ONE GPU.  ms per call.
set -o pipefail
run() {  # ranks flags extra
  local out
  out=$(RCCL_AMD_TUNE_FLAGS=$2 timeout -k 10 250 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 \
        --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus $1 --steps 10 --warmup 3 --no-cpu-baseline $3 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('ranks $1 flags $2 ${3:-allreduce 1 GiB}:', d['ms_per_step'], 'ms  verified', d['verified_against_oracle'])" "$out"
}
for i in 1 2; do
  for n in 2 4 8; do run $n 128 && run $n 0 || exit 1; done
  run 8 128 "--coll reducescatter --bytes 134217728" && run 8 0 "--coll reducescatter --bytes 134217728" || exit 1
  run 8 128 "--coll allgather --bytes 134217728" && run 8 0 "--coll allgather --bytes 134217728" || exit 1
done
