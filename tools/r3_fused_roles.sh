#!/bin/bash
# 1 GiB all-reduce on the all-pairs schedule, ranks sharing ONE GPU: three role workgroups per channel against one workgroup
# per channel that walks all three roles (RCCL_AMD_FUSED_ROLES=1), at channel counts that fill the GPU either way.
set -o pipefail
run() {  # ranks channels fused
  local out
  out=$(RCCL_AMD_FUSED_ROLES=$3 RCCL_AMD_NCHANNELS=$2 timeout -k 10 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 \
        --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus $1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); v=d.get('verification') or {}; print('ranks $1 channels $2 fused $3:', d['ms_per_step'], 'ms  hbm frac', d['roofline']['frac'], 'mismatches', v.get('whole_buffer_mismatches_all_ranks'))" "$out"
}
run 8 10 0 && run 8 32 1 && run 8 24 1 && run 4 21 0 && run 4 64 1 && run 4 48 1 && run 3 26 0 && run 3 80 1
