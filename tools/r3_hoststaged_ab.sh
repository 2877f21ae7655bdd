#!/bin/bash
# Host-staged all-reduce, 256 MiB, 8 ranks, 4 channels: stamps against a full walk of the flag lines in every pass; step timers off.
set -o pipefail
run() {
  local out
  out=$(env "$@" NCCL_P2P_DISABLE=1 NCCL_SHM_DISABLE=1 RCCL_AMD_NCHANNELS=4 timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port $((29900 + RANDOM % 90)) bench.py --gpus 8 --steps 8 --warmup 2 --bytes 268435456 --no-cpu-baseline 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('$*:', d['ms_per_step'], 'ms', {k: round(v) for k, v in d['config']['host_staged_proxy_rank0'].items() if k.endswith('_ms')})" "$out"
}
for i in 1 2; do run RCCL_AMD_NET_IDLE_SLEEP_US=0 && run RCCL_AMD_NET_IDLE_SLEEP_US=20 && run RCCL_AMD_NET_IDLE_SLEEP_US=100 && run RCCL_AMD_NET_FULL_SCAN=1 || exit 1; done
