#!/bin/bash
# What one GPU of an 8-GPU node would run, reproduced as total workgroups on ONE GPU (8 ranks sharing it): 168 workgroups
# for 56 channels x 3 roles, against 56 / 112 / 168 workgroups of the one-workgroup-per-channel form.  1 GiB all-reduce.
set -o pipefail
run() {  # channels fused
  local out
  out=$(RCCL_AMD_FUSED_ROLES=$2 RCCL_AMD_NCHANNELS=$1 timeout -k 10 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus 8 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); w=8*$1*(1 if $2 else 3); print('channels per rank $1 fused $2 workgroups on the GPU', w, ':', d['ms_per_step'], 'ms ', d['roofline']['achieved'], 'GB/s of HBM traffic')" "$out"
}
run 7 0 && run 7 1 && run 14 1 && run 21 1 && run 28 1
