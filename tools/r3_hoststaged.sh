#!/bin/bash
# The host-staged benchmark configuration: ncclAllReduce sum fp32, 256 MiB, 8 ranks, every hop through pinned host memory
# and loopback TCP (NCCL_P2P_DISABLE=1).  Bench line with the proxy threads' timers and the kernels' step timers.
set -o pipefail
O=gpurun_out
for ch in 4 2; do
  NCCL_P2P_DISABLE=1 NCCL_SHM_DISABLE=1 NCCL_NET=Socket RCCL_AMD_NCHANNELS=$ch RCCL_AMD_TUNE_FLAGS=32 timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port $((29900 + ch)) bench.py --gpus 8 --steps 5 --warmup 2 --bytes 268435456 --no-cpu-baseline 2>$O/hs_$ch.err | tail -1 > $O/r03_bench_hoststaged_tcp_8ranks_1gpu_256MiB_${ch}ch.json || { tail -5 $O/hs_$ch.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('$O/r03_bench_hoststaged_tcp_8ranks_1gpu_256MiB_${ch}ch.json').read())
print('channels $ch:', d['ms_per_step'], 'ms', 'busbw', d['value'], 'verified', d['verified_against_oracle'])
print(' proxy', d['config']['host_staged_proxy_rank0'])
print(' steps', d['config']['step_timers_rank0'])"
done
