#!/bin/bash
# usage: tools/ll_threshold.sh <ranks> : all-reduce time around the low-latency / FIFO switch-over, both protocols
N=${1:-2}
for B in 131072 262144 393216 524288 786432 1048576; do
  for MODE in LL Simple; do
    if [ $MODE = LL ]; then E="RCCL_AMD_LL_MAX_BYTES=1048576"; else E="NCCL_PROTO=Simple"; fi
    env $E timeout -k 10 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
      --master-port 29517 bench.py --gpus $N --steps 300 --warmup 20 --bytes $B 2>/dev/null | grep "^{" | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ranks $N allreduce bytes $B $MODE us/op', round(d['ms_per_step']*1000,2), 'proto', d['config']['plan']['proto'])" || exit 1
  done
done
