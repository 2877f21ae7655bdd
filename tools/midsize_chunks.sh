#!/bin/bash
# usage: tools/midsize_chunks.sh RANKS : all-reduce time for 2-32 MiB against the smallest piece a round may move
N=${1:-2}
for B in 2097152 4194304 8388608 16777216 33554432; do
  for MC in 16384 32768 65536 131072; do
    RCCL_AMD_MIN_CHUNK_BYTES=$MC timeout -k 10 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29971 \
      bench.py --gpus $N --steps 200 --warmup 20 --bytes $B 2>/dev/null | grep "^{" | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['config']['plan']; print('ranks $N bytes $B min_chunk $MC us', round(d['ms_per_step']*1000,1), 'ch', p['nChannels'], 'chunkElems', p['chunkElems'])" || exit 1
  done
done
