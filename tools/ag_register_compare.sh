#!/bin/bash
# All-gather through the FIFOs vs into registered receive buffers, ranks sharing one GPU (loop-back).
# usage: tools/ag_register_compare.sh RANKS BYTES_PER_RANK...
n=$1; shift
port=29610
for b in "$@"; do
  for reg in "" "--register"; do
    port=$((port + 1))
    timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus $n --coll allgather --bytes $b --steps 10 --warmup 3 --no-cpu-baseline $reg 2>/dev/null |
      python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ranks',d['n_gpus'],'bytes/rank',$b,'registered' if '$reg' else 'fifo      ','ms',d['ms_per_step'],'busbw',d['value'],d['config']['gather_phase_pieces'])" || exit 1
  done
done
