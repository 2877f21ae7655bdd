#!/bin/bash
# 8-rank 1 GiB loop-back all-reduce (8 processes on one GPU) under a few settings.
run() { echo "== $*"; env "$@" timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['config']['plan'])
"; }
run X=1
run X=2
run RCCL_AMD_TUNE_FLAGS=8
run RCCL_AMD_NCHANNELS=10
run RCCL_AMD_NCHANNELS=6
