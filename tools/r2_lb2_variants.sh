#!/bin/bash
# 2-rank 1 GiB loop-back all-reduce (both ranks on one GPU) under a few settings; one line each.
run() { echo "== $*"; env "$@" timeout -k 10 120 python tools/loopback_inproc.py --ranks 2 --iters 10 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_allreduce'], d['hbm_GBps'], d['exact'], d['plan'], d.get('step_timers'))
    else: print(l[:300])
"; }
if [ $# -gt 0 ]; then for v in "$@"; do run $(echo $v | tr ',' ' '); done; exit 0; fi
run X=1
run RCCL_AMD_TUNE_FLAGS=8
run RCCL_AMD_WINDOW_MEM=fine
run RCCL_AMD_WINDOW_MEM=coarse
run RCCL_AMD_NCHANNELS=48
run RCCL_AMD_NCHANNELS=56
