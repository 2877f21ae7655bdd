#!/bin/bash
# 1 GiB all-reduce and 128 MiB-per-rank reduce-scatter at 3, 5, 6 and 7 ranks sharing ONE GPU (folds with a source count
# known only at run time).
set -o pipefail
run() {  # ranks coll bytes
  local out
  out=$(timeout -k 10 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 \
        --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) bench.py --gpus $1 --steps 10 --warmup 3 --no-cpu-baseline --coll $2 --bytes $3 2>/dev/null | tail -1) || return 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); v=d.get('verification') or {}; print('ranks $1 $2 $3 B:', d['ms_per_step'], 'ms  hbm frac', d['roofline']['frac'], 'mismatches', v.get('whole_buffer_mismatches_all_ranks'), 'plan', d['config'].get('plan'))" "$out"
}
for n in 3 5 6 7; do echo "ranks $n"; run $n allreduce 1073741824 && run $n reducescatter 134217728 || exit 1; done
