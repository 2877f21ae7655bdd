#!/bin/bash
# Where a mid-size one-exchange all-reduce spends its time: step timers and kernel durations (2 ranks, one GPU).
set -o pipefail
mkdir -p gpurun_out
for B in 2097152 8388608; do
  RCCL_AMD_TUNE_FLAGS=32 timeout -k 10 100 python tools/loopback_inproc.py --ranks 2 --bytes $B --iters 200 2>&1 | grep "^{" || exit 1
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/oneshot_prof -o r -- python3 $GRAFT_REPO_ROOT/tools/loopback_inproc.py --ranks 2 --bytes 2097152 --iters 200 > /dev/null 2>&1 || exit 1
find $GRAFT_REPO_ROOT/gpurun_out/oneshot_prof -name "*kernel_stats.csv" | head -1 | xargs head -8
