#!/bin/bash
# Measurements that go with the profile set: per-role step timers of the 1 GiB all-reduce at 4 and 8 ranks on one GPU,
# 3 ranks for the odd case, then the registered-buffer comparisons.  Results under gpurun_out/r03x/.
set -o pipefail
O=gpurun_out/r03x
mkdir -p $O
for n in 4 8; do
  RCCL_AMD_TUNE_FLAGS=32 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29650+n)) bench.py --gpus $n --steps 10 --warmup 3 --no-cpu-baseline > $O/r03_bench_loopback_${n}ranks_1gpu_step_timers.json 2> $O/timers$n.err || { tail -20 $O/timers$n.err; exit 1; }
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29659 bench.py --gpus 3 --steps 10 --warmup 3 --no-cpu-baseline > $O/r03_bench_loopback_3ranks_1gpu.json 2> $O/lb3.err || { tail -20 $O/lb3.err; exit 1; }
bash tools/r3_registered.sh > $O/r3_registered.txt 2>&1 || { tail $O/r3_registered.txt; exit 1; }
bash tools/r3_registered_rs_ag.sh > $O/r3_registered_rs_ag.txt 2>&1 || { tail $O/r3_registered_rs_ag.txt; exit 1; }
BYTES=134217728 bash tools/r3_registered_rs_ag.sh > $O/r3_registered_rs_ag_128MiB.txt 2>&1 || { tail $O/r3_registered_rs_ag_128MiB.txt; exit 1; }
cat $O/*.txt
