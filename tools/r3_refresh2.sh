#!/bin/bash
# Loop-back comparisons with bench.py's channel cap derived from the workgroups per channel of the schedule in use.
set -o pipefail
O=gpurun_out/r03y
mkdir -p $O
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29654 bench.py --gpus 4 --steps 10 --warmup 3 > $O/r03_bench_loopback_4ranks_1gpu.json 2> $O/lb4.err || { tail -20 $O/lb4.err; exit 1; }
bash tools/r3_registered.sh > $O/r3_registered.txt 2>&1 || { tail $O/r3_registered.txt; exit 1; }
bash tools/r3_registered_rs_ag.sh > $O/r3_registered_rs_ag.txt 2>&1 || { tail $O/r3_registered_rs_ag.txt; exit 1; }
BYTES=134217728 bash tools/r3_registered_rs_ag.sh > $O/r3_registered_rs_ag_128MiB.txt 2>&1 || { tail $O/r3_registered_rs_ag_128MiB.txt; exit 1; }
tail -c 400 $O/r03_bench_loopback_4ranks_1gpu.json; cat $O/*.txt
