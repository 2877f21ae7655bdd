#!/bin/bash
# Full GPU pass: tests, N=1 bench, loop-back bench lines at 2 and 8 ranks (ranks sharing the one GPU), step timers.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -rs > gpurun_out/r2_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r2_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > gpurun_out/r2_bench_n1.json 2> gpurun_out/r2_bench_n1.err || exit 1
for n in 2 8; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --steps 10 --warmup 3 > gpurun_out/r2_bench_lb$n.json 2> gpurun_out/r2_bench_lb$n.err || { tail -20 gpurun_out/r2_bench_lb$n.err; exit 1; }
done
RCCL_AMD_TUNE_FLAGS=32 timeout -k 10 200 python tools/loopback_inproc.py --ranks 2 --iters 10 > gpurun_out/r2_lb2_timers.json 2>&1 || exit 1
cat gpurun_out/r2_bench_n1.json gpurun_out/r2_bench_lb2.json gpurun_out/r2_bench_lb8.json gpurun_out/r2_lb2_timers.json
