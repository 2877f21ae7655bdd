#!/bin/bash
# usage: tools/profiles.sh r03   (tag in front of every file name)
# Profile set: N=1 bench alone and under rocprofv3 (kernel trace), PMC passes for its HBM traffic,
# 2-rank loop-back (both ranks on one GPU) under rocprofv3 and with step timers.  Results under gpurun_out/${TAG}/.
set -o pipefail
TAG=${1:-r03}
R=$PWD
O=$R/gpurun_out/${TAG}
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 300 python3 bench.py > $O/${TAG}_bench_n1.json 2> $O/bench_n1.err || { tail $O/bench_n1.err; exit 1; }
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_n1 -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/${TAG}_bench_n1_under_rocprof.json 2> $O/prof_n1.err || { tail $O/prof_n1.err; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/pmc_$c.json 2> $O/pmc_$c.err || { tail $O/pmc_$c.err; exit 1; }
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lb2 -- python3 $R/tools/loopback_inproc.py --ranks 2 --iters 20 > $O/${TAG}_loopback_2ranks_1gpu.json 2> $O/prof_lb2.err || { tail $O/prof_lb2.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_os8m -- python3 $R/tools/loopback_inproc.py --ranks 2 --bytes 8388608 --iters 300 > $O/${TAG}_loopback_2ranks_1gpu_8MiB_one_exchange.json 2> $O/prof_os8m.err || { tail $O/prof_os8m.err; exit 1; }
cd $R
RCCL_AMD_TUNE_FLAGS=32 timeout -k 10 200 python3 tools/loopback_inproc.py --ranks 2 --iters 20 > $O/${TAG}_loopback_2ranks_1gpu_step_timers.json 2> $O/timers.err || exit 1
for n in 2 4 8; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600+n)) bench.py --gpus $n --steps 10 --warmup 3 > $O/${TAG}_bench_loopback_${n}ranks_1gpu.json 2> $O/bench_lb$n.err || { tail -20 $O/bench_lb$n.err; exit 1; }
done
timeout -k 10 300 python3 tools/local_rc_bench.py > $O/${TAG}_local_reduce_copy_dtypes.txt 2> $O/local_rc.err || { tail $O/local_rc.err; exit 1; }
find $O -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head
# keep the merged output small: only the stats and counter files
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
du -sh $O
