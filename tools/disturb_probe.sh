#!/bin/bash
# usage: tools/disturb_probe.sh <tune flags> <tag> : torch fill/copy probe while a 2-rank all-reduce loop runs in another process
F=$1; TAG=$2
RCCL_AMD_TUNE_FLAGS=$F timeout -k 10 150 python -u tools/loopback_inproc.py --ranks 2 --bytes 268435456 --iters ${ITERS:-40000} > gpurun_out/dist_$TAG.log 2>&1 &
D=$!
sleep 8
timeout -k 10 120 python -u tools/torch_fill_copy_probe.py --procs 4 --iters 1500 2>&1 | grep -v amdgpu.ids > gpurun_out/probe_$TAG.log
echo "flags $F: $(grep -c differ gpurun_out/probe_$TAG.log) stale iterations; $(tail -1 gpurun_out/probe_$TAG.log)"
wait $D
echo "disturber exit $?: $(tail -1 gpurun_out/dist_$TAG.log | cut -c1-160)"
