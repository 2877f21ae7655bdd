#!/bin/bash
# For each earlier test X of the 2-rank bed: run X then the registered test (which prints the probe) in fresh processes.
for X in "allreduce_cases and Direct" "allreduce_cases and Ring" "allreduce_cases and Tree" premul_sum scalar_in_device "reduce_scatter_cases" "all_gather_cases" broadcast_and_reduce managed_memory unaligned_user back_to_back two_communicators low_latency hip_graph host_staged; do
  echo "=== $X"
  timeout -k 10 200 python -u -m pytest tests/test_gpu_collectives.py -m gpu -x -q -s --timeout 150 --timeout-method thread -k "(($X) or registered_receive) and 2 and not missing" 2>&1 | grep -E "PROBE|passed|failed" | cut -c1-700
done
