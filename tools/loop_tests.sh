#!/bin/bash
# usage: tools/loop_tests.sh N "<pytest -k expression>" [tag]   -- repeat a GPU test selection, count failures
N=$1; K=$2; TAG=${3:-loop}
fails=0
for i in $(seq 1 $N); do
  timeout -k 10 120 python -u -m pytest tests/test_gpu_collectives.py -x -q -m gpu -k "$K" --timeout 100 --timeout-method thread > gpurun_out/${TAG}_$i.log 2>&1
  e=$?
  if [ $e -ne 0 ]; then fails=$((fails+1)); grep -h "^E  *Assert" gpurun_out/${TAG}_$i.log | cut -c1-200; fi
  if [ $e -ge 124 ]; then echo "stopping: exit $e"; break; fi
done
echo "$TAG: $fails failures of $N ($K)"
grep -h "PRE-CHECK\|POST-CHECK" gpurun_out/${TAG}_*.log | sort | uniq -c | cut -c1-400 | head -20
