#!/bin/bash
# usage: tools/loop_tests.sh N "<pytest -k expression or empty>" [tag]   -- repeat a GPU test selection, count failures
N=$1; K=$2; TAG=${3:-loop}
fails=0
for i in $(seq 1 $N); do
  if [ -n "$K" ]; then
    timeout -k 10 300 python -u -m pytest tests -x -q -m gpu -k "$K" --timeout 200 --timeout-method thread > gpurun_out/${TAG}_$i.log 2>&1
  else
    timeout -k 10 300 python -u -m pytest tests -x -q -m gpu --timeout 200 --timeout-method thread > gpurun_out/${TAG}_$i.log 2>&1
  fi
  e=$?
  if [ $e -ne 0 ]; then fails=$((fails+1)); grep -h "^E  *\(Assert\|Runtime\)\|^FAILED" gpurun_out/${TAG}_$i.log | cut -c1-300; fi
  if [ $e -ge 124 ]; then echo "stopping: exit $e"; break; fi
done
echo "$TAG: $fails failures of $N ($K)"
grep -h "read back differently" gpurun_out/${TAG}_*.log | sort | uniq -c | head -5
