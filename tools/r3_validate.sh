#!/bin/bash
# Validation beyond the test suite: the dtype x op x schedule matrix at three rank counts, then random collectives
# (plain and registered) at several rank counts.  Each step under its own time limit; stops at the first failure.
set -o pipefail
O=gpurun_out/validate
mkdir -p $O
for n in 2 4 8; do
  echo "matrix $n ranks"
  timeout -k 10 400 python3 -u tools/matrix_sweep.py --ranks $n --bytes 4194304 > $O/matrix_$n.txt 2>&1 || { tail -20 $O/matrix_$n.txt; exit 1; }
  tail -2 $O/matrix_$n.txt
done
for n in 3 4 8; do
  echo "fuzz $n ranks"
  timeout -k 10 400 python3 -u tools/fuzz_collectives.py --ranks $n --cases 400 --seed $((n+100)) > $O/fuzz_$n.txt 2>&1 || { tail -20 $O/fuzz_$n.txt; exit 1; }
  tail -2 $O/fuzz_$n.txt
done
for n in 3 4 8; do
  echo "fuzz registered $n ranks"
  timeout -k 10 400 python3 -u tools/fuzz_registered.py --ranks $n --cases 150 --seed $((n+200)) > $O/fuzzreg_$n.txt 2>&1 || { tail -20 $O/fuzzreg_$n.txt; exit 1; }
  tail -2 $O/fuzzreg_$n.txt
done
