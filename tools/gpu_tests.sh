#!/bin/bash
# The GPU test suite with its log under gpurun_out/; exits non-zero when a test fails or the GPU runtime reports a fault.
mkdir -p gpurun_out
timeout -k 10 ${2:-1000} python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -rs --durations=12 ${1:+-k "$1"} > gpurun_out/tests.log 2>&1
rc=$?
grep -v "^frame" gpurun_out/tests.log | tail -25
if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/tests.log; then exit 3; fi
exit $rc
