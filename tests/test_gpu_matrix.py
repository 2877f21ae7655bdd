"""Every (element type, operator) instantiation of the reducing kernels through every schedule once.

The kernels are compiled per element type and operator, so a fault of one instantiation (wrong bytes, a device-side
wait that never ends) shows only where that pair meets a given schedule: the reference covers the same ground with
its typed test lists (test/AllReduceTests.cpp:10-130, test/ReduceScatterTests.cpp:10-120).  Sizes of several rounds
per channel; every result bit-exact against the oracle's fold in the schedule's association order.
"""
import importlib.util
import os

import pytest

from testbed import TestBed

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location(
    "matrix_sweep", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "matrix_sweep.py"))
sweep = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(sweep)


@pytest.mark.parametrize("ranks,nbytes", [(2, 3 << 20), (4, 3 << 20), (8, 1 << 20)])
def test_every_instantiation_through_every_schedule(ranks, nbytes):
    import torch
    bed = TestBed(ranks, devices=tuple(range(torch.cuda.device_count())), env={"RCCL_AMD_SPIN_TIMEOUT_MS": "5000"})
    # the case under way goes to a file as it starts: a run that takes the process down still names its case
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    trace = sweep.tracer_to(os.path.join(out, f"matrix_trace_{ranks}.log")) if os.path.isdir(out) and os.access(out, os.W_OK) else None
    try:
        failures = sweep.run(bed, nbytes, trace=trace)
    finally:
        bed.close()
    assert not failures, failures[:5]
