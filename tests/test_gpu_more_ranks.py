"""Rank counts 5, 7 and 12 (the reference sweeps every count from UT_MIN_GPUS to UT_MAX_GPUS, test/common/TestBed.cpp:614-765).
A module of its own: its 12-rank bed must not be alive together with the beds of other modules (a GPU box lets 16 processes
use the card at once)."""
import numpy as np
import pytest

import expected as ex
import oracle as orc
from test_gpu_fullsize import ODD_ENVS, _check_allreduce, _devices, _es
from testbed import COLL_ALLGATHER, COLL_ALLREDUCE, COLL_REDUCESCATTER, TestBed

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[5, 7, 12])
def more_bed(request):
    tb = TestBed(request.param, devices=_devices())
    yield tb
    tb.close()


@pytest.mark.parametrize("mode", ["default", "Ring", "Tree", "Ring+LL"])
def test_more_rank_counts(more_bed, mode):
    """A prime rank count on both sides of the two-exchange split, and one above eight ranks, where the all-pairs schedule
    hands over to the ring by itself: averages, sums, an integer product, reduce-scatter and all-gather, ragged counts."""
    bed = more_bed
    n = bed.nranks
    bed.init_comms(ODD_ENVS[mode])
    for dt, op, count in ((orc.F32, orc.AVG, 1250), (orc.F16, orc.AVG, 70001), (orc.F32, orc.SUM, (1 << 20) + 3),
                          (orc.BF16, orc.AVG, 300007), (orc.I32, orc.AVG, 70001), (orc.I64, orc.PROD, 5001)):
        _check_allreduce(bed, dt, op, count)
    _check_allreduce(bed, orc.F32, orc.SUM, 262144 + 5, in_place=True)
    x = np.arange(1250, dtype=np.int32)
    for o in bed.run(COLL_ALLREDUCE, orc.I32, orc.SUM, 1250, [x] * n, 1250 * 4):
        assert np.array_equal(o.view(np.int32), n * x)
    for dt, op, count in ((orc.F32, orc.AVG, 1000), (orc.I32, orc.SUM, 262144 + 7)):
        ins = [orc.fill_hashed(dt, count * n, r, 0x0DD) for r in range(n)]
        outs = bed.run(COLL_REDUCESCATTER, dt, op, count, ins, count * _es(dt))
        plan = bed.plan(COLL_REDUCESCATTER, count, dt)
        for r in range(n):
            assert ex.check(dt, outs[r], ex.reduce_scatter(dt, op, ins, r, plan), exact=True) is None, (orc.DT_NAMES[dt], count, r, plan)
    for dt, count in ((orc.F16, 1001), (orc.U8, 1039203)):
        ins = [orc.fill_hashed(dt, count, r, 0xA6) for r in range(n)]
        outs = bed.run(COLL_ALLGATHER, dt, orc.SUM, count, ins, count * n * _es(dt))
        for r in range(n):
            assert ex.check(dt, outs[r], orc.all_gather_expected(ins), exact=True) is None, (orc.DT_NAMES[dt], count, r)
    bed.init_comms({"NCCL_ALGO": None, "NCCL_PROTO": None})
