"""Communicators at the geometry of a real 8-GPU node with the 8 ranks sharing the box's GPU(s): 56 channels, i.e. a
1464 MiB window per rank (the reference's per-connection buffers, src/transport/p2p.cc:447-450), every rank mapping the
seven others through HIP IPC.  The collectives that follow are the ones that need few enough workgroups to be resident
together on one GPU: the line protocol, one and two exchanges, and the FIFO schedules (all-pairs, ring, chain) at payloads
that occupy 8 of the 56 channels.  The full 168-workgroup launch itself cannot run on one GPU; its step protocol is what
tests/test_schedule_sim.py executes on the CPU at this geometry."""
import numpy as np
import pytest

import expected as ex
import oracle as orc
from testbed import COLL_ALLGATHER, COLL_ALLREDUCE, COLL_REDUCESCATTER, TestBed

pytestmark = pytest.mark.gpu


def _devices():
    import torch
    return tuple(range(torch.cuda.device_count()))


@pytest.fixture(scope="module")
def node_bed():
    tb = TestBed(8, devices=_devices(), env={"RCCL_AMD_NCHANNELS": "56"})
    yield tb
    tb.close()


def _allreduce(bed, count, dt=orc.F32, op=orc.SUM):
    n = bed.nranks
    ins = [orc.fill_hashed(dt, count, r, 0x56 + count) for r in range(n)]
    plan = bed.plan(COLL_ALLREDUCE, count, dt)
    outs = bed.run(COLL_ALLREDUCE, dt, op, count, ins, count * np.dtype(orc.NP_OF[dt]).itemsize)
    want = ex.allreduce(dt, op, ins, plan)
    for r in range(n):
        assert ex.check(dt, outs[r], want, exact=True) is None, (count, r, plan)
    return plan


def test_window_of_the_default_eight_rank_layout_is_allocated_and_mapped(node_bed):
    bed = node_bed
    bed.init_comms({"NCCL_ALGO": None, "NCCL_PROTO": None, "RCCL_AMD_ONESHOT_MAX_BYTES": None, "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": None})
    protos = [_allreduce(bed, c)["proto"] for c in (1250, 65536, 262144 + 3, (1 << 20) + 5, (4 << 20) + 7)]
    assert protos == [1, 1, 2, 3, 3], protos               # lines, lines, one exchange, two exchanges
    _allreduce(bed, 300007, orc.BF16, orc.AVG)


@pytest.mark.parametrize("algo", ["Direct", "Ring", "Tree"])
def test_fifo_schedules_inside_the_default_eight_rank_layout(node_bed, algo):
    """Payloads that give 8 channels of the 56 (8 x 3 x 8 = 192 workgroups at most): slot addresses, flag lines and step
    counters of the full-size layout."""
    bed = node_bed
    bed.init_comms({"NCCL_ALGO": algo, "NCCL_PROTO": "Simple", "RCCL_AMD_ONESHOT_MAX_BYTES": "0", "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": "0"})
    n = bed.nranks
    count = (512 << 10) + 3 if algo != "Tree" else (64 << 10) + 3          # all-pairs and ring cut the payload by ranks first
    plan = _allreduce(bed, count)
    assert plan["proto"] == 0 and 1 <= plan["nChannels"] <= 9, plan
    plan = _allreduce(bed, count, orc.F16, orc.AVG)
    if algo != "Tree":
        c = (64 << 10) + 5
        ins = [orc.fill_hashed(orc.F32, c * n, r, 0x57) for r in range(n)]
        outs = bed.run(COLL_REDUCESCATTER, orc.F32, orc.SUM, c, ins, c * 4)
        plan = bed.plan(COLL_REDUCESCATTER, c, orc.F32)
        assert plan["nChannels"] <= 9, plan
        for r in range(n):
            assert ex.check(orc.F32, outs[r], ex.reduce_scatter(orc.F32, orc.SUM, ins, r, plan), exact=True) is None, (r, plan)
        ins = [orc.fill_hashed(orc.U8, 4 * c, r, 0x58) for r in range(n)]
        outs = bed.run(COLL_ALLGATHER, orc.U8, orc.SUM, 4 * c, ins, 4 * c * n)
        for r in range(n):
            assert ex.check(orc.U8, outs[r], orc.all_gather_expected(ins), exact=True) is None, r
    bed.init_comms({"NCCL_ALGO": None, "NCCL_PROTO": None, "RCCL_AMD_ONESHOT_MAX_BYTES": None, "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": None})
