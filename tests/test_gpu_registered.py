"""Registered buffers on the ring, the chain and the reduce-scatter (zero-copy phases).  These tests allocate and free
many fine-grained buffers of tens of MiB; they run in a bed (processes) of their own."""
import numpy as np
import pytest

import expected as ex
import oracle as orc
from testbed import COLL_ALLGATHER, COLL_ALLREDUCE, COLL_REDUCESCATTER, TestBed

pytestmark = pytest.mark.gpu


def _devices():
    import torch
    return tuple(range(torch.cuda.device_count()))


@pytest.fixture(scope="module", params=[2, 4, 8])
def bed(request):
    tb = TestBed(request.param, devices=_devices())
    yield tb
    tb.close()


@pytest.mark.parametrize("algo", ["Ring", "Tree"])
def test_registered_ring_and_chain_pass_finished_pieces_from_recvbuff_to_recvbuff(bed, algo):
    """The reference's Direct = 1 primitives (src/device/prims_simple.h:243-283; off for its rings, all_reduce.h:62):
    with registered receive buffers the circulation phase of the ring and the way down the chain write finished pieces
    straight into the next rank's recvbuff, which forwards them from there: no FIFO traffic in that phase
    (fifo_pieces == 0), the same bits as through the FIFOs.  Mixed registration states, in place, nothing registered,
    coarse-grained memory (never written by a peer), and a payload below the size from which the words are exchanged."""
    bed.init_comms({"NCCL_ALGO": algo, "NCCL_PROTO": "Simple"})
    n = bed.nranks
    count = (2 << 20) + 24          # 8 MiB fp32: above the 4 MiB from which the registration words are exchanged
    ins = [orc.fill_pattern(orc.F32, count, r) for r in range(n)]
    plan = bed.plan(COLL_ALLREDUCE, count, orc.F32)
    assert plan["proto"] == 0 and plan["algo"] == (ex.ALGO_RING if algo == "Ring" else ex.ALGO_TREE)
    want = ex.allreduce(orc.F32, orc.SUM, ins, plan)
    senders = range(n)
    for outs, stats in bed.run_registered(orc.F32, orc.SUM, count, ins, runs=2):
        for r in range(n):
            assert ex.check(orc.F32, outs[r], want, exact=True) is None, r
            assert stats[r]["fifo_pieces"] == 0, (r, stats[r])
        assert sum(stats[r]["direct_pieces"] for r in senders) > 0
    # replayed from a hipGraph after the first launch mapped the registrations
    for outs, stats in bed.run_registered(orc.F32, orc.SUM, count, ins, runs=2, graph=True):
        for r in range(n):
            assert ex.check(orc.F32, outs[r], want, exact=True) is None, r
            assert stats[r]["fifo_pieces"] == 0, (r, stats[r])
    # only rank 0 registered: whoever sends finished pieces to rank 0 writes directly, everybody else keeps the FIFO
    # (rank 0 is the root of every chain and receives nothing on the way down: no direct piece there)
    (outs, stats), = bed.run_registered(orc.F32, orc.SUM, count, ins, register_ranks=[0])
    for r in range(n):
        assert ex.check(orc.F32, outs[r], want, exact=True) is None, r
    assert stats[0]["direct_pieces"] == 0
    if algo == "Ring":
        assert sum(s["direct_pieces"] for s in stats) > 0 and (n == 2 or sum(s["fifo_pieces"] for s in stats) > 0), stats
    else:
        assert sum(s["direct_pieces"] for s in stats) == 0, stats
    # all but rank 0 registered (rank 0 forwards out of FIFO slots into registered buffers)
    (outs, stats), = bed.run_registered(orc.F32, orc.SUM, count, ins, register_ranks=list(range(1, n)))
    for r in range(n):
        assert ex.check(orc.F32, outs[r], want, exact=True) is None, r
    assert sum(s["direct_pieces"] for s in stats) > 0, stats
    # in place, ragged bf16 count, average
    c2 = 3 * 1024 * 1024 + 1000
    ins2 = [orc.fill_pattern(orc.BF16, c2, r) for r in range(n)]
    want2 = ex.allreduce(orc.BF16, orc.AVG, ins2, bed.plan(COLL_ALLREDUCE, c2, orc.BF16))
    (outs, stats), = bed.run_registered(orc.BF16, orc.AVG, c2, ins2, in_place=True)
    for r in range(n):
        assert ex.check(orc.BF16, outs[r], want2, exact=True) is None, r
        assert stats[r]["fifo_pieces"] == 0, (r, stats[r])
    # nothing registered, and registered coarse-grained memory: FIFOs only, same bits
    for kw in (dict(register_ranks=[]), dict(fine=False)):
        (outs, stats), = bed.run_registered(orc.F32, orc.SUM, count, ins, **kw)
        for r in range(n):
            assert ex.check(orc.F32, outs[r], want, exact=True) is None, (kw, r)
            assert stats[r]["direct_pieces"] == 0, (kw, r, stats[r])
        assert sum(s["fifo_pieces"] for s in stats) > 0
    # below the threshold no words are exchanged: registered buffers, FIFO path, same bits
    c3 = (512 << 10) + 8
    ins3 = [orc.fill_pattern(orc.F32, c3, r) for r in range(n)]
    want3 = ex.allreduce(orc.F32, orc.SUM, ins3, bed.plan(COLL_ALLREDUCE, c3, orc.F32))
    (outs, stats), = bed.run_registered(orc.F32, orc.SUM, c3, ins3)
    for r in range(n):
        assert ex.check(orc.F32, outs[r], want3, exact=True) is None, r
        assert stats[r]["direct_pieces"] == 0, (r, stats[r])


def test_registered_ring_all_gather_writes_blocks_directly(bed):
    """Ring all-gather into registered receive buffers: every hop writes the block into the next rank's recvbuff, which
    forwards it from there (blocks of 4 MiB and more per rank)."""
    bed.init_comms({"NCCL_ALGO": "Ring", "NCCL_PROTO": "Simple"})
    n = bed.nranks
    count = (4 << 20) + 4099                  # bytes per rank: over the threshold, ragged tail
    ins = [orc.fill_pattern(orc.U8, count, r) for r in range(n)]
    want = orc.all_gather_expected(ins)
    for outs, stats in bed.run_registered(orc.U8, orc.SUM, count, ins, runs=2, coll=COLL_ALLGATHER):
        for r in range(n):
            assert ex.check(orc.U8, outs[r], want, exact=True) is None, r
            assert stats[r]["direct_pieces"] > 0 and stats[r]["fifo_pieces"] == 0, (r, stats[r])
    (outs, stats), = bed.run_registered(orc.U8, orc.SUM, count, ins, register_ranks=[n - 1], coll=COLL_ALLGATHER)
    for r in range(n):
        assert ex.check(orc.U8, outs[r], want, exact=True) is None, r
    assert stats[n - 1]["direct_pieces"] == 0 and sum(s["direct_pieces"] for s in stats) > 0
    (outs, stats), = bed.run_registered(orc.U8, orc.SUM, count, ins, register_ranks=[], coll=COLL_ALLGATHER, in_place=True)
    for r in range(n):
        assert ex.check(orc.U8, outs[r], want, exact=True) is None, r
        assert stats[r]["direct_pieces"] == 0, (r, stats[r])


def test_registered_send_buffers_are_read_directly_by_reduce_scatter(bed):
    """Reduce-scatter with registered SEND buffers (the reference's direct-read primitives, src/device/prims_simple.h:
    243-283): a reducer reads the peers' contributions straight from their sendbuffs, nothing goes through the FIFOs
    (fifo_pieces == 0) and the senders stay in the kernel until they have been read.  Same bits as through the FIFOs;
    mixed registration states, in place, coarse-grained memory, a pre-multiplied sum (never read directly) and blocks
    below the size from which the words are exchanged."""
    bed.init_comms({"NCCL_ALGO": None, "RCCL_AMD_ONESHOT_MAX_BYTES": "0", "NCCL_PROTO": "Simple"})
    n = bed.nranks
    count = (1 << 20) + 1056                 # 4 MiB fp32 per rank, several rounds, ragged
    ins = [orc.fill_pattern(orc.F32, count * n, r) for r in range(n)]
    plan = bed.plan(COLL_REDUCESCATTER, count, orc.F32)
    assert plan["proto"] == 0 and plan["algo"] == ex.ALGO_DIRECT
    want = [ex.reduce_scatter(orc.F32, orc.SUM, ins, r, plan) for r in range(n)]
    for outs, stats in bed.run_registered(orc.F32, orc.SUM, count, ins, runs=2, coll=COLL_REDUCESCATTER):
        for r in range(n):
            assert ex.check(orc.F32, outs[r], want[r], exact=True) is None, r
            assert stats[r]["direct_pieces"] > 0 and stats[r]["fifo_pieces"] == 0, (r, stats[r])
    for outs, stats in bed.run_registered(orc.F32, orc.SUM, count, ins, runs=2, coll=COLL_REDUCESCATTER, graph=True):
        for r in range(n):
            assert ex.check(orc.F32, outs[r], want[r], exact=True) is None, r
            assert stats[r]["fifo_pieces"] == 0, (r, stats[r])
    # only rank 0's sendbuff registered: every other reducer reads that one contribution directly
    (outs, stats), = bed.run_registered(orc.F32, orc.SUM, count, ins, register_ranks=[0], coll=COLL_REDUCESCATTER)
    for r in range(n):
        assert ex.check(orc.F32, outs[r], want[r], exact=True) is None, r
    assert stats[0]["direct_pieces"] == 0
    for r in range(1, n):
        assert stats[r]["direct_pieces"] > 0 and stats[r]["fifo_pieces"] == (n - 2) * stats[r]["direct_pieces"], (r, stats[r])
    # in place (recvbuff = own block of the registered sendbuff), integer average
    ins2 = [orc.fill_pattern(orc.I32, count * n, r) for r in range(n)]
    want2 = [ex.reduce_scatter(orc.I32, orc.AVG, ins2, r, bed.plan(COLL_REDUCESCATTER, count, orc.I32)) for r in range(n)]
    (outs, stats), = bed.run_registered(orc.I32, orc.AVG, count, ins2, in_place=True, coll=COLL_REDUCESCATTER)
    for r in range(n):
        assert ex.check(orc.I32, outs[r], want2[r], exact=True) is None, r
        assert stats[r]["fifo_pieces"] == 0, (r, stats[r])
    # a float average is a pre-multiplied sum: contributions are scaled by their sender, so they travel through the FIFOs
    want3 = [ex.reduce_scatter(orc.F32, orc.AVG, ins, r, plan) for r in range(n)]
    (outs, stats), = bed.run_registered(orc.F32, orc.AVG, count, ins, coll=COLL_REDUCESCATTER)
    for r in range(n):
        assert ex.check(orc.F32, outs[r], want3[r], exact=True) is None, r
        assert stats[r]["direct_pieces"] == 0, (r, stats[r])
    # nothing registered / coarse-grained memory: FIFOs, same bits
    for kw in (dict(register_ranks=[]), dict(fine=False)):
        (outs, stats), = bed.run_registered(orc.F32, orc.SUM, count, ins, coll=COLL_REDUCESCATTER, **kw)
        for r in range(n):
            assert ex.check(orc.F32, outs[r], want[r], exact=True) is None, (kw, r)
            assert stats[r]["direct_pieces"] == 0, (kw, r, stats[r])
    # below the threshold: no words, FIFOs
    c4 = (128 << 10) + 16
    ins4 = [orc.fill_pattern(orc.F32, c4 * n, r) for r in range(n)]
    want4 = [ex.reduce_scatter(orc.F32, orc.SUM, ins4, r, bed.plan(COLL_REDUCESCATTER, c4, orc.F32)) for r in range(n)]
    (outs, stats), = bed.run_registered(orc.F32, orc.SUM, c4, ins4, coll=COLL_REDUCESCATTER)
    for r in range(n):
        assert ex.check(orc.F32, outs[r], want4[r], exact=True) is None, r
        assert stats[r]["direct_pieces"] == 0, (r, stats[r])


