"""The benchmark configurations at their stated sizes, and rank counts that are not powers of two.

Full size.  Every configuration the headline numbers are quoted on runs once at its real size with all ranks on the
box's GPU(s) and the WHOLE output of every rank is compared with the CPU oracle: bit for bit against the oracle's fold
in the schedule's association order, and within the reference tolerance (test/common/PtrUnion.cpp:313-342) against the
reference's rank-order host fold (test/common/PrepDataFuncs.cpp:95-146).  Buckets come from the oracle's hash pattern
(every rank makes its own, the parent makes all of them for the expectation), because the reference's own pattern
repeats every 256 elements and would not notice a piece that lands a multiple of 256 elements off.
  2 ranks : ncclAllReduce sum fp32, 1 GiB, all-pairs and ring schedules, out of place and in place
  8 ranks : ncclReduceScatter + ncclAllGather fp32, 256 MiB per rank
            ncclAllReduce sum bf16 and fp16, 1 GiB, NCCL_ALGO=Tree
            ncclAllReduce sum fp32, 1 GiB, default schedule (the headline configuration)
            ncclAllReduce sum fp32, 256 MiB, NCCL_P2P_DISABLE=1 (host-staged, loopback TCP)
(reference counterparts of the sizes: test/AllReduceTests.cpp:105-130 runs its bf16 case at 64 Mi elements.)

Odd rank counts.  The reference sweeps every rank count from UT_MIN_GPUS to UT_MAX_GPUS (test/common/TestBed.cpp:614-765);
3 and 6 ranks take the reduce-copy shapes that have no specialised loop, the ring of 3, and averages whose 1/n is not a
power of two: with a fused multiply-add the pre-multiplied sum would differ in the last bit from the reference's two
roundings (FuncPreMulSum then FuncSum, src/device/reduce_kernel.h:364-531).
"""
import os
import tempfile

import numpy as np
import pytest

import expected as ex
import oracle as orc
from testbed import COLL_ALLGATHER, COLL_ALLREDUCE, COLL_REDUCESCATTER, TestBed

pytestmark = pytest.mark.gpu

GiB = 1 << 30


def _devices():
    import torch
    return tuple(range(torch.cuda.device_count()))


def _es(dt):
    return np.dtype(orc.NP_OF[dt]).itemsize


@pytest.fixture(scope="module")
def scratch_dir():
    with tempfile.TemporaryDirectory(prefix="rccl_amd_fullsize_") as d:
        yield d


def _expect_allreduce(bed, dt, op, count, seed, scratch_dir, tag):
    """Expected all-reduce result in the planned schedule's order (want) and the reference's host fold (host), saved
    where the ranks can map them."""
    n = bed.nranks
    ins = [orc.fill_hashed(dt, count, r, seed) for r in range(n)]
    plan = bed.plan(COLL_ALLREDUCE, count, dt)
    want = ex.allreduce(dt, op, ins, plan)
    host = orc.host_fold(dt, op, ins)
    del ins
    wp, hp = os.path.join(scratch_dir, f"{tag}_want.npy"), os.path.join(scratch_dir, f"{tag}_host.npy")
    np.save(wp, want)
    np.save(hp, host)
    return plan, wp, hp


def _assert_reports(reports, what, plan=None):
    for r, rep in enumerate(reports):
        assert rep["err"] == 0, (what, r, "asynchronous error", rep["err"])
        assert rep["guard_ok"], (what, r, "bytes behind the receive buffer were written")
        assert rep["bad"] == 0, (what, r, f"{rep['bad']} of {rep['elems']} elements differ from the oracle, first at {rep['first']}",
                                 rep.get("got"), rep.get("want"), plan)
        assert rep["host_bad"] == 0, (what, r, f"{rep['host_bad']} elements outside the reference tolerance of the host fold, "
                                      f"first at {rep['host_first']}", plan)


def _allreduce_full(bed, dt, count, seed, scratch_dir, tag, in_place=False):
    plan, wp, hp = _expect_allreduce(bed, dt, orc.SUM, count, seed, scratch_dir, tag)
    reps = bed.run_big(COLL_ALLREDUCE, dt, orc.SUM, count, seed, want=wp, host=hp, in_place=in_place)
    _assert_reports(reps, tag, plan)
    os.remove(wp)
    os.remove(hp)
    return plan


@pytest.fixture(scope="module")
def bed2():
    tb = TestBed(2, devices=_devices(), timeout=600)
    yield tb
    tb.close()


@pytest.fixture(scope="module")
def bed8():
    tb = TestBed(8, devices=_devices(), timeout=600)
    yield tb
    tb.close()


@pytest.mark.parametrize("algo", ["Direct", "Ring"])
def test_allreduce_fp32_1GiB_two_ranks(bed2, scratch_dir, algo):
    """ncclAllReduce sum fp32, 1 GiB, 2 ranks: the three-phase FIFO schedule over every channel and round."""
    bed2.init_comms({"NCCL_ALGO": algo})
    count = GiB // 4
    plan = _allreduce_full(bed2, orc.F32, count, 0x5EED, scratch_dir, f"ar2_{algo}")
    assert plan["proto"] == 0 and plan["algo"] == {"Direct": ex.ALGO_DIRECT, "Ring": ex.ALGO_RING}[algo], plan
    if algo == "Direct":
        _allreduce_full(bed2, orc.F32, count, 0x5EEE, scratch_dir, "ar2_Direct_in_place", in_place=True)
    # one element short of a whole number of packs: ragged last round, sub-pack tail
    _allreduce_full(bed2, orc.F32, count // 4 - 1, 0x5EEF, scratch_dir, f"ar2_{algo}_ragged")
    if algo == "Direct":
        # the checker itself: one flipped bit in the very last element of the expectation must be found, there and only there
        c2 = count // 4 - 1
        plan, wp, hp = _expect_allreduce(bed2, orc.F32, orc.SUM, c2, 0x5EEF, scratch_dir, "ar2_control")
        want = np.load(wp)
        want.view(np.uint32)[c2 - 1] ^= 1
        np.save(wp, want)
        for rep in bed2.run_big(COLL_ALLREDUCE, orc.F32, orc.SUM, c2, 0x5EEF, want=wp, host=hp):
            assert rep["elems"] == c2 and rep["bad"] == 1 and rep["first"] == c2 - 1 and rep["host_bad"] == 0, rep
    bed2.init_comms({"NCCL_ALGO": None})


def test_reduce_scatter_all_gather_256MiB_per_rank_eight_ranks(bed8, scratch_dir):
    """ncclReduceScatter + ncclAllGather fp32, 256 MiB per rank, 8 ranks (2 GiB send / receive buffers)."""
    bed8.init_comms({"NCCL_ALGO": None})
    n, count, seed = 8, (256 << 20) // 4, 0xC3
    ins = [orc.fill_hashed(orc.F32, count * n, r, seed) for r in range(n)]
    plan = bed8.plan(COLL_REDUCESCATTER, count, orc.F32)
    assert plan["algo"] == ex.ALGO_DIRECT and plan["proto"] == 0, plan
    want = orc.device_fold(orc.F32, orc.SUM, ins)          # all-pairs: rank order, the reference's host order
    host = orc.host_fold(orc.F32, orc.SUM, ins)
    del ins
    wp, hp = os.path.join(scratch_dir, "rs_want.npy"), os.path.join(scratch_dir, "rs_host.npy")
    np.save(wp, want)
    np.save(hp, host)
    del want, host
    _assert_reports(bed8.run_big(COLL_REDUCESCATTER, orc.F32, orc.SUM, count, seed, want=wp, host=hp), "reduce-scatter", plan)
    os.remove(wp)
    os.remove(hp)
    _assert_reports(bed8.run_big(COLL_ALLGATHER, orc.F32, orc.SUM, count, seed), "all-gather")


@pytest.mark.parametrize("dt", [orc.BF16, orc.F16])
def test_allreduce_16bit_1GiB_tree_eight_ranks(bed8, scratch_dir, dt):
    """ncclAllReduce sum bf16 / fp16, 1 GiB, 8 ranks, NCCL_ALGO=Tree (the chain in ring order)."""
    bed8.init_comms({"NCCL_ALGO": "Tree"})
    plan = _allreduce_full(bed8, dt, GiB // 2, 0x7EE + dt, scratch_dir, f"tree_{orc.DT_NAMES[dt]}")
    assert plan["algo"] == ex.ALGO_TREE and plan["proto"] == 0, plan
    bed8.init_comms({"NCCL_ALGO": None})


def test_allreduce_fp32_1GiB_eight_ranks(bed8, scratch_dir):
    """The headline configuration: ncclAllReduce sum fp32, 1 GiB, 8 ranks, default (all-pairs) schedule."""
    bed8.init_comms({"NCCL_ALGO": None})
    plan = _allreduce_full(bed8, orc.F32, GiB // 4, 0xA11, scratch_dir, "ar8")
    assert plan["algo"] == ex.ALGO_DIRECT and plan["proto"] == 0, plan


def test_allreduce_fp32_256MiB_host_staged_eight_ranks(bed8, scratch_dir):
    """ncclAllReduce sum fp32, 256 MiB, 8 ranks, every hop through pinned host memory and loopback TCP."""
    try:
        bed8.init_comms({"NCCL_P2P_DISABLE": "1", "RCCL_AMD_NCHANNELS": "4", "NCCL_ALGO": None})
        _allreduce_full(bed8, orc.F32, (256 << 20) // 4, 0x7C9, scratch_dir, "tcp8")
    finally:
        bed8.destroy_comms()
        bed8.init_comms({"NCCL_P2P_DISABLE": None, "RCCL_AMD_NCHANNELS": bed8.base_env["RCCL_AMD_NCHANNELS"], "NCCL_ALGO": None})


# ------------------------------------------------------------------ rank counts that are not powers of two
@pytest.fixture(scope="module", params=[3, 6])
def odd_bed(request):
    tb = TestBed(request.param, devices=_devices())
    yield tb
    tb.close()


def _check_allreduce(bed, dt, op, count, in_place=False):
    n = bed.nranks
    ins = [orc.fill_hashed(dt, count, r, 0x0DD + count) for r in range(n)]
    outs = bed.run(COLL_ALLREDUCE, dt, op, count, ins, count * _es(dt), in_place=in_place)
    plan = bed.plan(COLL_ALLREDUCE, count, dt)
    want = ex.allreduce(dt, op, ins, plan)
    host = orc.host_fold(dt, op, ins)
    for r in range(n):
        assert ex.check(dt, outs[r], want, exact=True) is None, (r, orc.DT_NAMES[dt], orc.OP_NAMES[op], count, plan,
                                                                  ex.check(dt, outs[r], want, exact=True))
        assert ex.check(dt, outs[r], host, exact=False) is None, (r, "vs host fold", plan)


ODD_ENVS = {
    "default": {"NCCL_ALGO": None, "NCCL_PROTO": None},
    "DirectFifo": {"NCCL_ALGO": "Direct", "NCCL_PROTO": "Simple", "RCCL_AMD_ONESHOT_MAX_BYTES": "0", "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": "0"},
    "TwoExchange": {"NCCL_ALGO": "Direct", "NCCL_PROTO": "Simple", "RCCL_AMD_ONESHOT_MAX_BYTES": "0", "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": str(64 << 20)},
    "DirectFifoOneWorkgroup": {"NCCL_ALGO": "Direct", "NCCL_PROTO": "Simple", "RCCL_AMD_ONESHOT_MAX_BYTES": "0", "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": "0",
                               "RCCL_AMD_FUSED_ROLES": "1"},
    "Ring+LL": {"NCCL_ALGO": "Ring", "NCCL_PROTO": "LL"},
    "Tree+LL": {"NCCL_ALGO": "Tree", "NCCL_PROTO": "LL"},
    "Ring": {"NCCL_ALGO": "Ring", "NCCL_PROTO": "Simple"},
    "Tree": {"NCCL_ALGO": "Tree", "NCCL_PROTO": "Simple"},
}


@pytest.mark.parametrize("mode", list(ODD_ENVS))
def test_odd_rank_counts(odd_bed, mode):
    """Averages (1/3 and 1/6 are not powers of two), sums and integer averages through every schedule and protocol."""
    bed = odd_bed
    n = bed.nranks
    bed.init_comms(ODD_ENVS[mode])
    for count in (1250, 70001, 300007, (3 << 20) + 5):
        for dt in (orc.F16, orc.F32):
            _check_allreduce(bed, dt, orc.AVG, count)
    _check_allreduce(bed, orc.BF16, orc.AVG, 100003)
    _check_allreduce(bed, orc.F64, orc.AVG, 30011)
    _check_allreduce(bed, orc.F32, orc.SUM, (1 << 20) + 3, in_place=True)
    _check_allreduce(bed, orc.I32, orc.AVG, 70001)
    _check_allreduce(bed, orc.U8, orc.AVG, 300007)
    _check_allreduce(bed, orc.I64, orc.PROD, 5001)
    # the known-answer case of test/StandaloneTests.cpp:191-246 at this rank count
    x = np.arange(1250, dtype=np.int32)
    for o in bed.run(COLL_ALLREDUCE, orc.I32, orc.SUM, 1250, [x] * n, 1250 * 4):
        assert np.array_equal(o.view(np.int32), n * x)
    # reduce-scatter and all-gather, small and FIFO-sized, ragged
    for dt, op, count in ((orc.F32, orc.AVG, 1000), (orc.F16, orc.AVG, 350003), (orc.I32, orc.SUM, 1048576 + 7), (orc.BF16, orc.MAX, 104857)):
        ins = [orc.fill_hashed(dt, count * n, r, 0x0DD) for r in range(n)]
        outs = bed.run(COLL_REDUCESCATTER, dt, op, count, ins, count * _es(dt))
        plan = bed.plan(COLL_REDUCESCATTER, count, dt)
        for r in range(n):
            assert ex.check(dt, outs[r], ex.reduce_scatter(dt, op, ins, r, plan), exact=True) is None, (orc.DT_NAMES[dt], count, r, plan)
            assert ex.check(dt, outs[r], orc.reduce_scatter_expected(dt, op, ins, r), exact=False) is None
    for dt, count, in_place in ((orc.F16, 1001, False), (orc.U8, 1039203, False), (orc.I32, 1048576 + 3, True)):
        ins = [orc.fill_hashed(dt, count, r, 0xA6) for r in range(n)]
        outs = bed.run(COLL_ALLGATHER, dt, orc.SUM, count, ins, count * n * _es(dt), in_place=in_place)
        for r in range(n):
            assert ex.check(dt, outs[r], orc.all_gather_expected(ins), exact=True) is None, (orc.DT_NAMES[dt], count, r)
    bed.init_comms({"NCCL_ALGO": None, "NCCL_PROTO": None, "RCCL_AMD_ONESHOT_MAX_BYTES": None, "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": None})

