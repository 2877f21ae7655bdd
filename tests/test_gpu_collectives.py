"""GPU parity of ncclAllReduce / ncclReduceScatter / ncclAllGather against the CPU oracle.

Ranks are separate processes created with ncclCommInitRank (HIP IPC windows), several per GPU
when the box has fewer GPUs than ranks — the reference's multi-process, ranks-per-GPU test mode
(test/common/TestBed.cpp:564-583).  Cases follow the reference's own lists
(test/AllReduceTests.cpp:10-130, test/ReduceScatterTests.cpp:10-120, test/AllGatherTests.cpp:10-120,
test/StandaloneTests.cpp:191-246), with the largest sizes trimmed to keep the suite short.

Bars: integer types and copies bit-exact; floating point bit-exact against the oracle's fold in the
schedule's own association order AND within the reference tolerance (fp32 1e-5, fp64 1e-12,
fp16/bf16 9e-2 absolute, test/common/PtrUnion.cpp:313-342) of the reference's rank-order host fold.
"""
import numpy as np
import pytest

import expected as ex
import oracle as orc
from testbed import COLL_ALLGATHER, COLL_ALLREDUCE, COLL_BROADCAST, COLL_REDUCE, COLL_REDUCESCATTER, TestBed

pytestmark = pytest.mark.gpu

ALGOS = {"Direct": ex.ALGO_DIRECT, "Ring": ex.ALGO_RING, "Tree": ex.ALGO_TREE, "OneExchange": ex.ALGO_DIRECT,
         "TwoExchange": ex.ALGO_DIRECT, "DirectOneWorkgroup": ex.ALGO_DIRECT}
# "Direct": the three-phase all-pairs schedule at every size; "OneExchange" / "TwoExchange": every all-reduce above the
# low-latency limit as one exchange / as two (the planner's choices for mid-size payloads, here forced for all sizes; two
# exchanges exist for more than two ranks); "DirectOneWorkgroup": the three-phase schedule with one workgroup per channel
# walking all three roles (RCCL_AMD_FUSED_ROLES=1); other tests run the default mix.
ALGO_ENV = {"Direct": {"NCCL_ALGO": "Direct", "RCCL_AMD_ONESHOT_MAX_BYTES": "0", "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": "0"},
            "DirectOneWorkgroup": {"NCCL_ALGO": "Direct", "RCCL_AMD_ONESHOT_MAX_BYTES": "0", "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": "0",
                                   "RCCL_AMD_FUSED_ROLES": "1"},
            "TwoExchange": {"NCCL_ALGO": "Direct", "RCCL_AMD_ONESHOT_MAX_BYTES": "0", "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": str(64 << 20)},
            "Ring": {"NCCL_ALGO": "Ring", "RCCL_AMD_ONESHOT_MAX_BYTES": None},
            "Tree": {"NCCL_ALGO": "Tree", "RCCL_AMD_ONESHOT_MAX_BYTES": None},
            "OneExchange": {"NCCL_ALGO": "Direct", "RCCL_AMD_ONESHOT_MAX_BYTES": str(32 << 20)}}


def _devices():
    import torch
    return tuple(range(torch.cuda.device_count()))


@pytest.fixture(scope="module", params=[2, 4, 8])
def bed(request):
    tb = TestBed(request.param, devices=_devices())
    yield tb
    tb.close()


def _es(dt):
    return np.dtype(orc.NP_OF[dt]).itemsize


def _check_allreduce(bed, dt, op, count, in_place=False, pattern=True, seed=0, host_tol=True):
    n = bed.nranks
    if pattern:
        ins = [orc.fill_pattern(dt, count, r) for r in range(n)]
    else:
        rng = np.random.default_rng(seed)
        ins = [rng.uniform(-1, 1, count).astype(np.float32) for _ in range(n)]
    outs = bed.run(COLL_ALLREDUCE, dt, op, count, ins, count * _es(dt), in_place=in_place)
    plan = bed.plan(COLL_ALLREDUCE, count, dt)
    want = ex.allreduce(dt, op, ins, plan)
    host = orc.host_fold(dt, op, ins)
    for r in range(n):
        assert ex.check(dt, outs[r], want, exact=True) is None, (r, ex.check(dt, outs[r], want, exact=True), plan)
        if host_tol:   # (fp8 sums in another order differ by whole ulps; the reference tests fp8 with min/max only)
            assert ex.check(dt, outs[r], host, exact=False) is None, (r, "vs host fold", plan)


@pytest.mark.parametrize("algo", list(ALGOS))
def test_allreduce_cases(bed, algo):
    bed.init_comms(ALGO_ENV[algo])
    assert bed.plan(COLL_ALLREDUCE, 1 << 20, orc.F32)["algo"] == ALGOS[algo]
    want_proto = {"OneExchange": 2, "TwoExchange": 3 if bed.nranks > 2 else 0}.get(algo, 0)
    assert bed.plan(COLL_ALLREDUCE, 1 << 22, orc.F32)["proto"] == want_proto
    # known answer: int32 sum N=1250, in[i]=i on every rank -> nranks*i (StandaloneTests.cpp:191-246)
    x = np.arange(1250, dtype=np.int32)
    outs = bed.run(COLL_ALLREDUCE, orc.I32, orc.SUM, 1250, [x] * bed.nranks, 1250 * 4)
    for o in outs:
        assert np.array_equal(o.view(np.int32), bed.nranks * x)
    # AllReduceTests.cpp:10-27 fp32 sum out of place
    for count in (393216, 384):
        _check_allreduce(bed, orc.F32, orc.SUM, count)
    # :29-46 fp16/fp64/fp8 min
    for dt in (orc.F16, orc.F64, orc.F8E4M3):
        _check_allreduce(bed, dt, orc.MIN, 12888)
    # :67-84 fp8 (E5M2) max in place; fp8 sum/avg are bit-exact against the schedule-order fold only
    _check_allreduce(bed, orc.F8E5M2, orc.MAX, 12888, in_place=True)
    _check_allreduce(bed, orc.F8E4M3, orc.SUM, 100003, host_tol=False)
    _check_allreduce(bed, orc.F8E5M2, orc.AVG, 70001, host_tol=False)
    # :48-65 int32/int8 prod in place
    for dt in (orc.I32, orc.I8):
        _check_allreduce(bed, dt, orc.PROD, 384, in_place=True)
    # :67-84 int32 max in place
    _check_allreduce(bed, orc.I32, orc.MAX, 12888, in_place=True)
    # :86-103 uint8/uint64 sum
    for dt in (orc.U8, orc.U64):
        _check_allreduce(bed, dt, orc.SUM, 2500)
    # :105-130 bf16 sum, large (trimmed from 64 Mi elements)
    _check_allreduce(bed, orc.BF16, orc.SUM, 3 * 1024 * 1024 + 1000)
    # average: integer post-divide, float pre-multiply (enqueue.cc:2006-2044)
    for dt in (orc.I64, orc.U8, orc.U32, orc.F32, orc.F16, orc.BF16, orc.F64):
        _check_allreduce(bed, dt, orc.AVG, 70001)
    # random fp32 data, several rounds per channel, ragged tail
    _check_allreduce(bed, orc.F32, orc.SUM, 5 * 1024 * 1024 + 123, pattern=False, seed=5)
    _check_allreduce(bed, orc.F32, orc.SUM, 3 * 1024 * 1024 + 77, in_place=True, pattern=False, seed=6)
    bed.init_comms({"NCCL_ALGO": None, "RCCL_AMD_ONESHOT_MAX_BYTES": None})


def test_allreduce_premul_sum(bed):
    """User-created pre-multiply-sum op with a different scalar per rank (AllReduceTests.cpp:176-243)."""
    bed.init_comms({"NCCL_ALGO": None})
    n, count = bed.nranks, 50000
    ins = [orc.fill_pattern(orc.F32, count, r) for r in range(n)]
    scal = [np.float32(1.0 / (r + 2)) for r in range(n)]
    outs = bed.run(COLL_ALLREDUCE, orc.F32, orc.SUM, count, ins, count * 4, premul=[s.tobytes() for s in scal])
    want = orc.device_fold(orc.F32, orc.SUM, [a * s for a, s in zip(ins, scal)])
    for o in outs:
        assert np.array_equal(o.view(np.float32), want)


def test_premul_sum_scalar_in_device_memory(bed):
    """ncclScalarDevice (AllReduceTests.cpp:176-243 runs its pre-multiply cases with the scalar in device memory too):
    the kernel reads the scalar when it runs.  Eager launches through both protocols, then a launch captured into a
    hipGraph and replayed after the scalar changed: the second replay must use the new value, which a scalar fetched
    at enqueue time would not (reference: redOpArg carried as a pointer, src/enqueue.cc:1962-2057)."""
    n = bed.nranks
    for proto in (None, "Simple"):
        bed.init_comms({"NCCL_ALGO": None, "NCCL_PROTO": proto})
        for count in (3001, 300007):
            ins = [orc.fill_pattern(orc.F32, count, r) for r in range(n)]
            s1 = [np.float32(1.0 / (r + 2)) for r in range(n)]
            s2 = [np.float32(r + 1.5) for r in range(n)]
            outs = bed.run(COLL_ALLREDUCE, orc.F32, orc.SUM, count, ins, count * 4, premul=[s.tobytes() for s in s1],
                           premul_device=True, iters=2)
            want = orc.device_fold(orc.F32, orc.SUM, [a * s for a, s in zip(ins, s1)])
            for o in outs:
                assert np.array_equal(o.view(np.float32), want), (proto, count)
            outs = bed.run(COLL_ALLREDUCE, orc.F32, orc.SUM, count, ins, count * 4, premul=[s.tobytes() for s in s1],
                           premul_device=True, premul2=[s.tobytes() for s in s2], graph=True)
            want = orc.device_fold(orc.F32, orc.SUM, [a * s for a, s in zip(ins, s2)])
            for o in outs:
                assert np.array_equal(o.view(np.float32), want), (proto, count, "replay after the scalar changed")
    # half precision through the reduce-scatter kernel path
    bed.init_comms({"NCCL_ALGO": None, "NCCL_PROTO": None})
    count = 40001
    ins = [orc.fill_pattern(orc.F16, count * n, r) for r in range(n)]
    sc = [np.float16(0.5 + r) for r in range(n)]
    outs = bed.run(COLL_REDUCESCATTER, orc.F16, orc.SUM, count, ins, count * 2, premul=[s.tobytes() for s in sc], premul_device=True)
    full = orc.device_fold(orc.F16, orc.SUM, [a * s for a, s in zip(ins, sc)])
    for r in range(n):
        assert np.array_equal(outs[r].view(np.float16), full[r * count:(r + 1) * count]), r


# "Direct": the default mix (line protocol, one exchange, FIFO schedule by size); "DirectFifo": the FIFO schedule at every size
# above the line protocol's
RS_AG_ENV = {"Direct": {"NCCL_ALGO": "Direct"}, "Ring": {"NCCL_ALGO": "Ring"},
             "DirectFifo": {"NCCL_ALGO": "Direct", "RCCL_AMD_ONESHOT_MAX_BYTES": "0"}}


@pytest.mark.parametrize("algo", list(RS_AG_ENV))
def test_reduce_scatter_cases(bed, algo):
    bed.init_comms(RS_AG_ENV[algo])
    n = bed.nranks
    # ReduceScatterTests.cpp: fp32 max {393216,384}; fp64/bf16 max; int32 prod in place 542357 (trimmed);
    # uint8/fp16 min in place 246; int64/uint8 avg 1024; uint32/uint64 avg (trimmed)
    cases = [(orc.F32, orc.MAX, 393216, False), (orc.F32, orc.MAX, 384, False), (orc.F64, orc.MAX, 104857, False),
             (orc.BF16, orc.MAX, 104857, False), (orc.F8E4M3, orc.MAX, 104857, False), (orc.I32, orc.PROD, 54235, True), (orc.U8, orc.MIN, 246, True),
             (orc.F16, orc.MIN, 246, True), (orc.I64, orc.AVG, 1024, False), (orc.U8, orc.AVG, 1024, False),
             (orc.U32, orc.AVG, 648541, False), (orc.U64, orc.AVG, 64853, False), (orc.F32, orc.SUM, 1048576 + 7, False),
             (orc.BF16, orc.AVG, 200003, False), (orc.F32, orc.AVG, 150001, True), (orc.F32, orc.SUM, 524288 + 5, True)]
    for dt, op, count, in_place in cases:
        ins = [orc.fill_pattern(dt, count * n, r) for r in range(n)]
        outs = bed.run(COLL_REDUCESCATTER, dt, op, count, ins, count * _es(dt), in_place=in_place)
        plan = bed.plan(COLL_REDUCESCATTER, count, dt)
        for r in range(n):
            want = ex.reduce_scatter(dt, op, ins, r, plan)
            host = orc.reduce_scatter_expected(dt, op, ins, r)
            assert ex.check(dt, outs[r], want, exact=True) is None, (orc.DT_NAMES[dt], orc.OP_NAMES[op], count, r, plan)
            assert ex.check(dt, outs[r], host, exact=False) is None


def test_integer_average_through_the_fifos(bed):
    """ncclAvg of every integer type (sum, then divide by the rank count: FuncSumPostDiv, src/device/reduce_kernel.h:546-573)
    on the FIFO schedules at sizes of several rounds per channel: the reducers divide, so they are the slow side and
    the senders run into the credit limit.  Reduce-scatter with a ragged last round, then all-reduce."""
    n = bed.nranks
    bed.init_comms({"NCCL_ALGO": "Direct", "RCCL_AMD_ONESHOT_MAX_BYTES": "0", "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": "0"})
    for dt in (orc.I8, orc.U8, orc.I32, orc.U32, orc.I64, orc.U64):
        count = (2 << 20) // _es(dt) + 48          # above the line protocol's limit for every type
        ins = [orc.fill_pattern(dt, count * n, r) for r in range(n)]
        outs = bed.run(COLL_REDUCESCATTER, dt, orc.AVG, count, ins, count * _es(dt))
        plan = bed.plan(COLL_REDUCESCATTER, count, dt)
        assert plan["proto"] == 0, plan
        for r in range(n):
            assert ex.check(dt, outs[r], ex.reduce_scatter(dt, orc.AVG, ins, r, plan)) is None, (orc.DT_NAMES[dt], r, plan)
        _check_allreduce(bed, dt, orc.AVG, (4 << 20) // _es(dt) + 16)
    bed.init_comms({"NCCL_ALGO": None, "RCCL_AMD_ONESHOT_MAX_BYTES": None, "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": None})


@pytest.mark.parametrize("algo", list(RS_AG_ENV))
def test_all_gather_cases(bed, algo):
    bed.init_comms(RS_AG_ENV[algo])
    n = bed.nranks
    # AllGatherTests.cpp: fp16/fp32 {1048576,500}; bf16/fp64 586; int32 in place {104857,264}; int8/int64 in place 958;
    # uint8 {1039203,2500}; uint32/uint64 896
    cases = [(orc.F16, 1048576, False), (orc.F32, 500, False), (orc.BF16, 586, False), (orc.F64, 586, False),
             (orc.F8E5M2, 586, False),
             (orc.I32, 104857, True), (orc.I32, 264, True), (orc.I8, 958, True), (orc.I64, 958, True),
             (orc.U8, 1039203, False), (orc.U8, 2500, False), (orc.U32, 896, False), (orc.U64, 896, False)]
    for dt, count, in_place in cases:
        ins = [orc.fill_pattern(dt, count, r) for r in range(n)]
        outs = bed.run(COLL_ALLGATHER, dt, orc.SUM, count, ins, count * n * _es(dt), in_place=in_place)
        want = orc.all_gather_expected(ins)
        for r in range(n):
            assert ex.check(dt, outs[r], want, exact=True) is None, (orc.DT_NAMES[dt], count, in_place, r)


def test_broadcast_and_reduce(bed):
    """ncclBroadcast / ncclReduce (expected values as test/common/PrepDataFuncs.cpp:43-78 and :80-148):
    every rank ends with the root's pattern; the root ends with the rank-order fold.  Payloads up to 512 KiB take
    the low-latency protocol unless NCCL_PROTO=Simple; the larger ones the FIFO."""
    n = bed.nranks
    for proto in (None, "Simple"):
        bed.init_comms({"NCCL_ALGO": None, "NCCL_PROTO": proto})
        assert bed.plan(COLL_BROADCAST, 53327, orc.F16)["proto"] == (1 if proto is None else 0)
        assert bed.plan(COLL_REDUCE, 131073, orc.F32)["proto"] == 0
        _broadcast_and_reduce_cases(bed)
    bed.init_comms({"NCCL_ALGO": None, "NCCL_PROTO": None})
    # rooted low-latency launches between all-reduces, roots changing every launch: the line buffers' epochs stay in step
    x = np.arange(3000, dtype=np.int32)
    for k in range(6):
        root = (3 * k + 1) % n
        ins = [orc.fill_pattern(orc.I32, 3000, root if r == root else 100 + r) for r in range(n)]
        outs = bed.run(COLL_BROADCAST, orc.I32, orc.SUM, 3000, ins, 12000, root=root, iters=2)
        for r in range(n):
            assert ex.check(orc.I32, outs[r], ins[root], exact=True) is None, (k, r)
        outs = bed.run(COLL_REDUCE, orc.I32, orc.SUM, 3000, [x + r for r in range(n)], 12000, root=(root + 1) % n, iters=3)
        assert np.array_equal(outs[(root + 1) % n].view(np.int32), n * x + n * (n - 1) // 2)
        outs = bed.run(COLL_ALLREDUCE, orc.I32, orc.SUM, 3000, [x] * n, 12000)
        for o in outs:
            assert np.array_equal(o.view(np.int32), n * x)


def _broadcast_and_reduce_cases(bed):
    n = bed.nranks
    for root in (0, n - 1):
        for dt, count, in_place in ((orc.F32, 1048576, False), (orc.F16, 53327, False), (orc.I8, 1000003, True), (orc.F64, 4099, False),
                                    (orc.U8, 524288, True), (orc.U8, 3, False)):
            ins = [orc.fill_pattern(dt, count, root if r == root else 200 + r) for r in range(n)]   # only the root's input counts
            outs = bed.run(COLL_BROADCAST, dt, orc.SUM, count, ins, count * _es(dt), in_place=in_place, root=root)
            for r in range(n):
                assert ex.check(dt, outs[r], ins[root], exact=True) is None, (root, orc.DT_NAMES[dt], count, r)
        for dt, op, count in ((orc.F32, orc.SUM, 1048576), (orc.I32, orc.PROD, 384), (orc.BF16, orc.MAX, 53327), (orc.U64, orc.AVG, 70001),
                              (orc.F32, orc.AVG, 70001), (orc.I8, orc.SUM, 1), (orc.F32, orc.SUM, 131072)):
            ins = [orc.fill_pattern(dt, count, r) for r in range(n)]
            outs = bed.run(COLL_REDUCE, dt, op, count, ins, count * _es(dt), root=root)
            assert ex.check(dt, outs[root], orc.device_fold(dt, op, ins), exact=True) is None, (root, orc.DT_NAMES[dt], orc.OP_NAMES[op])
            assert ex.check(dt, outs[root], orc.host_fold(dt, op, ins), exact=False) is None
            for r in range(n):
                if r != root:      # recvbuff of the other ranks is left alone
                    assert (outs[r] == 0xA5).all()


def test_managed_memory_buffers(bed):
    """hipMallocManaged send/receive buffers, the reference's managed-memory cases (AllReduceTests.cpp:86-103 uint8/uint64
    sum N=2500; ReduceScatterTests.cpp int64/uint8 avg N=1024; AllGatherTests.cpp uint8 N in {1039203, 2500})."""
    bed.init_comms({"NCCL_ALGO": None})
    n = bed.nranks
    for dt in (orc.U8, orc.U64):
        ins = [orc.fill_pattern(dt, 2500, r) for r in range(n)]
        outs = bed.run(COLL_ALLREDUCE, dt, orc.SUM, 2500, ins, 2500 * _es(dt), managed=True)
        for o in outs:
            assert ex.check(dt, o, orc.host_fold(dt, orc.SUM, ins), exact=True) is None
    for dt in (orc.I64, orc.U8):
        ins = [orc.fill_pattern(dt, 1024 * n, r) for r in range(n)]
        outs = bed.run(COLL_REDUCESCATTER, dt, orc.AVG, 1024, ins, 1024 * _es(dt), managed=True)
        for r in range(n):
            assert ex.check(dt, outs[r], orc.reduce_scatter_expected(dt, orc.AVG, ins, r), exact=True) is None
    for count in (1039203 // 4, 2500):                      # (the large case trimmed to a quarter)
        ins = [orc.fill_pattern(orc.U8, count, r) for r in range(n)]
        outs = bed.run(COLL_ALLGATHER, orc.U8, orc.SUM, count, ins, count * n, managed=True)
        for o in outs:
            assert ex.check(orc.U8, o, orc.all_gather_expected(ins), exact=True) is None
    # a large managed all-reduce through the FIFO protocol
    count = (1 << 20) + 5
    ins = [orc.fill_pattern(orc.F32, count, r) for r in range(n)]
    outs = bed.run(COLL_ALLREDUCE, orc.F32, orc.SUM, count, ins, count * 4, managed=True)
    want = ex.allreduce(orc.F32, orc.SUM, ins, bed.plan(COLL_ALLREDUCE, count, orc.F32))
    for o in outs:
        assert ex.check(orc.F32, o, want, exact=True) is None


def test_unaligned_user_pointers_and_guards(bed):
    """sendbuff / recvbuff aligned to the element only (not to 16 bytes), every collective and schedule; the test bed
    also checks that no byte outside recvbuff changes (tools/fuzz_collectives.py is the long, randomised version)."""
    n = bed.nranks
    for algo in ("Direct", "Ring"):
        bed.init_comms({"NCCL_ALGO": algo, "NCCL_PROTO": "Simple"})
        for dt, op, count, pad, in_place in ((orc.F32, orc.SUM, 70001, (4, 12), False), (orc.BF16, orc.AVG, 33333, (2, 6), False),
                                             (orc.I8, orc.MAX, 100003, (1, 3), True), (orc.F64, orc.PROD, 9001, (8, 8), True)):
            e = _es(dt)
            ins = [orc.fill_pattern(dt, count, r) for r in range(n)]
            outs = bed.run(COLL_ALLREDUCE, dt, op, count, ins, count * e, in_place=in_place, pad=pad)
            want = ex.allreduce(dt, op, ins, bed.plan(COLL_ALLREDUCE, count, dt))
            for r in range(n):
                assert ex.check(dt, outs[r], want, exact=True) is None, (algo, orc.DT_NAMES[dt], r)
            c2 = count // n
            ins = [orc.fill_pattern(dt, c2 * n, r) for r in range(n)]
            outs = bed.run(COLL_REDUCESCATTER, dt, op, c2, ins, c2 * e, in_place=in_place, pad=pad)
            plan = bed.plan(COLL_REDUCESCATTER, c2, dt)
            for r in range(n):
                assert ex.check(dt, outs[r], ex.reduce_scatter(dt, op, ins, r, plan), exact=True) is None, (algo, orc.DT_NAMES[dt], r)
            ins = [orc.fill_pattern(dt, c2, r) for r in range(n)]
            outs = bed.run(COLL_ALLGATHER, dt, orc.SUM, c2, ins, c2 * n * e, in_place=in_place, pad=pad)
            for o in outs:
                assert ex.check(dt, o, orc.all_gather_expected(ins), exact=True) is None, (algo, orc.DT_NAMES[dt])
    bed.init_comms({"NCCL_ALGO": None, "NCCL_PROTO": None})


def test_back_to_back_and_reinit(bed):
    """Step counters persist across launches; communicators can be destroyed and recreated."""
    bed.init_comms({"NCCL_ALGO": None})
    n, count = bed.nranks, 200000
    ins = [orc.fill_pattern(orc.I32, count, r) for r in range(n)]
    want = orc.host_fold(orc.I32, orc.SUM, ins)
    outs = bed.run(COLL_ALLREDUCE, orc.I32, orc.SUM, count, ins, count * 4, iters=5)
    for o in outs:
        assert np.array_equal(o.view(np.int32), want)
    bed.init_comms({"NCCL_ALGO": None})
    outs = bed.run(COLL_ALLREDUCE, orc.I32, orc.SUM, count, ins, count * 4)
    for o in outs:
        assert np.array_equal(o.view(np.int32), want)


def test_two_communicators_side_by_side(bed):
    """Two communicators over the same ranks keep separate windows and step counters: launches alternating
    between them on one stream each give the rank-order fold (small: low-latency protocol; large: FIFO)."""
    bed.init_comms({"NCCL_ALGO": None, "NCCL_PROTO": None})
    n = bed.nranks
    for count in (1250, 300001):
        a = [orc.fill_pattern(orc.I32, count, r) for r in range(n)]
        b = [orc.fill_pattern(orc.I32, count, 50 + r) for r in range(n)]
        want_a, want_b = orc.host_fold(orc.I32, orc.SUM, a), orc.host_fold(orc.I32, orc.SUM, b)
        for r, outs in enumerate(bed.run_two_comms(orc.I32, orc.SUM, count, a, b)):
            for k, want in enumerate((want_a, want_b, want_b, want_a)):
                assert np.array_equal(outs[k].view(np.int32), want), (count, r, k)
    # the first communicator still works after the second is gone
    _check_allreduce(bed, orc.F32, orc.SUM, 70001)


def test_low_latency_protocol(bed):
    """Small all-reduces take the one-shot low-latency protocol unless NCCL_PROTO=Simple; both give the
    rank-order fold (StandaloneTests.cpp:177-187 runs its known-answer case under LL and Simple)."""
    n = bed.nranks
    x = np.arange(1250, dtype=np.int32)
    for proto, want_proto in ((None, 1), ("Simple", 0), ("LL", 1)):
        bed.init_comms({"NCCL_ALGO": None, "NCCL_PROTO": proto})
        # (NCCL_PROTO=Simple: no line protocol; a small all-reduce is then one exchange through the FIFOs, proto 2)
        assert bed.plan(COLL_ALLREDUCE, 1250, orc.I32)["proto"] == (want_proto or 2)
        assert bed.plan(COLL_ALLREDUCE, 1 << 22, orc.F32)["proto"] in (0, 2, 3)  # mid-size: one or two exchanges, or the FIFO schedule
        assert bed.plan(COLL_ALLREDUCE, 1 << 26, orc.F32)["proto"] == 0          # large: always the three-phase FIFO protocol
        # two ranks, no NCCL_ALGO: large payloads go around the ring (one workgroup per channel walks all three phases), smaller
        # ones keep the all-pairs schedule and its one-exchange protocols; more ranks keep the all-pairs schedule throughout
        ring = 1 if n == 2 else 0
        assert bed.plan(COLL_ALLREDUCE, 6 << 20, orc.F32)["algo"] == ring         # 24 MiB
        assert bed.plan(COLL_ALLREDUCE, (6 << 20) - 1, orc.F32)["algo"] == 0
        assert bed.plan(COLL_REDUCESCATTER, 4 << 20, orc.F32)["algo"] == ring     # 16 MiB per rank
        assert bed.plan(COLL_ALLGATHER, 4 << 20, orc.F32)["algo"] == ring
        assert bed.plan(COLL_ALLGATHER, (4 << 20) - 1, orc.F32)["algo"] == 0
        if proto is None and n == 2:
            _check_allreduce(bed, orc.F32, orc.SUM, (6 << 20) + 1028)
        outs = bed.run(COLL_ALLREDUCE, orc.I32, orc.SUM, 1250, [x] * n, 1250 * 4, iters=7)
        for o in outs:
            assert np.array_equal(o.view(np.int32), n * x)
        for dt, op, count in ((orc.F32, orc.SUM, 384), (orc.F16, orc.MIN, 12888), (orc.I8, orc.PROD, 383), (orc.U8, orc.SUM, 2501),
                              (orc.BF16, orc.AVG, 65536 + 3), (orc.I64, orc.AVG, 1023), (orc.F64, orc.SUM, 32768), (orc.F32, orc.SUM, 65536)):
            _check_allreduce(bed, dt, op, count)
        _check_allreduce(bed, orc.I32, orc.PROD, 384, in_place=True)
        # reduce-scatter and all-gather take the same protocol up to 256 KiB per rank (unaligned blocks included)
        assert bed.plan(COLL_REDUCESCATTER, 1000, orc.F32)["proto"] == (want_proto or 2)
        assert bed.plan(COLL_ALLGATHER, 1000, orc.F32)["proto"] == (want_proto or 2)
        assert bed.plan(COLL_ALLGATHER, 262145, orc.F32)["proto"] in (0, 2)      # 1 MiB + 4 B per rank: not the line protocol
        assert bed.plan(COLL_ALLGATHER, 1 << 21, orc.F32)["proto"] == 0          # 8 MiB per rank: the FIFO schedule
        for dt, op, count, in_place in ((orc.F32, orc.SUM, 1000, False), (orc.I32, orc.PROD, 5423, True), (orc.U8, orc.MIN, 246, True),
                                        (orc.BF16, orc.AVG, 10481, False), (orc.F64, orc.MAX, 32768, False), (orc.I64, orc.AVG, 1023, False)):
            ins = [orc.fill_pattern(dt, count * n, r) for r in range(n)]
            outs = bed.run(COLL_REDUCESCATTER, dt, op, count, ins, count * _es(dt), in_place=in_place, iters=1)
            plan = bed.plan(COLL_REDUCESCATTER, count, dt)
            for r in range(n):
                assert ex.check(dt, outs[r], ex.reduce_scatter(dt, op, ins, r, plan), exact=True) is None, (dt, op, count, r)
        for dt, count, in_place in ((orc.F16, 1001, False), (orc.I32, 264, True), (orc.I8, 958, True), (orc.F64, 586, False), (orc.U8, 2500, False)):
            ins = [orc.fill_pattern(dt, count, r) for r in range(n)]
            outs = bed.run(COLL_ALLGATHER, dt, orc.SUM, count, ins, count * n * _es(dt), in_place=in_place, iters=1 if in_place else 3)
            for o in outs:
                assert ex.check(dt, o, orc.all_gather_expected(ins), exact=True) is None, (dt, count)
    # NCCL_PROTO=LL asked for by name together with NCCL_ALGO=Tree|Ring (the reference's LL tree / LL ring,
    # src/device/all_reduce.h:17-199,358-551): small all-reduces run the line protocol ALONG THE CHAIN / AROUND THE RING,
    # folded in that schedule's order like its FIFO form; larger ones keep the FIFOs
    for algo in ("Tree", "Ring"):
        bed.init_comms({"NCCL_ALGO": algo, "NCCL_PROTO": "LL"})
        plan = bed.plan(COLL_ALLREDUCE, 1250, orc.I32)
        assert plan["proto"] == 1 and plan["algo"] == ALGOS[algo], plan
        assert bed.plan(COLL_ALLREDUCE, 1 << 22, orc.F32)["algo"] == ALGOS[algo]
        assert bed.plan(COLL_ALLREDUCE, 1 << 22, orc.F32)["proto"] == 0
        _check_allreduce(bed, orc.I32, orc.SUM, 1250)
        _check_allreduce(bed, orc.BF16, orc.SUM, 40001)
        _check_allreduce(bed, orc.F32, orc.SUM, 1 << 17)                 # 512 KiB: the largest payload of the chain's lines
        _check_allreduce(bed, orc.F32, orc.SUM, (1 << 17) + 1)           # one element more: the named schedule's FIFOs
        _check_allreduce(bed, orc.F16, orc.AVG, 70001)
        _check_allreduce(bed, orc.I8, orc.AVG, 30003, in_place=True)
        _check_allreduce(bed, orc.F8E4M3, orc.MAX, 12888)
        # a user pre-multiply by scalars that are not powers of two: the product and the sum behind it round separately
        # (a fused multiply-add would differ in the last bit), folded in the schedule's order
        count = 30011
        ins = [orc.fill_pattern(orc.F32, count, r) for r in range(n)]
        scal = [np.float32(1.0 / (r + 3)) for r in range(n)]
        outs = bed.run(COLL_ALLREDUCE, orc.F32, orc.SUM, count, ins, count * 4, premul=[s_.tobytes() for s_ in scal])
        want = ex.allreduce(orc.F32, orc.SUM, [a * s_ for a, s_ in zip(ins, scal)], bed.plan(COLL_ALLREDUCE, count, orc.F32))
        for r in range(n):
            assert ex.check(orc.F32, outs[r], want) is None, (algo, r, ex.check(orc.F32, outs[r], want))
        # launches of the chain's lines and of the one-hop lines share buffers and launch counters: alternate them
        for count in (5000, 3, 70000):
            _check_allreduce(bed, orc.F32, orc.SUM, count)
            ins = [orc.fill_pattern(orc.I32, count * n, r) for r in range(n)]
            outs = bed.run(COLL_REDUCESCATTER, orc.I32, orc.SUM, count, ins, count * 4)
            rs_plan = bed.plan(COLL_REDUCESCATTER, count, orc.I32)
            for r in range(n):
                assert ex.check(orc.I32, outs[r], ex.reduce_scatter(orc.I32, orc.SUM, ins, r, rs_plan)) is None, (algo, count, r, rs_plan)
    bed.init_comms({"NCCL_ALGO": None, "NCCL_PROTO": None})


def test_hip_graph_capture_and_replay(bed):
    """Collectives captured into a hipGraph and replayed (the reference's graph-mode cases, e.g.
    AllReduceTests.cpp:29-46): step counters live in device memory, so a replay continues the FIFOs."""
    bed.init_comms({"NCCL_ALGO": None})
    n = bed.nranks
    count = 12888
    for dt, op in ((orc.F16, orc.MIN), (orc.F64, orc.MIN), (orc.I32, orc.MAX)):
        ins = [orc.fill_pattern(dt, count, r) for r in range(n)]
        outs = bed.run(COLL_ALLREDUCE, dt, op, count, ins, count * _es(dt), graph=True)
        want = orc.host_fold(dt, op, ins)
        for o in outs:
            assert ex.check(dt, o, want, exact=True) is None
    ins = [orc.fill_pattern(orc.BF16, 586, r) for r in range(n)]
    outs = bed.run(COLL_ALLGATHER, orc.BF16, orc.SUM, 586, ins, 586 * n * 2, graph=True)
    for o in outs:
        assert ex.check(orc.BF16, o, orc.all_gather_expected(ins), exact=True) is None
    count = 104857
    ins = [orc.fill_pattern(orc.F64, count * n, r) for r in range(n)]
    outs = bed.run(COLL_REDUCESCATTER, orc.F64, orc.MAX, count, ins, count * 8, graph=True)
    for r in range(n):
        assert ex.check(orc.F64, outs[r], orc.reduce_scatter_expected(orc.F64, orc.MAX, ins, r), exact=True) is None
    # a mid-size all-reduce (one exchange): its launch parity comes from device memory, so replays alternate the buffers too
    count = 300001
    assert bed.plan(COLL_ALLREDUCE, count, orc.F32)["proto"] == 2
    ins = [orc.fill_pattern(orc.F32, count, r) for r in range(n)]
    outs = bed.run(COLL_ALLREDUCE, orc.F32, orc.SUM, count, ins, count * 4, graph=True)
    for o in outs:
        assert ex.check(orc.F32, o, orc.device_fold(orc.F32, orc.SUM, ins), exact=True) is None
    count = 1001                      # small enough for the low-latency protocol: its line flags survive a replay too
    ins = [orc.fill_pattern(orc.I32, count * n, r) for r in range(n)]
    outs = bed.run(COLL_REDUCESCATTER, orc.I32, orc.SUM, count, ins, count * 4, graph=True)
    for r in range(n):
        assert ex.check(orc.I32, outs[r], orc.reduce_scatter_expected(orc.I32, orc.SUM, ins, r), exact=True) is None
    # rooted low-latency launches replayed: their presence lines carry the replay's epoch as well
    ins = [orc.fill_pattern(orc.I32, count, 7 if r == n - 1 else 100 + r) for r in range(n)]
    outs = bed.run(COLL_BROADCAST, orc.I32, orc.SUM, count, ins, count * 4, root=n - 1, graph=True)
    for r in range(n):
        assert ex.check(orc.I32, outs[r], ins[n - 1], exact=True) is None
    ins = [orc.fill_pattern(orc.F32, count, r) for r in range(n)]
    outs = bed.run(COLL_REDUCE, orc.F32, orc.SUM, count, ins, count * 4, root=0, graph=True)
    assert ex.check(orc.F32, outs[0], orc.device_fold(orc.F32, orc.SUM, ins), exact=True) is None
    # the three-phase FIFO schedule replayed, several rounds per channel, in both role layouts
    count = (1 << 20) + 1028
    ins = [orc.fill_pattern(orc.F32, count, r) for r in range(n)]
    for fused in (None, "1"):
        bed.init_comms({"NCCL_ALGO": "Direct", "RCCL_AMD_ONESHOT_MAX_BYTES": "0", "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": "0", "RCCL_AMD_FUSED_ROLES": fused})
        plan = bed.plan(COLL_ALLREDUCE, count, orc.F32)
        assert plan["proto"] == 0
        outs = bed.run(COLL_ALLREDUCE, orc.F32, orc.AVG, count, ins, count * 4, graph=True)
        for o in outs:
            assert ex.check(orc.F32, o, ex.allreduce(orc.F32, orc.AVG, ins, plan), exact=True) is None
    bed.init_comms({"NCCL_ALGO": None})


def test_registered_receive_buffers_are_written_directly(bed):
    """ncclCommRegister (src/register.cc:154-185): with the receive buffer registered, the other ranks write their
    finished all-reduce pieces straight into it (no broadcast-phase FIFO traffic, no gather copy); results are the
    same bits as through the FIFO.  Mixed cases: only some ranks registered, in place, after deregistration."""
    bed.init_comms({"NCCL_ALGO": None, "RCCL_AMD_ONESHOT_MAX_BYTES": "0", "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": "0"})    # the three-phase schedule: it has the gather phase
    n = bed.nranks
    count = (1 << 20) + 24          # 4 MiB fp32: the FIFO protocol, several rounds on every channel
    ins = [orc.fill_pattern(orc.F32, count, r) for r in range(n)]
    plan = bed.plan(COLL_ALLREDUCE, count, orc.F32)
    assert plan["proto"] == 0 and plan["algo"] == ex.ALGO_DIRECT
    want = ex.allreduce(orc.F32, orc.SUM, ins, plan)
    # every rank registered: all pieces go direct, on the first launch already and on the second
    for outs, stats in bed.run_registered(orc.F32, orc.SUM, count, ins, runs=2):
        for r in range(n):
            assert ex.check(orc.F32, outs[r], want, exact=True) is None, r
            assert stats[r]["direct_pieces"] > 0 and stats[r]["fifo_pieces"] == 0, (r, stats[r])
    # captured into a hipGraph after the first (eager) launch mapped the registrations: replays write directly too
    runs = bed.run_registered(orc.F32, orc.SUM, count, ins, runs=2, graph=True)
    for outs, stats in runs:
        for r in range(n):
            assert ex.check(orc.F32, outs[r], want, exact=True) is None, r
            assert stats[r]["direct_pieces"] > 0 and stats[r]["fifo_pieces"] == 0, (r, stats[r])
    # only rank 0 registered: every other rank writes one of its n-1 pieces per round directly
    (outs, stats), = bed.run_registered(orc.F32, orc.SUM, count, ins, register_ranks=[0])
    for r in range(n):
        assert ex.check(orc.F32, outs[r], want, exact=True) is None, r
    assert stats[0]["direct_pieces"] == 0
    for r in range(1, n):
        assert stats[r]["direct_pieces"] > 0 and stats[r]["fifo_pieces"] == (n - 2) * stats[r]["direct_pieces"], (r, stats[r])
    # in place, ragged bf16 count, average
    c2 = 3 * 1024 * 1024 + 1000
    ins2 = [orc.fill_pattern(orc.BF16, c2, r) for r in range(n)]
    want2 = ex.allreduce(orc.BF16, orc.AVG, ins2, bed.plan(COLL_ALLREDUCE, c2, orc.BF16))
    (outs, stats), = bed.run_registered(orc.BF16, orc.AVG, c2, ins2, in_place=True)
    for r in range(n):
        assert ex.check(orc.BF16, outs[r], want2, exact=True) is None, r
        assert stats[r]["fifo_pieces"] == 0, (r, stats[r])
    # nothing registered any more: back to the FIFO, same bits
    (outs, stats), = bed.run_registered(orc.F32, orc.SUM, count, ins, register_ranks=[])
    for r in range(n):
        assert ex.check(orc.F32, outs[r], want, exact=True) is None, r
        assert stats[r]["direct_pieces"] == 0 and stats[r]["fifo_pieces"] > 0, (r, stats[r])
    # memory-type rule: a registered receive buffer in ordinary (coarse-grained) hipMalloc memory is cached by its
    # owner's kernels, so no peer may store into it: the registration stays local and every piece takes the FIFO
    for outs, stats in bed.run_registered(orc.F32, orc.SUM, count, ins, runs=2, fine=False):
        for r in range(n):
            assert ex.check(orc.F32, outs[r], want, exact=True) is None, r
            assert stats[r]["direct_pieces"] == 0 and stats[r]["fifo_pieces"] > 0, (r, stats[r])
    # the same with one workgroup per channel walking all three roles: all registered, only rank 0, none
    bed.init_comms({"NCCL_ALGO": None, "RCCL_AMD_ONESHOT_MAX_BYTES": "0", "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": "0", "RCCL_AMD_FUSED_ROLES": "1"})
    for outs, stats in bed.run_registered(orc.F32, orc.SUM, count, ins, runs=2):
        for r in range(n):
            assert ex.check(orc.F32, outs[r], want, exact=True) is None, r
            assert stats[r]["direct_pieces"] > 0 and stats[r]["fifo_pieces"] == 0, (r, stats[r])
    (outs, stats), = bed.run_registered(orc.F32, orc.SUM, count, ins, register_ranks=[0])
    for r in range(n):
        assert ex.check(orc.F32, outs[r], want, exact=True) is None, r
    assert stats[0]["direct_pieces"] == 0 and all(stats[r]["direct_pieces"] > 0 for r in range(1, n))
    (outs, stats), = bed.run_registered(orc.F32, orc.SUM, count, ins, register_ranks=[])
    for r in range(n):
        assert ex.check(orc.F32, outs[r], want, exact=True) is None, r
        assert stats[r]["direct_pieces"] == 0 and stats[r]["fifo_pieces"] > 0, (r, stats[r])
    bed.init_comms({"NCCL_ALGO": None})


def test_registered_all_gather_writes_blocks_directly(bed):
    """All-gather into registered receive buffers: blocks of 4 MiB and more are written by their owners straight into
    every registered recvbuff (no FIFO, no gather copy); smaller blocks, unregistered ranks and a ragged last chunk
    give the same bytes through the FIFOs."""
    bed.init_comms({"NCCL_ALGO": None})
    n = bed.nranks
    count = (4 << 20) + 4099                  # bytes per rank: over the announce threshold, ragged tail
    ins = [orc.fill_pattern(orc.U8, count, r) for r in range(n)]
    want = orc.all_gather_expected(ins)
    runs = bed.run_registered(orc.U8, orc.SUM, count, ins, runs=2, coll=COLL_ALLGATHER)
    for outs, stats in runs:
        for r in range(n):
            assert ex.check(orc.U8, outs[r], want, exact=True) is None, r
            assert stats[r]["direct_pieces"] > 0 and stats[r]["fifo_pieces"] == 0, (r, stats[r])
    # only the last rank registered: each other rank writes that one block directly, the rest through the FIFOs
    (outs, stats), = bed.run_registered(orc.U8, orc.SUM, count, ins, register_ranks=[n - 1], coll=COLL_ALLGATHER)
    for r in range(n):
        assert ex.check(orc.U8, outs[r], want, exact=True) is None, r
    assert stats[n - 1]["direct_pieces"] == 0
    for r in range(n - 1):
        assert stats[r]["direct_pieces"] > 0 and stats[r]["fifo_pieces"] == (n - 2) * stats[r]["direct_pieces"], (r, stats[r])
    # in place (sendbuff is this rank's block of the registered recvbuff), 2-byte elements
    c2 = (2 << 20) + 77
    ins2 = [orc.fill_pattern(orc.BF16, c2, r) for r in range(n)]
    (outs, stats), = bed.run_registered(orc.BF16, orc.SUM, c2, ins2, in_place=True, coll=COLL_ALLGATHER)
    for r in range(n):
        assert ex.check(orc.BF16, outs[r], orc.all_gather_expected(ins2), exact=True) is None, r
        assert stats[r]["fifo_pieces"] == 0, (r, stats[r])
    # below the threshold nothing is announced: registered or not, the FIFOs carry it
    small = [orc.fill_pattern(orc.U8, 1 << 20, r) for r in range(n)]
    (outs, stats), = bed.run_registered(orc.U8, orc.SUM, 1 << 20, small, coll=COLL_ALLGATHER)
    for r in range(n):
        assert ex.check(orc.U8, outs[r], orc.all_gather_expected(small), exact=True) is None, r
        assert stats[r]["direct_pieces"] == 0, (r, stats[r])
    # nothing registered: the announce round carries zeros and every block takes the FIFO
    (outs, stats), = bed.run_registered(orc.U8, orc.SUM, count, ins, register_ranks=[], coll=COLL_ALLGATHER)
    for r in range(n):
        assert ex.check(orc.U8, outs[r], want, exact=True) is None, r
        assert stats[r]["direct_pieces"] == 0 and stats[r]["fifo_pieces"] > 0, (r, stats[r])
    # registered, but ordinary hipMalloc memory: same as not registered (memory-type rule)
    (outs, stats), = bed.run_registered(orc.U8, orc.SUM, count, ins, coll=COLL_ALLGATHER, fine=False)
    for r in range(n):
        assert ex.check(orc.U8, outs[r], want, exact=True) is None, r
        assert stats[r]["direct_pieces"] == 0 and stats[r]["fifo_pieces"] > 0, (r, stats[r])


def test_host_staged_tcp_transport(bed):
    """NCCL_P2P_DISABLE=1: every hop goes GPU -> pinned host -> loopback TCP -> pinned host -> GPU
    (the reference's socket transport, src/transport/net.cc + net_socket.cc).  Same kernels, same results.
    With 8 ranks (the rank count the host-staged configuration is quoted at) the payloads are the reference's two
    fp32-sum sizes and its known-answer case, on one or two channels; 2 and 4 ranks run the longer list."""
    n = bed.nranks
    try:
        bed.init_comms({"NCCL_P2P_DISABLE": "1", "RCCL_AMD_NCHANNELS": "2", "NCCL_ALGO": None})
        if n > 4:
            for dt, op, count in ((orc.F32, orc.SUM, 393216), (orc.I32, orc.SUM, 1250), (orc.F32, orc.SUM, 384)):
                _check_allreduce(bed, dt, op, count)
            bed.init_comms({"NCCL_P2P_DISABLE": "1", "RCCL_AMD_NCHANNELS": "1", "NCCL_ALGO": None})
            _check_allreduce(bed, orc.F32, orc.SUM, 393216)
            return
        for dt, op, count in ((orc.F32, orc.SUM, 393216), (orc.I32, orc.SUM, 1250), (orc.BF16, orc.SUM, 100003),
                              (orc.F32, orc.AVG, 70001)):
            _check_allreduce(bed, dt, op, count)
        count = 54235
        ins = [orc.fill_pattern(orc.I32, count * n, r) for r in range(n)]
        outs = bed.run(COLL_REDUCESCATTER, orc.I32, orc.PROD, count, ins, count * 4)
        for r in range(n):
            assert np.array_equal(outs[r].view(np.int32), orc.reduce_scatter_expected(orc.I32, orc.PROD, ins, r))
        ins = [orc.fill_pattern(orc.U8, 1039203, r) for r in range(n)]
        outs = bed.run(COLL_ALLGATHER, orc.U8, orc.SUM, 1039203, ins, 1039203 * n)
        for r in range(n):
            assert np.array_equal(outs[r], orc.all_gather_expected(ins))
        bed.init_comms({"NCCL_P2P_DISABLE": "1", "RCCL_AMD_NCHANNELS": "2", "NCCL_ALGO": "Ring"})
        _check_allreduce(bed, orc.F32, orc.SUM, 393216)
    finally:
        bed.destroy_comms()
        bed.init_comms({"NCCL_P2P_DISABLE": None, "RCCL_AMD_NCHANNELS": bed.base_env["RCCL_AMD_NCHANNELS"], "NCCL_ALGO": None})


def test_comm_split_across_processes(bed):
    """ncclCommSplit with every rank in its own process (the reference's cases, test/StandaloneTests.cpp:18-153, in its
    multi-process mode): colours by parity, one colour with reversed keys, upper half NCCL_SPLIT_NOCOLOR.  Every new
    communicator then runs an all-reduce whose result names exactly its members."""
    n = bed.nranks
    bed.init_comms({"NCCL_ALGO": None, "NCCL_PROTO": None})
    cases = [([r % 2 for r in range(n)], list(range(n))),
             ([0] * n, [n - r for r in range(n)]),
             ([0 if r < n // 2 else -1 for r in range(n)], list(range(n)))]
    count = 5000
    for colors, keys in cases:
        res = bed.split(colors, keys, count)
        for r in range(n):
            if colors[r] < 0:
                assert res[r] is None
                continue
            members = sorted((q for q in range(n) if colors[q] == colors[r]), key=lambda q: (keys[q], q))
            size, new_rank, err, out = res[r]
            assert (size, new_rank, err) == (len(members), members.index(r), 0), (r, colors, keys, size, new_rank, err)
            want = len(members) * np.arange(count, dtype=np.int64) + 1000 * sum(members)
            assert np.array_equal(out.astype(np.int64), want), (r, colors, keys)
    # ncclConfig_t::maxCTAs (src/init.cc:2091-2100) bounds the channels, and with them the window, of the new communicators
    colors, keys = [0] * n, list(range(n))
    for r, x in enumerate(bed.split(colors, keys, count, max_ctas=3)):
        size, new_rank, err, out, channels = x
        assert (size, new_rank, err) == (n, r, 0) and 1 <= channels <= 3, (r, x[:3], channels)
        assert np.array_equal(out.astype(np.int64), n * np.arange(count, dtype=np.int64) + 1000 * sum(range(n))), r
    # the parent is unharmed
    _check_allreduce(bed, orc.F32, orc.SUM, 70000)


def test_non_blocking_creation(bed):
    """config.blocking = 0 (test/NonBlockingTests.cpp, test/common/TestBedChild.cpp:250-258): ncclCommInitRankConfig
    returns at once, ncclCommGetAsyncError says ncclInProgress until the creation has ended, and the communicator
    then behaves like any other."""
    bed.init_comms({"NCCL_ALGO": None, "NCCL_PROTO": None}, blocking=False)
    _check_allreduce(bed, orc.F32, orc.SUM, 1048576)
    _check_allreduce(bed, orc.I32, orc.SUM, 1250)
    bed.init_comms()


def test_missing_peer_is_detected_not_hung():
    """Failure detection (reference: abort flag polled by every device spin, prims_simple.h:82-92, surfaced by
    ncclCommGetAsyncError, init.cc:2842-2849).  A rank that never joins a collective must not hang the others:
    their device-side waits expire, the error is reported asynchronously, later calls fail, ncclCommAbort cleans up."""
    tb = TestBed(2, devices=_devices(), env={"RCCL_AMD_SPIN_TIMEOUT_MS": "1500"})
    try:
        tb.init_comms()
        count = 1 << 20
        ins = [orc.fill_pattern(orc.F32, count, r) for r in range(2)]
        errs = tb.run(COLL_ALLREDUCE, orc.F32, orc.SUM, count, ins, count * 4, skip_ranks=(1,), expect_errors=True)
        assert errs[0] == 6 and errs[1] == 0                     # ncclRemoteError on the rank that waited
        assert tb.command(0, "try_coll") == 6                    # the communicator stays failed
        tb.command(0, "abort")
        tb.command(1, "destroy")
        tb.inited = False
        # fresh communicators on the same processes work again
        tb.init_comms()
        outs = tb.run(COLL_ALLREDUCE, orc.F32, orc.SUM, count, ins, count * 4)
        want = orc.host_fold(orc.F32, orc.SUM, ins)
        for o in outs:
            assert np.array_equal(o.view(np.float32), want)
    finally:
        tb.close()
