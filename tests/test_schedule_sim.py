"""The FIFO protocol and every schedule built on it, run on the host model of tests/schedule_sim.py with the library's
own geometry (planner) and rings (ring search): no credit deadlock, every slot holds the piece its reader expects,
every output element is written exactly once with every rank's contribution.

Covers what one GPU cannot run: 8 ranks at the DEFAULT geometry (56 channels, 2 x 768 KiB slots, 168 workgroups per
rank) for the benchmark configurations at their stated sizes (the reference's step accounting for the same walk:
src/enqueue.cc:238-243; its channel count on this topology: src/graph/connect.cc:638-660), several launches in a row on
one set of FIFO counters, rank counts 2..8 and 9..16 (ring only), ragged sizes.
"""
import pytest

import rccl_amd as ra
import schedule_sim as sim

F32, BF16, U8 = ra.ncclFloat32, ra.ncclBfloat16, ra.ncclUint8
SLOT = 768 << 10


def _rings(n):
    adj = [[0 if i == j else 1 for j in range(n)] for i in range(n)]
    return ra.topo_rings(n, adj) or [list(range(n))]


def _channels(n):
    return min(56, max(32, 8 * (n - 1)))          # readTunables in comm.cc


def _run(net, n, coll, algo, count, dt, rings, channels=None, slot=SLOT):
    es = ra.TYPE_SIZE[dt]
    plan = ra.plan_for_config(n, coll, count, dt, algo=algo, max_channels=channels or _channels(n), slot_bytes=slot)
    assert plan["algo"] == algo, plan
    assert plan["chunkElems"] * es <= slot, plan
    out, steps = sim.launch(net, coll, algo, plan, count, es, rings)
    everyone = frozenset(range(n))
    if coll == sim.ALLGATHER:
        nb = count * es
        out.check(n * nb, lambda off: frozenset({off // nb}))
    else:
        out.check(count, everyone)
    assert net.max_depth <= net.nslots
    return plan, steps


def test_default_geometry_eight_ranks_full_size():
    """8 ranks, 56 channels: all-reduce 1 GiB fp32 on the all-pairs and ring schedules, 1 GiB bf16 on the chain,
    reduce-scatter and all-gather of 256 MiB per rank; then the same FIFOs carry a second, ragged launch of each."""
    n = 8
    rings = _rings(n)
    assert len(rings) == 14
    net = sim.Net(n)
    plan, _ = _run(net, n, sim.ALLREDUCE, sim.DIRECT, 1 << 28, F32, rings)
    assert plan["nChannels"] == 56 and plan["grid"] == 168, plan
    _run(net, n, sim.ALLREDUCE, sim.RING, 1 << 28, F32, rings)
    _run(net, n, sim.ALLREDUCE, sim.TREE, 1 << 29, BF16, rings)
    _run(net, n, sim.REDUCESCATTER, sim.DIRECT, 1 << 26, F32, rings)
    _run(net, n, sim.ALLGATHER, sim.DIRECT, 1 << 26, F32, rings)
    _run(net, n, sim.REDUCESCATTER, sim.RING, 1 << 26, F32, rings)
    _run(net, n, sim.ALLGATHER, sim.RING, 1 << 26, F32, rings)
    # the counters have moved on by odd numbers of slots on some FIFOs; ragged sizes next
    for count in ((1 << 28) - 12345, 100003, 5 * 1024 * 1024 + 123, 7):
        _run(net, n, sim.ALLREDUCE, sim.DIRECT, count, F32, rings)
        _run(net, n, sim.ALLREDUCE, sim.RING, count, F32, rings)
        _run(net, n, sim.ALLREDUCE, sim.TREE, count, BF16, rings)
        _run(net, n, sim.REDUCESCATTER, sim.DIRECT, max(1, count // n), F32, rings)
        _run(net, n, sim.ALLGATHER, sim.DIRECT, max(1, count // n), U8, rings)


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 7])
def test_every_rank_count_up_to_eight(n):
    rings = _rings(n)
    net = sim.Net(n)
    for count in (1 << 28, 393216, 1250, (3 << 20) + 5):
        for algo in (sim.DIRECT, sim.RING, sim.TREE):
            plan, _ = _run(net, n, sim.ALLREDUCE, algo, count, F32, rings)
            if n == 2 and algo == sim.DIRECT and plan["nChannels"] % 2 == 0 and plan["nChannels"] >= 2:
                assert plan["grid"] == 2 * plan["nChannels"]          # scatter / gather workgroups serve two channels
        for algo in (sim.DIRECT, sim.RING):
            _run(net, n, sim.REDUCESCATTER, algo, max(1, count // n), F32, rings)
            _run(net, n, sim.ALLGATHER, algo, max(1, count // n), F32, rings)
    # the 1 GiB all-gather blocks are above the announce threshold, the small ones below: both kinds share FIFOs
    _run(net, n, sim.ALLGATHER, sim.DIRECT, (4 << 20) + 4099, U8, rings)
    _run(net, n, sim.ALLGATHER, sim.DIRECT, 1 << 20, U8, rings)


@pytest.mark.parametrize("n", [9, 12, 16])
def test_ring_schedules_above_eight_ranks(n):
    """More than 8 ranks: the planner turns the all-pairs schedule into the ring (RA_MAX_SRCS sources per fold)."""
    rings = _rings(n)
    net = sim.Net(n)
    for count in (1 << 26, 100003):
        plan = ra.plan_for_config(n, sim.ALLREDUCE, count, F32, max_channels=56, slot_bytes=512 << 10)
        assert plan["algo"] == sim.RING
        _run(net, n, sim.ALLREDUCE, sim.RING, count, F32, rings, channels=56, slot=512 << 10)
        _run(net, n, sim.ALLREDUCE, sim.TREE, count, F32, rings, channels=56, slot=512 << 10)
        _run(net, n, sim.REDUCESCATTER, sim.RING, count // n, F32, rings, channels=56, slot=512 << 10)
        _run(net, n, sim.ALLGATHER, sim.RING, count // n, F32, rings, channels=56, slot=512 << 10)


def test_the_model_notices_a_credit_deadlock_and_a_misplaced_piece():
    """The checker itself: a FIFO one slot deep cannot carry the ring's two steps in flight... it can, slowly; but a ring
    whose last rank never forwards deadlocks, and a reducer reading the wrong round is caught."""
    n = 4
    rings = _rings(n)
    net = sim.Net(n)
    plan = ra.plan_for_config(n, sim.ALLREDUCE, 1 << 22, F32, algo=sim.RING, max_channels=8, slot_bytes=SLOT)
    out = sim.Outputs(n)
    progs = [sim.ring_allreduce(n, r, ch, plan, 1 << 22, rings, out) for r in range(n - 1) for ch in range(plan["nChannels"])]
    with pytest.raises(sim.Deadlock):
        sim.run(net, progs)                       # rank 3 never launched
    net = sim.Net(n)
    plan = ra.plan_for_config(n, sim.ALLREDUCE, 1 << 22, F32, algo=sim.DIRECT, max_channels=8, slot_bytes=SLOT)
    wrong = dict(plan, chunkElems=plan["chunkElems"] // 2)          # one rank's scatter walks another geometry
    out = sim.Outputs(n)
    progs = []
    for r in range(n):
        for ch in range(plan["nChannels"]):
            for role in range(3):
                progs.append(sim.direct_allreduce(n, r, role, [ch], wrong if (r == 1 and role == 0) else plan, 1 << 22, out))
    with pytest.raises(AssertionError):
        sim.run(net, progs)
        out.check(1 << 22, frozenset(range(n)))


def test_one_workgroup_per_channel_all_reduce():
    """RCCL_AMD_FUSED_ROLES=1: one workgroup per channel runs scatter of round k, reduce of round k - 1 and gather of round
    k - 2.  With the layout's two slots per FIFO it completes at every rank count and size, same FIFOs reused by a ragged
    second launch and by the three-role form; with ONE slot per FIFO the same sequence waits on itself (the scatter of
    round k needs the credit of the peers' reduce of round k - 1, which they run behind their own scatter of round k),
    and the model says so."""
    for n in (3, 4, 5, 8):
        rings = _rings(n)
        net = sim.Net(n)
        for count in (1 << 28, (1 << 24) + 12345, 70001, 1 << 20):
            for channels in (_channels(n), 112):
                plan = ra.plan_for_config(n, sim.ALLREDUCE, count, F32, algo=sim.DIRECT, max_channels=channels, slot_bytes=SLOT)
                plan = dict(plan, grid=plan["nChannels"])
                out, _ = sim.launch(net, sim.ALLREDUCE, sim.DIRECT, plan, count, 4, rings, one_workgroup=True)
                out.check(count, frozenset(range(n)))
                assert net.max_depth <= net.nslots
            _run(net, n, sim.ALLREDUCE, sim.DIRECT, count, F32, rings)         # the three-role form continues the same FIFOs
    net1 = sim.Net(4, nslots=1)
    plan = ra.plan_for_config(4, sim.ALLREDUCE, 1 << 24, F32, algo=sim.DIRECT, max_channels=32, slot_bytes=SLOT)
    with pytest.raises(sim.Deadlock):
        sim.launch(net1, sim.ALLREDUCE, sim.DIRECT, dict(plan, grid=plan["nChannels"]), 1 << 24, 4, _rings(4), one_workgroup=True)
