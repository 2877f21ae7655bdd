"""Host logic on CPU: the planner's geometry, the expected-value builder that mirrors it, and the
multi-process rendezvous (world_size 2 over gloo + the library's own bootstrap)."""
import multiprocessing as mp
import os

import numpy as np
import pytest

import expected as ex
import oracle as orc
import rccl_amd as ra


@pytest.mark.parametrize("n", [2, 4, 8])
@pytest.mark.parametrize("dt", [ra.ncclFloat32, ra.ncclBfloat16, ra.ncclUint8, ra.ncclFloat64])
@pytest.mark.parametrize("count", [1, 384, 1250, 393216, 5 * 1024 * 1024 + 123, 1 << 28])
def test_allreduce_plan_geometry(n, dt, count):
    es = ra.TYPE_SIZE[dt]
    p = ra.plan_for_config(n, 0, count, dt)
    unit = 16 // es
    assert p["algo"] == 0 and 1 <= p["nChannels"] <= 16
    assert p["chunkElems"] % unit == 0 and p["chunkElems"] * es <= 512 << 10        # one FIFO slot
    # all workgroups resident; with two ranks and an even channel count the scatter and gather workgroups serve two
    # channels each (a reducer step moves twice their bytes), so the launch has two workgroups per channel
    paired = n == 2 and p["nChannels"] % 2 == 0
    assert p["grid"] == (2 if paired else 3) * p["nChannels"] <= 256
    # every element belongs to exactly one (round, piece)
    chunk, last = p["chunkElems"], p["lastChunkElems"]
    nrounds, last_expected = ex._last_chunk(count, n, chunk, {4: orc.F32, 2: orc.BF16, 1: orc.U8, 8: orc.F64}[es])
    assert last == last_expected and last % unit == 0
    covered = (nrounds - 1) * n * chunk + min(count - (nrounds - 1) * n * chunk, n * last)
    assert covered == count


def test_plan_round_and_row_rules():
    """Chunks are whole 1 KiB rows once they exceed the minimum and whole tiles of the copy loops once they hold four of
    them; with more than two ranks a channel gets about four rounds, eight where four would need chunks above 256 KiB;
    with two ranks one round of the largest piece that fits a slot; payloads of more than a slot per channel take
    rounds of about equal size; small counts use only as many channels as have a full minimum chunk each."""
    f32 = ra.ncclFloat32
    # 1 GiB, 8 ranks, 56 channels: four rounds would need 599 KiB chunks -> eight rounds of 320 KiB (5 tiles of 64 KiB)
    big = ra.plan_for_config(8, 0, 1 << 28, f32, max_channels=56)
    assert big["nChannels"] == 56 and big["grid"] == 168
    assert big["chunkElems"] == 81920 and big["chunkElems"] * 4 % (64 << 10) == 0
    rounds = -(-(1 << 28) // (8 * 56 * big["chunkElems"]))
    assert rounds == 8
    # 1 GiB, 2 ranks, 32 channels: the slot bounds the chunk (512 KiB given here; 768 KiB is the library's default)
    assert ra.plan_for_config(2, 0, 1 << 28, f32, max_channels=32)["chunkElems"] == (512 << 10) // 4
    assert ra.plan_for_config(2, 0, 1 << 28, f32, max_channels=64, slot_bytes=768 << 10)["chunkElems"] == (768 << 10) // 4
    # 128 MiB, 2 ranks, 64 channels, 768 KiB slots: 1 MiB per channel and rank -> two rounds of 512 KiB, not 768 + 256
    assert ra.plan_for_config(2, 0, 1 << 25, f32, max_channels=64, slot_bytes=768 << 10)["chunkElems"] == (512 << 10) // 4
    # 16 MiB, 2 ranks: one round of 256 KiB per channel (with two ranks a step's cost is mostly fixed below a full slot)
    mid = ra.plan_for_config(2, 0, 1 << 22, f32, max_channels=32)
    assert mid["nChannels"] == 32 and mid["chunkElems"] == 65536
    # 16 MiB, 4 ranks: four rounds of 32 KiB stay (more ranks: a step moves n-1 pieces and pipelining pays)
    mid4 = ra.plan_for_config(4, 0, 1 << 22, f32, max_channels=32)
    assert mid4["nChannels"] == 32 and mid4["chunkElems"] == 8192
    # exactly 16 minimum chunks per rank pair -> 16 channels; one element more opens a 17th with a 4-element tail
    assert ra.plan_for_config(2, 0, 1 << 18, f32, max_channels=32)["nChannels"] == 16
    edge = ra.plan_for_config(2, 0, (1 << 18) + 1, f32, max_channels=32)
    assert edge["nChannels"] == 17 and edge["lastChunkElems"] == 4
    # a smaller slot bounds the chunk whatever the round rule asks for
    assert ra.plan_for_config(8, 0, 1 << 28, f32, max_channels=56, slot_bytes=256 << 10)["chunkElems"] * 4 <= 256 << 10
    with pytest.raises(ra.NcclError):
        ra.plan_for_config(8, 3, 100, f32)          # broadcast/reduce have no channel plan


def test_plan_algorithm_overrides():
    assert ra.plan_for_config(8, 0, 1 << 20, ra.ncclFloat32, algo=1)["grid"] == ra.plan_for_config(8, 0, 1 << 20, ra.ncclFloat32, algo=1)["nChannels"]
    t = ra.plan_for_config(8, 0, 1 << 20, ra.ncclFloat32, algo=2)
    assert t["algo"] == 2 and t["grid"] == 2 * t["nChannels"]
    # tree is an all-reduce schedule only: the other two fall back to the ring (as the reference does)
    assert ra.plan_for_config(8, 1, 1 << 20, ra.ncclFloat32, algo=2)["algo"] == 1
    # more than 8 ranks cannot use the 8-source all-pairs fold
    assert ra.plan_for_config(16, 0, 1 << 20, ra.ncclFloat32)["algo"] == 1


def test_protocol_choice_by_size():
    """Which protocol a collective takes (reference: the size switch of src/graph/tuning.cc:632-671): the one-hop line
    protocol up to 512 KiB, one exchange up to the limit a communicator of that size holds, the FIFO schedule above;
    NCCL_PROTO=Simple (allow_ll 0) removes the line protocol, NCCL_PROTO=LL named alone (2) extends it to ring / tree."""
    f32, AR, RS, AG, BC, RD = ra.ncclFloat32, 0, 1, 2, 3, 4
    K, M = 1 << 10, 1 << 20
    lim = {2: 64 * M, 4: 4 * M, 8: 2 * M}
    for n in (2, 4, 8):
        proto = lambda coll, nbytes, **kw: ra.plan_proto_for_config(n, coll, nbytes // 4, f32, one_shot_max_bytes=lim[n], **kw)[1]
        assert proto(AR, 4 * K) == 1 and proto(AR, 512 * K) == 1
        assert proto(AR, 512 * K + 4) == 2 and proto(AR, lim[n]) == 2
        assert proto(AR, lim[n] + 4) == 0 and proto(AR, 1 << 30) == 0
        assert proto(AR, 4 * K, allow_ll=0) == 2                          # no line protocol: one exchange from the start
        # reduce-scatter: the exchange at every size its window region holds; all-gather below 4 MiB per rank
        assert proto(RS, 256 * K) == 1 and proto(RS, M) == 2 and proto(RS, lim[n]) == 2 and proto(RS, lim[n] + 4) == 0
        assert proto(AG, 256 * K) == 1 and proto(AG, M) == 2 and proto(AG, 4 * M) == 0
        # rooted collectives: line protocol or FIFO
        assert proto(BC, 64 * K) == 1 and proto(BC, 2 * M) == 0 and proto(RD, 64 * K) == 1 and proto(RD, 2 * M) == 0
        # ring / tree: FIFO, unless NCCL_PROTO=LL was asked for by name: then all-reduces of up to 512 KiB run the line
        # protocol along the chain / around the ring (the plan keeps the schedule); the other collectives take the line
        # protocol on the all-pairs links, and the plan says so
        for algo in (1, 2):
            assert ra.plan_proto_for_config(n, AR, 1000, f32, algo=algo, one_shot_max_bytes=lim[n]) == (algo, 0)
            assert ra.plan_proto_for_config(n, AR, 1000, f32, algo=algo, allow_ll=2, one_shot_max_bytes=lim[n]) == (algo, 1)
            assert ra.plan_proto_for_config(n, AR, 1 << 24, f32, algo=algo, allow_ll=2, one_shot_max_bytes=lim[n]) == (algo, 0)
        assert ra.plan_proto_for_config(n, AR, 512 * K // 4, f32, algo=2, allow_ll=2, one_shot_max_bytes=lim[n]) == (2, 1)
        assert ra.plan_proto_for_config(n, AR, 512 * K // 4 + 1, f32, algo=2, allow_ll=2, one_shot_max_bytes=lim[n]) == (2, 0)
        assert ra.plan_proto_for_config(n, AR, 512 * K // 4, f32, algo=1, allow_ll=2, one_shot_max_bytes=lim[n]) == (1, 1)
        assert ra.plan_proto_for_config(n, AR, 512 * K // 4 + 1, f32, algo=1, allow_ll=2, one_shot_max_bytes=lim[n]) == (1, 0)
        assert ra.plan_proto_for_config(n, RS, 1000, f32, algo=2, allow_ll=2, one_shot_max_bytes=lim[n]) == (0, 1)   # no chain for these
    # two exchanges take over above the one-exchange limit, for more than two ranks
    for n in (4, 8):
        p3 = lambda nbytes: ra.plan_proto_for_config(n, AR, nbytes // 4, f32, one_shot_max_bytes=lim[n], two_exchange_max_bytes=64 * M)[1]
        assert p3(lim[n]) == 2 and p3(lim[n] + 4) == 3 and p3(64 * M) == 3 and p3(64 * M + 4) == 0
    assert ra.plan_proto_for_config(2, AR, 100 * M // 4, f32, one_shot_max_bytes=lim[2], two_exchange_max_bytes=128 * M)[1] == 0
    # without the exchange (limit 0, e.g. the host-staged transport) the line protocol keeps its own limits
    assert ra.plan_proto_for_config(2, AR, M // 4, f32)[1] == 1 and ra.plan_proto_for_config(8, AR, M // 4, f32)[1] == 0
    assert ra.plan_proto_for_config(2, AR, 2 * M // 4, f32)[1] == 0
    # more than 8 ranks: no all-pairs fold, so neither the exchange nor (by default) the line protocol
    assert ra.plan_proto_for_config(16, AR, 1000, f32, one_shot_max_bytes=2 * M) == (1, 0)


def test_expected_builder_orders():
    """tests/expected.py: direct == rank-order fold; ring/tree differ from it only within tolerance."""
    n, count = 4, 70001
    ins = [orc.fill_pattern(orc.F32, count, r) for r in range(n)]
    host = orc.host_fold(orc.F32, orc.SUM, ins)
    rings = ra.topo_rings(n, [[int(i != j) for j in range(n)] for i in range(n)])
    for algo in (0, 1, 2):
        p = ra.plan_for_config(n, 0, count, ra.ncclFloat32, algo=algo)
        want = ex.allreduce(orc.F32, orc.SUM, ins, p, rings)
        assert orc.first_mismatch(orc.F32, want, host) == -1
        if algo == 0:
            assert np.array_equal(want, host)
    i32 = [orc.fill_pattern(orc.I32, count, r) for r in range(n)]
    for algo in (0, 1, 2):
        p = ra.plan_for_config(n, 0, count, ra.ncclInt32, algo=algo)
        assert np.array_equal(ex.allreduce(orc.I32, orc.PROD, i32, p, rings), orc.host_fold(orc.I32, orc.PROD, i32))


def _gloo_rank(rank, world, port, q):
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        # what bench.py does: rank 0 creates the id, gloo carries it to the others
        uid = [ra.get_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        mine = bytes([rank + 1]) * 64
        got = ra.bootstrap_all_gather(uid[0], rank, world, mine)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, got))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))


@pytest.mark.parametrize("world", [2, 3])
def test_rendezvous_world_size_n_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + world + (os.getpid() % 200)
    ps = [ctx.Process(target=_gloo_rank, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=30)
    for r in range(world):
        assert res[r] == [bytes([k + 1]) * 64 for k in range(world)], res[r]


def test_window_stays_inside_its_budget():
    """The per-rank window (the reference's per-connection buffers, src/transport/p2p.cc:447-450) is fitted into
    1.5 GiB for every rank count and any channel count up to 64, well clear of the 2 GiB at which windows shared
    between processes are refused; the default geometry at 8 ranks is not touched by the fit."""
    MiB = 1 << 20
    for n in range(2, 17):
        for ch in (8, 16, 24, 32, 48, 56, 64):
            w = ra.window_bytes_for(n, ch)
            assert w["total"] <= 1536 * MiB, (n, ch, w)
            assert w["channels"] == ch, (n, ch, w)           # the fit never had to give up channels in this range
            assert w["fifo"] + w["exchange"] < w["total"]
    w = ra.window_bytes_for(8, 56)
    # 56 channels x 7 peers x 2 kinds x 2 slots x 768 KiB, two exchange buffers of 8 senders x 2 x (64 MiB / 8)
    assert w["fifo"] == 56 * 7 * 2 * 2 * 768 * 1024 and 256 * MiB <= w["exchange"] < 258 * MiB, w
    assert ra.window_bytes_for(8, 64)["exchange"] < 130 * MiB          # the exchange region gave way first
    assert ra.window_bytes_for(2, 32)["fifo"] == 32 * 1 * 2 * 2 * 768 * 1024


def test_rings_follow_the_links_that_exist():
    """Adjacency from link types and hop counts (hipExtGetLinkTypeAndHopCount; the reference: src/graph/xml.cc:780-791):
    every pair one xGMI hop -> all-pairs schedule stays the default; a pair that is not -> rings that avoid it."""
    n = 8
    xgmi = [[0 if i == j else 4 for j in range(n)] for i in range(n)]
    one = [[0 if i == j else 1 for j in range(n)] for i in range(n)]
    adj, all_direct = ra.topo_from_links(xgmi, one)
    assert all_direct and all(adj[i][j] == (i != j) for i in range(n) for j in range(n))
    # one direction of one pair goes over PCIe (type 2): the pair is out, both ways
    pcie = [row[:] for row in xgmi]
    pcie[2][5] = 2
    adj, all_direct = ra.topo_from_links(pcie, one)
    assert not all_direct and adj[2][5] == 0 and adj[5][2] == 0 and sum(map(sum, adj)) == n * (n - 1) - 2
    rings = ra.topo_rings(n, adj)
    assert rings
    for r in rings:
        assert sorted(r) == list(range(n))
        for i in range(n):
            assert adj[r[i]][r[(i + 1) % n]], (r, i)
    # two hops over xGMI is not a direct link either
    far = [row[:] for row in one]
    far[0][7] = far[7][0] = 2
    adj, all_direct = ra.topo_from_links(xgmi, far)
    assert not all_direct and adj[0][7] == 0
