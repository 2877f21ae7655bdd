"""A host-side model of the FIFO protocol and of the schedules built on it (test infrastructure, CPU only).

The collective kernels are sequential programs, one per (rank, role, channel) workgroup, whose only interaction is the
step/credit protocol of `Ctx::step` (rccl_amd/csrc/kernels_impl.h; the reference's waitPeer / postPeer,
src/device/prims_simple.h:111-182): a step waits until every FIFO it reads holds a published slot and every FIFO it
writes has a free one, moves data, then posts a credit for each slot consumed and a tail for each slot published.
FIFOs of `nSlots` slots with credits make this a process network with bounded channels, whose outcome (completion or
deadlock, and every value moved) does not depend on the order in which runnable workgroups are picked.  So ONE
execution of the model, with any fair pick, decides whether a geometry can deadlock, and tracking what every slot holds
shows whether every piece lands where it belongs exactly once.

The model restates the step sequence of every schedule from the kernels (all-pairs all-reduce with its two-rank role
pairing, reduce-scatter, all-gather with the announce round, ring all-reduce / reduce-scatter / all-gather, the chain)
and takes the geometry (channels, chunk, last chunk, grid) from the library's own planner and its rings from the
library's own ring search, so what it checks is the product's arithmetic, not a copy of it.

Data is symbolic: a value is (offset, elements, frozenset of contributing ranks).  Reads assert that a slot holds the
piece the reader expects from that sender; writes into a rank's output are collected and must tile it exactly once
with every rank contributing.
"""
from collections import defaultdict

ALLREDUCE, REDUCESCATTER, ALLGATHER = 0, 1, 2
DIRECT, RING, TREE = 0, 1, 2
AG_ANNOUNCE_BYTES = 4 << 20          # RA_AG_ANNOUNCE_BYTES


class Deadlock(AssertionError):
    pass


class Fifo:
    __slots__ = ("tail", "head", "slots")

    def __init__(self, nslots):
        self.tail = 0
        self.head = 0
        self.slots = [None] * nslots


class Net:
    """All FIFOs of one communicator; counters persist across launches as they do in device memory."""

    def __init__(self, n, nslots=2):
        self.n, self.nslots = n, nslots
        self.fifos = {}
        self.max_depth = 0

    def fifo(self, ch, sender, receiver, kind):
        key = (ch, sender, receiver, kind)
        f = self.fifos.get(key)
        if f is None:
            f = self.fifos[key] = Fifo(self.nslots)
        return f


class Step:
    """One protocol step of a workgroup of `rank` on channel `ch`: recv = [(peer, kind)], send = [(peer, kind)];
    body(read, write) with read(peer, kind) -> slot content, write(peer, kind, content)."""
    __slots__ = ("rank", "ch", "recv", "send", "body")

    def __init__(self, rank, ch, recv, send, body):
        self.rank, self.ch, self.recv, self.send, self.body = rank, ch, recv, send, body


def runnable(net, st):
    for p, k in st.recv:
        f = net.fifo(st.ch, p, st.rank, k)
        if f.tail < f.head + 1:
            return False
    for p, k in st.send:
        f = net.fifo(st.ch, st.rank, p, k)
        if f.tail + 1 - f.head > net.nslots:
            return False
    return True


def execute(net, st):
    def read(p, k):
        f = net.fifo(st.ch, p, st.rank, k)
        v = f.slots[f.head % net.nslots]
        assert v is not None, ("reading an empty slot", st.rank, st.ch, p, k)
        return v

    def write(p, k, content):
        f = net.fifo(st.ch, st.rank, p, k)
        f.slots[f.tail % net.nslots] = content

    st.body(read, write)
    for p, k in st.recv:
        f = net.fifo(st.ch, p, st.rank, k)
        f.slots[f.head % net.nslots] = None
        f.head += 1
    for p, k in st.send:
        f = net.fifo(st.ch, st.rank, p, k)
        f.tail += 1
        net.max_depth = max(net.max_depth, f.tail - f.head)


def run(net, programs):
    """programs: list of generators of Step.  Runs them to completion; raises Deadlock when none can move."""
    pending = []
    for g in programs:
        st = next(g, None)
        if st is not None:
            pending.append([g, st])
    steps = 0
    while pending:
        moved = False
        nxt = []
        for item in pending:
            g, st = item
            while st is not None and runnable(net, st):
                execute(net, st)
                steps += 1
                moved = True
                st = next(g, None)
            if st is not None:
                item[1] = st
                nxt.append(item)
        pending = nxt
        if pending and not moved:
            stuck = [(st.rank, st.ch, st.recv, st.send) for _, st in pending[:6]]
            raise Deadlock(f"{len(pending)} workgroups wait for one another, e.g. {stuck}")
    return steps


# ------------------------------------------------------------------ geometry (Rounds in kernels_impl.h)
class Rounds:
    def __init__(self, count, chunk, last_chunk, n):
        self.count, self.chunk, self.last, self.n = count, chunk, last_chunk, n
        self.round_elems = n * chunk
        self.nrounds = -(-count // self.round_elems)

    def piece(self, g, j):
        ck = self.last if g + 1 == self.nrounds else self.chunk
        off = g * self.round_elems + j * ck
        end = min(off + ck, self.count)
        ne = end - off if off < self.count else 0
        return min(off, self.count), ne


class Outputs:
    """What every rank wrote into its receive buffer."""

    def __init__(self, n):
        self.writes = [[] for _ in range(n)]

    def put(self, rank, off, ne, who):
        if ne:
            self.writes[rank].append((off, ne, who))

    def check(self, count, everyone, ranks=None, base=None):
        """Every listed rank's buffer [0, count) is tiled exactly once; `everyone`: the contributor set every element
        must carry, or a function offset -> set (all-gather: the block's owner)."""
        for r in (range(len(self.writes)) if ranks is None else ranks):
            at = 0
            for off, ne, who in sorted(self.writes[r]):
                assert off == at, (f"rank {r}: output not tiled exactly once around element {at}", off, ne)
                want = everyone(off) if callable(everyone) else everyone
                assert who == want, (f"rank {r}: elements at {off} carry contributions of {sorted(who)}", sorted(want))
                at = off + ne
            assert at == count, (f"rank {r}: output ends at {at} of {count}",)


def val(off, ne, who):
    return (off, ne, frozenset(who))


# ------------------------------------------------------------------ all-pairs schedules
def direct_allreduce(n, r, role, channels, plan, count, out):
    """directAllReduce: `channels` is the list of channels this workgroup serves (two for the scatter and gather
    workgroups of the two-rank role pairing, else one)."""
    R = Rounds(count, plan["chunkElems"], plan["lastChunkElems"], n)
    C = plan["nChannels"]
    peers = [p for p in range(n) if p != r]
    if role == 3:
        # one workgroup per channel walks all three roles, skewed by one round each: scatter of round k, reduce of round
        # k - 1, gather of round k - 2 (directAllReduce, role 3)
        ch = channels[0]
        k = 0
        while ch + (k - 2) * C < R.nrounds:
            for ph in range(3):
                if k - ph < 0 or ch + (k - ph) * C >= R.nrounds:
                    continue
                yield from _direct_allreduce_step(n, r, ph, ch, ch + (k - ph) * C, R, peers, out)
            k += 1
        return
    k = 0
    while channels[0] + k * C < R.nrounds:
        for ch in channels:
            g = ch + k * C
            if g >= R.nrounds:
                continue
            yield from _direct_allreduce_step(n, r, role, ch, g, R, peers, out)
        k += 1


def _direct_allreduce_step(n, r, role, ch, g, R, peers, out):
    if True:
        if True:
            if role == 0:
                def body(read, write, g=g):
                    for p in peers:
                        off, ne = R.piece(g, p)
                        write(p, 0, val(off, ne, {r}))
                yield Step(r, ch, [], [(p, 0) for p in peers], body)
            elif role == 1:
                def body(read, write, g=g):
                    off, ne = R.piece(g, r)
                    who = {r}
                    for p in peers:
                        got = read(p, 0)
                        assert got == val(off, ne, {p}), ("reducer", r, "round", g, "from", p, got, (off, ne))
                        who.add(p)
                    out.put(r, off, ne, frozenset(who))
                    for p in peers:
                        write(p, 1, val(off, ne, who))
                yield Step(r, ch, [(p, 0) for p in peers], [(p, 1) for p in peers], body)
            else:
                def body(read, write, g=g):
                    for p in peers:
                        off, ne = R.piece(g, p)
                        got = read(p, 1)
                        assert got[0] == off and got[1] == ne, ("gather", r, "round", g, "from", p, got, (off, ne))
                        out.put(r, off, ne, got[2])
                yield Step(r, ch, [(p, 1) for p in peers], [], body)


def direct_reduce_scatter(n, r, role, ch, plan, count, out, elem_size=4):
    chunk, C = plan["chunkElems"], plan["nChannels"]
    nrounds = -(-count // chunk)
    peers = [p for p in range(n) if p != r]
    if count * elem_size >= AG_ANNOUNCE_BYTES and ch < nrounds:
        # registered send buffers: the scatter role sends its word (kind 0) and waits for the reducers' answers (kind 1)
        if role == 0:
            yield Step(r, ch, [], [(p, 0) for p in peers], lambda read, write: [write(p, 0, "word") for p in peers])

            def answers(read, write):
                for p in peers:
                    assert read(p, 1) == "answer"
            yield Step(r, ch, [(p, 1) for p in peers], [], answers)
        else:
            def words(read, write):
                for p in peers:
                    assert read(p, 0) == "word"
            yield Step(r, ch, [(p, 0) for p in peers], [], words)
            yield Step(r, ch, [], [(p, 1) for p in peers], lambda read, write: [write(p, 1, "answer") for p in peers])
    for g in range(ch, nrounds, C):
        off = g * chunk
        ne = min(chunk, count - off)
        if role == 0:
            def body(read, write, off=off, ne=ne):
                for p in peers:
                    write(p, 0, val(p * count + off, ne, {r}))
            yield Step(r, ch, [], [(p, 0) for p in peers], body)
        else:
            def body(read, write, off=off, ne=ne):
                who = {r}
                for p in peers:
                    got = read(p, 0)
                    assert got == val(r * count + off, ne, {p}), ("reducer", r, g, p, got)
                    who.add(p)
                out.put(r, off, ne, frozenset(who))
            yield Step(r, ch, [(p, 0) for p in peers], [], body)


def direct_all_gather(n, r, role, ch, plan, nbytes, out):
    """directAllGather in bytes per rank; blocks of 4 MiB and more start with the announce round on the kind-0 flags."""
    chunk, C = plan["chunkElems"], plan["nChannels"]
    nrounds = -(-nbytes // chunk)
    peers = [p for p in range(n) if p != r]
    if nbytes >= AG_ANNOUNCE_BYTES and ch < nrounds:
        if role == 1:
            yield Step(r, ch, [], [(p, 0) for p in peers], lambda read, write: [write(p, 0, "word") for p in peers])
        else:
            def body(read, write):
                for p in peers:
                    assert read(p, 0) == "word"
            yield Step(r, ch, [(p, 0) for p in peers], [], body)
    for g in range(ch, nrounds, C):
        off = g * chunk
        nb = min(chunk, nbytes - off)
        if role == 0:
            def body(read, write, off=off, nb=nb):
                out.put(r, r * nbytes + off, nb, frozenset({r}))
                for p in peers:
                    write(p, 1, val(off, nb, {r}))
            yield Step(r, ch, [], [(p, 1) for p in peers], body)
        else:
            def body(read, write, off=off, nb=nb):
                for p in peers:
                    got = read(p, 1)
                    assert got == val(off, nb, {p}), ("gather", r, g, p, got)
                    out.put(r, p * nbytes + off, nb, frozenset({p}))
            yield Step(r, ch, [(p, 1) for p in peers], [], body)


# ------------------------------------------------------------------ ring schedules
def ring_pos(rings, ch, r, n):
    order = rings[ch % len(rings)]
    ix = order.index(r)
    return order, ix, order[(ix - 1) % n], order[(ix + 1) % n]


def announce_recv_buff(r, ch, frm, to):
    """announceRecvBuff: the registration word goes to the rank that sends finished pieces (`frm`) on the kind-1 flags,
    then the word of the rank this one sends them to (`to`) is awaited.  None: no such rank for this workgroup."""
    if frm is not None:
        yield Step(r, ch, [], [(frm, 1)], lambda read, write: write(frm, 1, "word"))
    if to is not None:
        def body(read, write):
            assert read(to, 1) == "word"
        yield Step(r, ch, [(to, 1)], [], body)


def ring_allreduce(n, r, ch, plan, count, rings, out, elem_size=4):
    R = Rounds(count, plan["chunkElems"], plan["lastChunkElems"], n)
    _, ix, prev, nxt = ring_pos(rings, ch, r, n)
    if count * elem_size >= AG_ANNOUNCE_BYTES and ch < R.nrounds:
        yield from announce_recv_buff(r, ch, prev, nxt)
    for g in range(ch, R.nrounds, plan["nChannels"]):
        off, ne = R.piece(g, (ix - 1) % n)
        yield Step(r, ch, [], [(nxt, 0)], lambda read, write, off=off, ne=ne: write(nxt, 0, val(off, ne, {r})))
        for j in range(2, n):
            off, ne = R.piece(g, (ix - j) % n)

            def body(read, write, off=off, ne=ne):
                got = read(prev, 0)
                assert got[0] == off and got[1] == ne and r not in got[2], ("ring hop", r, g, got, (off, ne))
                write(nxt, 0, val(off, ne, got[2] | {r}))
            yield Step(r, ch, [(prev, 0)], [(nxt, 0)], body)
        off, ne = R.piece(g, ix)

        def body(read, write, off=off, ne=ne):
            got = read(prev, 0)
            assert got[0] == off and got[1] == ne and len(got[2]) == n - 1 and r not in got[2], ("ring last hop", r, g, got)
            who = got[2] | {r}
            out.put(r, off, ne, who)
            write(nxt, 0, val(off, ne, who))
        yield Step(r, ch, [(prev, 0)], [(nxt, 0)], body)
        for j in range(1, n - 1):
            off, ne = R.piece(g, (ix - j) % n)

            def body(read, write, off=off, ne=ne):
                got = read(prev, 0)
                assert got[0] == off and got[1] == ne and len(got[2]) == n, ("ring forward", r, g, got, (off, ne))
                out.put(r, off, ne, got[2])
                write(nxt, 0, got)
            yield Step(r, ch, [(prev, 0)], [(nxt, 0)], body)
        off, ne = R.piece(g, (ix + 1) % n)

        def body(read, write, off=off, ne=ne):
            got = read(prev, 0)
            assert got[0] == off and got[1] == ne and len(got[2]) == n, ("ring final", r, g, got, (off, ne))
            out.put(r, off, ne, got[2])
        yield Step(r, ch, [(prev, 0)], [], body)


def ring_reduce_scatter(n, r, ch, plan, count, rings, out):
    chunk = plan["chunkElems"]
    order, ix, prev, nxt = ring_pos(rings, ch, r, n)
    nrounds = -(-count // chunk)
    for g in range(ch, nrounds, plan["nChannels"]):
        off = g * chunk
        ne = min(chunk, count - off)
        blk = order[(ix - 1) % n]
        yield Step(r, ch, [], [(nxt, 0)], lambda read, write, blk=blk, off=off, ne=ne: write(nxt, 0, val(blk * count + off, ne, {r})))
        for j in range(2, n):
            blk = order[(ix - j) % n]

            def body(read, write, blk=blk, off=off, ne=ne):
                got = read(prev, 0)
                assert got[0] == blk * count + off and got[1] == ne and r not in got[2], ("ring rs hop", r, g, got)
                write(nxt, 0, val(got[0], ne, got[2] | {r}))
            yield Step(r, ch, [(prev, 0)], [(nxt, 0)], body)

        def body(read, write, off=off, ne=ne):
            got = read(prev, 0)
            assert got[0] == r * count + off and got[1] == ne and len(got[2]) == n - 1 and r not in got[2], ("ring rs end", r, g, got)
            out.put(r, off, ne, got[2] | {r})
        yield Step(r, ch, [(prev, 0)], [], body)


def ring_all_gather(n, r, ch, plan, nbytes, rings, out):
    chunk = plan["chunkElems"]
    order, ix, prev, nxt = ring_pos(rings, ch, r, n)
    nrounds = -(-nbytes // chunk)
    if nbytes >= AG_ANNOUNCE_BYTES and ch < nrounds:
        yield from announce_recv_buff(r, ch, prev, nxt)
    for g in range(ch, nrounds, plan["nChannels"]):
        off = g * chunk
        nb = min(chunk, nbytes - off)

        def body(read, write, off=off, nb=nb):
            out.put(r, r * nbytes + off, nb, frozenset({r}))
            write(nxt, 0, val(r * nbytes + off, nb, {r}))
        yield Step(r, ch, [], [(nxt, 0)], body)
        for j in range(1, n - 1):
            blk = order[(ix - j) % n]

            def body(read, write, blk=blk, off=off, nb=nb):
                got = read(prev, 0)
                assert got == val(blk * nbytes + off, nb, {blk}), ("ring ag", r, g, got)
                out.put(r, got[0], nb, got[2])
                write(nxt, 0, got)
            yield Step(r, ch, [(prev, 0)], [(nxt, 0)], body)
        blk = order[(ix + 1) % n]

        def body(read, write, blk=blk, off=off, nb=nb):
            got = read(prev, 0)
            assert got == val(blk * nbytes + off, nb, {blk}), ("ring ag end", r, g, got)
            out.put(r, got[0], nb, got[2])
        yield Step(r, ch, [(prev, 0)], [], body)


# ------------------------------------------------------------------ chain ("tree") all-reduce
def chain_allreduce(n, r, role, ch, plan, count, rings, out, elem_size=4):
    chunk = plan["chunkElems"]
    order, ix, parent, child = ring_pos(rings, ch, r, n)
    root, leaf = ix == 0, ix == n - 1
    nrounds = -(-count // chunk)
    if count * elem_size >= AG_ANNOUNCE_BYTES and ch < nrounds:
        if role == 0:
            if root:
                yield from announce_recv_buff(r, ch, None, child)
        elif not root:
            yield from announce_recv_buff(r, ch, parent, None if leaf else child)
    for g in range(ch, nrounds, plan["nChannels"]):
        off = g * chunk
        ne = min(chunk, count - off)
        if role == 0:
            if leaf:
                yield Step(r, ch, [], [(parent, 0)], lambda read, write, off=off, ne=ne: write(parent, 0, val(off, ne, {r})))
            elif not root:
                def body(read, write, off=off, ne=ne):
                    got = read(child, 0)
                    assert got[0] == off and got[1] == ne and r not in got[2], ("chain up", r, g, got)
                    write(parent, 0, val(off, ne, got[2] | {r}))
                yield Step(r, ch, [(child, 0)], [(parent, 0)], body)
            else:
                def body(read, write, off=off, ne=ne):
                    got = read(child, 0)
                    assert got[0] == off and got[1] == ne and len(got[2]) == n - 1, ("chain root", r, g, got)
                    who = got[2] | {r}
                    out.put(r, off, ne, who)
                    write(child, 1, val(off, ne, who))
                yield Step(r, ch, [(child, 0)], [(child, 1)], body)
        else:
            if root:
                continue
            if leaf:
                def body(read, write, off=off, ne=ne):
                    got = read(parent, 1)
                    assert got[0] == off and got[1] == ne and len(got[2]) == n, ("chain down leaf", r, g, got)
                    out.put(r, off, ne, got[2])
                yield Step(r, ch, [(parent, 1)], [], body)
            else:
                def body(read, write, off=off, ne=ne):
                    got = read(parent, 1)
                    assert got[0] == off and got[1] == ne and len(got[2]) == n, ("chain down", r, g, got)
                    out.put(r, off, ne, got[2])
                    write(child, 1, got)
                yield Step(r, ch, [(parent, 1)], [(child, 1)], body)


# ------------------------------------------------------------------ one launch on every rank
def launch(net, coll, algo, plan, count, elem_size, rings, cus=256, one_workgroup=False):
    """All workgroups of one collective launch on all ranks, run to completion.  `count`: elements (all-reduce), elements
    per rank (reduce-scatter), elements per rank (all-gather; moved as bytes).  Returns (Outputs, steps run)."""
    n = net.n
    C = plan["nChannels"]
    out = Outputs(n)
    progs = []
    for r in range(n):
        wgs = 0
        if algo == DIRECT and coll == ALLREDUCE and one_workgroup:
            # RCCL_AMD_FUSED_ROLES=1: grid = channels
            for ch in range(C):
                progs.append(direct_allreduce(n, r, 3, [ch], plan, count, out))
            wgs = plan["grid"]
        elif algo == DIRECT and coll == ALLREDUCE:
            pair = n == 2 and C >= 2 and C % 2 == 0                 # pairsRoles in enqueue.cc
            for ch in range(C):
                progs.append(direct_allreduce(n, r, 1, [ch], plan, count, out))
                wgs += 1
            for role in (0, 2):
                for ch in (range(0, C, 2) if pair else range(C)):
                    progs.append(direct_allreduce(n, r, role, [ch, ch + 1] if pair else [ch], plan, count, out))
                    wgs += 1
        elif algo == DIRECT and coll == REDUCESCATTER:
            for role in (0, 1):
                for ch in range(C):
                    progs.append(direct_reduce_scatter(n, r, role, ch, plan, count, out, elem_size))
                    wgs += 1
        elif algo == DIRECT:
            for role in (0, 1):
                for ch in range(C):
                    progs.append(direct_all_gather(n, r, role, ch, dict(plan, chunkElems=plan["chunkElems"] * elem_size), count * elem_size, out))
                    wgs += 1
        elif algo == RING:
            for ch in range(C):
                if coll == ALLREDUCE:
                    progs.append(ring_allreduce(n, r, ch, plan, count, rings, out, elem_size))
                elif coll == REDUCESCATTER:
                    progs.append(ring_reduce_scatter(n, r, ch, plan, count, rings, out))
                else:
                    progs.append(ring_all_gather(n, r, ch, dict(plan, chunkElems=plan["chunkElems"] * elem_size), count * elem_size, rings, out))
                wgs += 1
        else:
            for role in (0, 1):
                for ch in range(C):
                    progs.append(chain_allreduce(n, r, role, ch, plan, count, rings, out, elem_size))
                    wgs += 1
        # every workgroup of a launch waits for workgroups elsewhere, so all must be resident: one per compute unit
        assert wgs == plan["grid"], (wgs, plan)
        assert wgs <= cus, f"{wgs} workgroups do not fit {cus} compute units"
    steps = run(net, progs)
    return out, steps
