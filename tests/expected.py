"""Expected results of the collectives, built from the CPU oracle.

The association order of a floating-point reduction depends on the schedule, so the
bit-exact expectation needs the schedule's geometry (channels, chunk, rings).  This module
restates that geometry on the host and asks the oracle for each piece's fold:

  direct : every element is folded in rank order 0..n-1  (== the reference's host oracle,
           test/common/PrepDataFuncs.cpp:95-146, for Sum/Prod/Min/Max)
  ring   : piece j of a round is seeded at ring position j+1 and finished at position j,
           each hop computing local (op) received  (src/device/all_reduce.h:82-144)
  tree   : chain in ring order, leaf first, root last (src/device/all_reduce.h:269-289)
"""
import numpy as np

import oracle as orc

ALGO_DIRECT, ALGO_RING, ALGO_TREE = 0, 1, 2


def full_rings(n):
    """Ring orders on n all-to-all ranks, from the committed fixture (tests/golden/ring_orders_all_to_all.json), not from
    the library: tests/test_topo.py holds the library's ring search to the same file, so an expectation in ring or chain
    order never silently follows a change of that search."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ring_orders_all_to_all.json")) as f:
        rings = json.load(f)[str(n)]
    return rings if rings else [list(range(n))]


def _unit(dt):
    return 16 // np.dtype(orc.NP_OF[dt]).itemsize


def _last_chunk(count, n, chunk, dt):
    unit = _unit(dt)
    round_elems = chunk * n
    nrounds = -(-count // round_elems)
    rem = count - (nrounds - 1) * round_elems
    last = -(-(-(-rem // n)) // unit) * unit
    return nrounds, min(last, chunk)


def allreduce(dt, op, inputs, plan, rings=None):
    """Bit-exact expectation of ncclAllReduce for the given plan dict (algo, nChannels, chunkElems)."""
    n = len(inputs)
    count = inputs[0].size
    algo, nch, chunk = plan["algo"], plan["nChannels"], plan["chunkElems"]
    if algo == ALGO_DIRECT:
        return orc.device_fold(dt, op, inputs)
    rings = rings or full_rings(n)
    out = orc.empty(dt, count)
    if algo == ALGO_TREE:
        nrounds = -(-count // chunk)
        for g in range(nrounds):
            ring = rings[(g % nch) % len(rings)]
            lo, hi = g * chunk, min(count, (g + 1) * chunk)
            out[lo:hi] = orc.chain_allreduce(dt, op, [a[lo:hi] for a in inputs], ring)
        return out
    nrounds, last = _last_chunk(count, n, chunk, dt)
    for g in range(nrounds):
        ring = rings[(g % nch) % len(rings)]
        ck = last if g == nrounds - 1 else chunk
        base = g * chunk * n
        for j in range(n):
            lo = base + j * ck
            hi = min(count, lo + ck)
            if lo >= hi:
                continue
            order = [ring[(j + 1 + k) % n] for k in range(n)]
            out[lo:hi] = orc.device_fold(dt, op, [a[lo:hi] for a in inputs], order, rank_first=True)
    return out


def reduce_scatter(dt, op, inputs, rank, plan, rings=None):
    """Bit-exact expectation of rank's ncclReduceScatter output (count = elements per rank)."""
    n = len(inputs)
    count = inputs[0].size // n
    algo, nch, chunk = plan["algo"], plan["nChannels"], plan["chunkElems"]
    blk = [a[rank * count:(rank + 1) * count] for a in inputs]
    if algo == ALGO_DIRECT:
        return orc.device_fold(dt, op, blk)
    rings = rings or full_rings(n)
    out = orc.empty(dt, count)
    nrounds = -(-count // chunk)
    for g in range(nrounds):
        ring = rings[(g % nch) % len(rings)]
        ix = ring.index(rank)
        order = [ring[(ix + 1 + k) % n] for k in range(n)]
        lo, hi = g * chunk, min(count, (g + 1) * chunk)
        out[lo:hi] = orc.device_fold(dt, op, [b[lo:hi] for b in blk], order, rank_first=True)
    return out


def check(dt, got_bytes, want, exact=True):
    """Compare a rank's raw output bytes with the expectation; returns an error string or None."""
    got = np.frombuffer(got_bytes, dtype=np.uint8)[:want.nbytes].view(want.dtype)
    if exact:
        if not np.array_equal(got.view(np.uint8), want.view(np.uint8)):
            bad = np.nonzero(got.view(np.uint8).reshape(-1, want.dtype.itemsize) !=
                             want.view(np.uint8).reshape(-1, want.dtype.itemsize))[0]
            i = int(bad[0])
            elems = np.unique(bad)
            cut = np.nonzero(np.diff(elems) != 1)[0]
            starts = np.concatenate(([elems[0]], elems[cut + 1]))
            ends = np.concatenate((elems[cut], [elems[-1]]))
            runs = ", ".join(f"{int(a)}-{int(b)}" for a, b in list(zip(starts, ends))[:12])
            return (f"first bit mismatch at element {i} of {want.size}: got {got[i]!r}, want {want[i]!r} ({bad.size} byte diffs, "
                    f"{elems.size} elements in {starts.size} runs: {runs}; got {got[elems[:6]].tolist()} want {want[elems[:6]].tolist()})")
        return None
    i = orc.first_mismatch(dt, got, want)
    if i >= 0:
        return f"element {i} outside the reference tolerance: got {got[i]!r}, want {want[i]!r}"
    return None
