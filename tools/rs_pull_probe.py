#!/usr/bin/env python3
"""Diagnostic: which earlier launch makes a later small reduce-scatter come out wrong?  Sequences of registered
reduce-scatter runs on the multi-process test bed, every output compared with the oracle; mismatching byte ranges printed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import expected as ex  # noqa: E402
import oracle as orc  # noqa: E402
from testbed import COLL_REDUCESCATTER, TestBed  # noqa: E402


def ranges(bad):
    idx = np.flatnonzero(bad)
    if idx.size == 0:
        return []
    cuts = np.flatnonzero(np.diff(idx) > 1)
    starts = np.concatenate(([idx[0]], idx[cuts + 1]))
    ends = np.concatenate((idx[cuts], [idx[-1]]))
    return [(int(a), int(b - a + 1)) for a, b in zip(starts, ends)][:8]


def run(bed, label, count, dt=orc.F32, op=orc.SUM, **kw):
    n = bed.nranks
    ins = [orc.fill_pattern(dt, count * n, r) for r in range(n)]
    plan = bed.plan(COLL_REDUCESCATTER, count, dt)
    for k, (outs, stats) in enumerate(bed.run_registered(dt, op, count, ins, coll=COLL_REDUCESCATTER, **kw)):
        for r in range(n):
            want = np.ascontiguousarray(ex.reduce_scatter(dt, op, ins, r, plan)).view(np.uint8).ravel()
            got = np.asarray(outs[r]).view(np.uint8).ravel()
            bad = got != want
            print(f"{label} run {k} rank {r}: {int(bad.sum())} wrong bytes {ranges(bad)} first got {got[bad][:8].tolist() if bad.any() else ''} stats {stats[r]} plan ch {plan['nChannels']} chunk {plan['chunkElems']}", flush=True)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    bed = TestBed(n, env={"RCCL_AMD_SPIN_TIMEOUT_MS": "5000"})
    try:
        bed.init_comms({"NCCL_ALGO": None, "RCCL_AMD_ONESHOT_MAX_BYTES": "0", "NCCL_PROTO": "Simple"})
        big, small = (1 << 20) + 1056, (128 << 10) + 16
        which = sys.argv[2] if len(sys.argv) > 2 else "all"
        run(bed, "small, fresh", small)
        if which == "graph_unreg":
            run(bed, "big unregistered, graph", big, runs=2, graph=True, register_ranks=[])
        elif which == "graph_coarse":
            run(bed, "big coarse, graph", big, runs=2, graph=True, fine=False)
        elif which == "eager3":
            run(bed, "big pulled x3", big, runs=3)
        else:
            run(bed, "big pulled, graph", big, runs=2, graph=True)
        run(bed, "small after", small)
        run(bed, "small after, again", small)
        run(bed, "small coarse", small, fine=False)
        run(bed, "small unregistered", small, register_ranks=[])
        run(bed, "mid 1 MiB per rank", 262144 + 16)
        run(bed, "small once more", small)
    finally:
        bed.close()


if __name__ == "__main__":
    main()
