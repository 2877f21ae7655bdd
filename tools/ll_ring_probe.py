#!/usr/bin/env python3
"""One all-reduce under NCCL_ALGO=<Ring|Tree> NCCL_PROTO=LL on the multi-process bed, compared with the schedule-order fold.
usage: ll_ring_probe.py RANKS ALGO DTYPE OP COUNT PAD_SEND PAD_RECV [REPEATS]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import expected as ex  # noqa: E402
import oracle as orc  # noqa: E402
from testbed import COLL_ALLREDUCE, TestBed  # noqa: E402


def main():
    n, algo = int(sys.argv[1]), sys.argv[2]
    dt, op, count, ps, pr = (int(x) for x in sys.argv[3:8])
    reps = int(sys.argv[8]) if len(sys.argv) > 8 else 1
    bed = TestBed(n, env={"RCCL_AMD_SPIN_TIMEOUT_MS": "5000"})
    try:
        bed.init_comms({"NCCL_ALGO": algo, "NCCL_PROTO": os.environ.get("PROBE_PROTO", "LL")})
        e = np.dtype(orc.NP_OF[dt]).itemsize
        for it in range(reps):
            ins = [orc.fill_pattern(dt, count, r) for r in range(n)]
            outs = bed.run(COLL_ALLREDUCE, dt, op, count, ins, count * e, pad=(ps, pr))
            plan = bed.plan(COLL_ALLREDUCE, count, dt)
            want = ex.allreduce(dt, op, ins, plan)
            bad = [ex.check(dt, outs[r], want) for r in range(n)]
            print("run", it, plan, "rings[0]", ex.full_rings(n)[0], ["ok" if not b else b[:120] for b in bad][:3], flush=True)
    finally:
        bed.close()


if __name__ == "__main__":
    main()
