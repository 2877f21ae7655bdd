#!/usr/bin/env python3
"""Probe for poison surviving in a coarse-grained (hipMalloc) receive buffer: repeats a 2-rank all-reduce whose
receive buffer the owner fills with 0x5A just before the launch, for several buffer/registration variants, and
reports where the output still shows the poison.

    python tools/coarse_probe.py [--iters 40] [--ranks 2]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--count", type=int, default=(1 << 20) + 24)
    a = ap.parse_args()
    import expected as ex
    import oracle as orc
    from testbed import COLL_ALLREDUCE, TestBed
    tb = TestBed(a.ranks, devices=(0,))
    try:
        tb.init_comms({"NCCL_ALGO": None})
        n, count = a.ranks, a.count
        ins = [orc.fill_pattern(orc.F32, count, r) for r in range(n)]
        plan = tb.plan(COLL_ALLREDUCE, count, orc.F32)
        want = np.ascontiguousarray(ex.allreduce(orc.F32, orc.SUM, ins, plan)).view(np.uint8)
        print("plan", plan, flush=True)
        for name, kw in (("coarse registered", dict(fine=False)),
                         ("coarse not registered", dict(fine=False, register_ranks=[])),
                         ("fine registered", dict(fine=True)),
                         ("fine not registered", dict(fine=True, register_ranks=[]))):
            bad_runs = 0
            for it in range(a.iters):
                (outs, stats), = tb.run_registered(orc.F32, orc.SUM, count, ins, **kw)
                for r in range(n):
                    diff = np.nonzero(outs[r] != want)[0]
                    if diff.size:
                        bad_runs += 1
                        poison = int((outs[r][diff] == 0x5A).sum())
                        lines = sorted(set((diff // 128).tolist()))
                        print(f"{name}: iter {it} rank {r}: {diff.size} bytes differ ({poison} are poison), first {diff[0]} last {diff[-1]}, "
                              f"{len(lines)} lines, first lines {lines[:8]}, stats {stats[r]}", flush=True)
            print(f"{name}: {bad_runs} bad outputs in {a.iters} iterations; last stats {stats}", flush=True)
    finally:
        tb.close()


if __name__ == "__main__":
    main()
