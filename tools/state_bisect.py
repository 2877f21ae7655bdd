#!/usr/bin/env python3
"""Which earlier activity of a process makes DMA read-backs of freshly allocated hipMalloc memory return stale bytes?
Runs the test bed's library-free probe (hipMemset poison -> hipMemcpyAsync D2D on a side stream -> synchronize ->
hipMemcpy D2H) between the activities the GPU test suite performs, in the order it performs them."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import oracle as orc
    from testbed import COLL_ALLGATHER, COLL_ALLREDUCE, TestBed
    n = 2
    tb = TestBed(n, devices=(0,))
    variants = ("fresh/dma/plainstores", "fresh/dma", "fresh/dma-after-sweep/plainstores", "fresh/kernel")

    def probe(tag):
        res = tb._all("probe_poison", [dict(nbytes=4194400, reps=6, variants=variants)] * n)
        print(tag, res, flush=True)

    try:
        tb._all("ping", [None] * n)
        probe("before any communicator")
        tb.init_comms({"NCCL_ALGO": None})
        probe("after init_comms")
        x = [orc.fill_pattern(orc.F32, 393216, r) for r in range(n)]
        tb.run(COLL_ALLREDUCE, orc.F32, orc.SUM, 393216, x, 393216 * 4)
        probe("after one all-reduce")
        for _ in range(3):
            tb.init_comms({"NCCL_ALGO": None})
        probe("after 3 re-inits")
        u = [orc.fill_pattern(orc.U8, 2500, r) for r in range(n)]
        tb.run(COLL_ALLREDUCE, orc.U8, orc.SUM, 2500, u, 2500, managed=True)
        probe("after a managed-memory all-reduce")
        tb.run(COLL_ALLREDUCE, orc.F32, orc.SUM, 393216, x, 393216 * 4, graph=True)
        probe("after a graph capture + replay")
        tb.init_comms({"NCCL_P2P_DISABLE": "1", "RCCL_AMD_NCHANNELS": "2", "NCCL_ALGO": None})
        tb.run(COLL_ALLREDUCE, orc.F32, orc.SUM, 393216, x, 393216 * 4)
        tb.init_comms({"NCCL_P2P_DISABLE": None, "RCCL_AMD_NCHANNELS": tb.base_env["RCCL_AMD_NCHANNELS"], "NCCL_ALGO": None})
        probe("after the host-staged transport")
        count = (1 << 20) + 24
        ins = [orc.fill_pattern(orc.F32, count, r) for r in range(n)]
        tb.run_registered(orc.F32, orc.SUM, count, ins, runs=2)
        probe("after registered (fine-grained) all-reduces")
        tb.run_registered(orc.F32, orc.SUM, count, ins, runs=2, graph=True)
        probe("after registered all-reduces captured into a graph")
        tb.run_registered(orc.F32, orc.SUM, count, ins, register_ranks=[0])
        probe("after only rank 0 registered")
        c2 = 3 * 1024 * 1024 + 1000
        ins2 = [orc.fill_pattern(orc.BF16, c2, r) for r in range(n)]
        tb.run_registered(orc.BF16, orc.AVG, c2, ins2, in_place=True)
        probe("after in-place registered")
        tb.run_registered(orc.F32, orc.SUM, count, ins, register_ranks=[])
        probe("after unregistered fine-grained")
    finally:
        tb.close()


if __name__ == "__main__":
    main()
