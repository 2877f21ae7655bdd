#!/usr/bin/env python3
"""Communicator set-up at the geometry of a real 8-GPU node (56 channels: a 1464 MiB window per rank, every rank mapping the
seven others through HIP IPC) with the 8 ranks sharing ONE GPU, then small collectives that need few workgroups (line
protocol, one exchange, two exchanges).  The FIFO schedules themselves cannot run at this geometry on one GPU (168
workgroups per rank); tests/test_schedule_sim.py covers their protocol."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import expected as ex  # noqa: E402
import oracle as orc  # noqa: E402
from testbed import COLL_ALLREDUCE, TestBed  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    bed = TestBed(n, env={"RCCL_AMD_NCHANNELS": "56", "RCCL_AMD_BED_ARENA_MIB": "64"})
    try:
        t0 = time.time()
        bed.init_comms({"NCCL_ALGO": None})
        print(f"{n} communicators with 56 channels up in {time.time() - t0:.1f} s", flush=True)
        for count in (1250, 65536, 262144 + 3, (1 << 20) + 5, (4 << 20) + 7):     # lines, one exchange, two exchanges (16 MiB)
            ins = [orc.fill_pattern(orc.F32, count, r) for r in range(n)]
            plan = bed.plan(COLL_ALLREDUCE, count, orc.F32)
            outs = bed.run(COLL_ALLREDUCE, orc.F32, orc.SUM, count, ins, count * 4)
            want = ex.allreduce(orc.F32, orc.SUM, ins, plan)
            bad = [ex.check(orc.F32, outs[r], want, exact=True) for r in range(n)]
            print(f"all-reduce of {count} fp32: plan {plan}, {'ok' if not any(bad) else bad}", flush=True)
    finally:
        bed.close()


if __name__ == "__main__":
    main()
