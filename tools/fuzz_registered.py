#!/usr/bin/env python3
"""Randomised parity run of the registered-buffer paths: all-reduce (all-pairs, ring, chain), all-gather (all-pairs,
ring) and reduce-scatter with a random subset of ranks registered, random element types, operators, counts on both sides
of the 4 MiB from which the registration words are exchanged, in place or not, sometimes coarse-grained memory (which
must stay on the FIFOs).  Every result bit for bit against the oracle's fold in the schedule's order.

    python tools/fuzz_registered.py [--ranks 4] [--cases 60] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import expected as ex  # noqa: E402
import oracle as orc  # noqa: E402
from testbed import COLL_ALLGATHER, COLL_ALLREDUCE, COLL_REDUCESCATTER, TestBed  # noqa: E402

DTYPES = [orc.I8, orc.U8, orc.I32, orc.U32, orc.I64, orc.U64, orc.F16, orc.F32, orc.F64, orc.BF16, orc.F8E4M3, orc.F8E5M2]
OPS = [orc.SUM, orc.PROD, orc.MAX, orc.MIN, orc.AVG]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=4)
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    n = args.ranks
    rng = np.random.default_rng(args.seed)
    bed = TestBed(n, env={"RCCL_AMD_SPIN_TIMEOUT_MS": "8000"})
    fails, done, t0 = 0, 0, time.time()
    try:
        while done < args.cases:
            algo = ["Direct", "Ring", "Tree"][int(rng.integers(0, 3))]
            bed.init_comms({"NCCL_ALGO": algo, "NCCL_PROTO": "Simple", "RCCL_AMD_ONESHOT_MAX_BYTES": "0",
                            "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": "0"})
            for _ in range(int(rng.integers(3, 7))):
                if done >= args.cases:
                    break
                done += 1
                coll = [COLL_ALLREDUCE, COLL_ALLREDUCE, COLL_ALLGATHER, COLL_REDUCESCATTER][int(rng.integers(0, 4))]
                dt = DTYPES[int(rng.integers(0, len(DTYPES)))]
                op = OPS[int(rng.integers(0, len(OPS)))]
                e = np.dtype(orc.NP_OF[dt]).itemsize
                per = n if coll != COLL_ALLREDUCE else 1
                nbytes = int(2 ** rng.uniform(20, 24.5)) // per if rng.integers(0, 4) else int(2 ** rng.uniform(10, 20))
                if coll != COLL_ALLREDUCE and rng.integers(0, 2):
                    nbytes = int(2 ** rng.uniform(22, 23.3))                      # per rank: above the threshold
                count = max(1, nbytes // e + int(rng.integers(-3, 4)))
                in_place = bool(rng.integers(0, 2))
                regs = [r for r in range(n) if rng.integers(0, 4)]              # each rank registered with probability 3/4
                fine = bool(rng.integers(0, 6))
                runs = 2 if not in_place and rng.integers(0, 2) else 1
                graph = runs == 2 and bool(rng.integers(0, 3) == 0)
                what = (f"case {done}: {algo} coll {coll} {orc.DT_NAMES[dt]} {orc.OP_NAMES[op]} count {count} in_place {in_place} "
                        f"registered {regs} fine {fine} runs {runs} graph {graph}")
                try:
                    if coll == COLL_ALLGATHER:
                        ins = [orc.fill_pattern(orc.U8, count * e, r) for r in range(n)]
                        res = bed.run_registered(orc.U8, orc.SUM, count * e, ins, in_place=in_place, register_ranks=regs, runs=runs,
                                                 coll=coll, graph=graph, fine=fine)
                        wants = [orc.all_gather_expected(ins)] * n
                        cdt = orc.U8
                    elif coll == COLL_REDUCESCATTER:
                        ins = [orc.fill_pattern(dt, count * n, r) for r in range(n)]
                        res = bed.run_registered(dt, op, count, ins, in_place=in_place, register_ranks=regs, runs=runs, coll=coll,
                                                 graph=graph, fine=fine)
                        plan = bed.plan(coll, count, dt)
                        wants = [ex.reduce_scatter(dt, op, ins, r, plan) for r in range(n)]
                        cdt = dt
                    else:
                        ins = [orc.fill_pattern(dt, count, r) for r in range(n)]
                        res = bed.run_registered(dt, op, count, ins, in_place=in_place, register_ranks=regs, runs=runs, graph=graph, fine=fine)
                        wants = [ex.allreduce(dt, op, ins, bed.plan(coll, count, dt))] * n
                        cdt = dt
                except Exception as exc:  # noqa: BLE001
                    print("ERROR", what, "->", repr(exc)[:600], flush=True)
                    print(f"STOPPED after {done} cases, {fails} failures and one error")
                    return 2
                bad = [b for outs, _ in res for b in (ex.check(cdt, outs[r], wants[r], exact=True) for r in range(n)) if b]
                direct = sum(s["direct_pieces"] for _, stats in res for s in stats)
                if not fine and direct:
                    bad.append(f"{direct} pieces were written directly into coarse-grained memory")
                if bad:
                    fails += 1
                    print("MISMATCH", what, "->", bad[0][:400], flush=True)
                if done % 10 == 0:
                    print(f"{done} cases, {fails} failures, {time.time() - t0:.0f} s (last: {direct} direct pieces)", flush=True)
    finally:
        bed.close()
    print(f"DONE {done} cases, {fails} failures")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
