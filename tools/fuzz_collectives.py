#!/usr/bin/env python3
"""Randomised parity run: collectives with random element types, operators, counts, algorithms, protocols,
in-place flags and unaligned user pointers, every result compared bit for bit with the CPU oracle.

    python tools/fuzz_collectives.py [--ranks 4] [--cases 200] [--seed 1]

Uses the multi-process test bed (several ranks on one GPU).  Not part of the pytest suite: it is the long
version of tests/test_gpu_collectives.py, run by hand when the kernels change.
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
from testbed import (COLL_ALLGATHER, COLL_ALLREDUCE, COLL_BROADCAST, COLL_REDUCE, COLL_REDUCESCATTER,  # noqa: E402
                     TestBed)

DTYPES = [orc.I8, orc.U8, orc.I32, orc.U32, orc.I64, orc.U64, orc.F16, orc.F32, orc.F64, orc.BF16, orc.F8E4M3, orc.F8E5M2]
OPS = [orc.SUM, orc.PROD, orc.MAX, orc.MIN, orc.AVG]


def es(dt):
    return np.dtype(orc.NP_OF[dt]).itemsize


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=4)
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-bytes", type=int, default=8 << 20)
    ap.add_argument("--arena-mib", type=int, default=600, help="size of the start-up arena the rank processes cut their buffers from (0: none)")
    args = ap.parse_args()
    n = args.ranks
    rng = np.random.default_rng(args.seed)
    # buffers are cut from a tensor made when the rank processes start (tests/testbed.py, _DevBuf.arena): blocks torch has
    # just obtained from the driver have shown stale 128-byte lines on read-back on the one-GPU box, whatever wrote them
    bed = TestBed(n, env={"RCCL_AMD_BED_ARENA_MIB": str(args.arena_mib) if args.arena_mib else ""})
    fails = 0
    t0 = time.time()
    done = 0
    while done < args.cases:
        algo = ["Direct", "Ring", "Tree"][int(rng.integers(0, 3))]
        proto = [None, "Simple", "LL"][int(rng.integers(0, 3))]
        exchange = [None, None, "0"][int(rng.integers(0, 3))]          # "0": the FIFO schedules at every size above the line protocol's
        two = [None, None, "0", str(64 << 20)][int(rng.integers(0, 4))]
        bed.init_comms({"NCCL_ALGO": algo, "NCCL_PROTO": proto, "RCCL_AMD_ONESHOT_MAX_BYTES": exchange,
                        "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": two})
        for _ in range(int(rng.integers(4, 12))):
            if done >= args.cases:
                break
            done += 1
            # broadcast / reduce are one-hop schedules for up to 8 ranks
            coll = [COLL_ALLREDUCE, COLL_ALLREDUCE, COLL_REDUCESCATTER, COLL_ALLGATHER, COLL_BROADCAST, COLL_REDUCE][int(rng.integers(0, 6 if n <= 8 else 4))]
            dt = DTYPES[int(rng.integers(0, len(DTYPES)))]
            op = OPS[int(rng.integers(0, len(OPS)))]
            e = es(dt)
            per_rank_mult = n if coll in (COLL_REDUCESCATTER, COLL_ALLGATHER) else 1
            max_count = max(1, args.max_bytes // (e * per_rank_mult))
            count = int(min(max_count, max(1, 2 ** rng.uniform(0, 23) + rng.integers(-3, 4))))
            in_place = bool(rng.integers(0, 2))
            pad = (int(rng.integers(0, 5)) * e, int(rng.integers(0, 5)) * e)
            root = int(rng.integers(0, n))
            what = (f"case {done}: algo {algo} proto {proto} exchange {exchange} two {two} coll {coll} {orc.DT_NAMES[dt]} {orc.OP_NAMES[op]} count {count} "
                    f"in_place {in_place} pad {pad} root {root}")
            try:
                if coll == COLL_ALLREDUCE:
                    ins = [orc.fill_pattern(dt, count, r) for r in range(n)]
                    outs = bed.run(coll, dt, op, count, ins, count * e, in_place=in_place, pad=pad)
                    want = ex.allreduce(dt, op, ins, bed.plan(coll, count, dt))
                    bad = [ex.check(dt, outs[r], want) for r in range(n)]
                elif coll == COLL_REDUCESCATTER:
                    ins = [orc.fill_pattern(dt, count * n, r) for r in range(n)]
                    outs = bed.run(coll, dt, op, count, ins, count * e, in_place=in_place, pad=pad)
                    plan = bed.plan(coll, count, dt)
                    bad = [ex.check(dt, outs[r], ex.reduce_scatter(dt, op, ins, r, plan)) for r in range(n)]
                elif coll == COLL_ALLGATHER:
                    ins = [orc.fill_pattern(dt, count, r) for r in range(n)]
                    outs = bed.run(coll, dt, orc.SUM, count, ins, count * n * e, in_place=in_place, pad=pad)
                    want = orc.all_gather_expected(ins)
                    bad = [ex.check(dt, outs[r], want) for r in range(n)]
                elif coll == COLL_BROADCAST:
                    ins = [orc.fill_pattern(dt, count, root if r == root else 100 + r) for r in range(n)]
                    outs = bed.run(coll, dt, orc.SUM, count, ins, count * e, in_place=in_place, root=root, pad=pad)
                    bad = [ex.check(dt, outs[r], ins[root]) for r in range(n)]
                else:
                    ins = [orc.fill_pattern(dt, count, r) for r in range(n)]
                    outs = bed.run(coll, dt, op, count, ins, count * e, in_place=in_place, root=root, pad=pad)
                    bad = [ex.check(dt, outs[root], orc.device_fold(dt, op, ins))]
            except Exception as exc:  # noqa: BLE001
                # a rank that faulted or timed out: nothing more is started on the GPU
                print("ERROR", what, "->", repr(exc)[:600], flush=True)
                bed.close()
                print(f"STOPPED after {done} cases, {fails} failures and one error")
                return 2
            bad = [b for b in bad if b]
            if bad:
                fails += 1
                print("MISMATCH", what, "->", bad[0][:400], flush=True)
            if done % 25 == 0:
                print(f"{done} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
    bed.close()
    print(f"DONE {done} cases, {fails} failures")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
