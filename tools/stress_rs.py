#!/usr/bin/env python3
"""Stress the unaligned in-place reduce-scatter path and describe any mismatch.

    python tools/stress_rs.py [--ranks 4] [--iters 60] [--algo Direct]

Every rank j contributes ((it*7 + i) & 0xff) << (8*j) at element i, so byte lane j of a wrong
sum shows whose piece was wrong and whether it held the previous iteration's value.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from testbed import TestBed, COLL_REDUCESCATTER  # noqa: E402

I32, F8, SUM, MAX = 2, 10, 0, 2


def runs(idx):
    out, s, p = [], idx[0], idx[0]
    for v in idx[1:]:
        if v != p + 1:
            out.append((s, p))
            s = v
        p = v
    out.append((s, p))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=4)
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--algo", default="Direct")
    ap.add_argument("--count", type=int, default=54235)
    ap.add_argument("--filler", type=int, default=104857)
    ap.add_argument("--reinit", action="store_true", help="fresh communicators and the test suite's case prefix every iteration")
    ap.add_argument("--ll-prelude", action="store_true", help="a low-latency all-reduce on a previous communicator first")
    args = ap.parse_args()
    n, count = args.ranks, args.count
    bed = TestBed(n)
    bed.init_comms({"NCCL_ALGO": args.algo})
    bad = 0
    i = np.arange(count * n, dtype=np.int64)
    import oracle as orc
    prefix = [(orc.F32, orc.MAX, 393216), (orc.F32, orc.MAX, 384), (orc.F64, orc.MAX, 104857), (orc.BF16, orc.MAX, 104857),
              (orc.F8E4M3, orc.MAX, 104857)]
    for it in range(args.iters):
        if args.reinit:
            for variant in (0, 1):
                if args.ll_prelude:
                    bed.init_comms({"NCCL_ALGO": None})
                    pin = [orc.fill_pattern(orc.F32, 50000, r) for r in range(n)]
                    bed.run(0, orc.F32, orc.SUM, 50000, pin, 200000)
                bed.init_comms({"NCCL_ALGO": args.algo})
                for dt, op, c in prefix:
                    pin = [orc.fill_pattern(dt, c * n, r) for r in range(n)]
                    bed.run(COLL_REDUCESCATTER, dt, op, c, pin, c * np.dtype(orc.NP_OF[dt]).itemsize)
                if variant == 0:
                    pin = [orc.fill_pattern(orc.I32, count * n, r) for r in range(n)]
                    outs = bed.run(COLL_REDUCESCATTER, orc.I32, orc.PROD, count, pin, count * 4, in_place=True)
                    for r in range(n):
                        want = orc.reduce_scatter_expected(orc.I32, orc.PROD, pin, r).view(np.uint32)
                        got = outs[r].view(np.uint32)
                        d = np.nonzero(got != want)[0]
                        if d.size:
                            bad += 1
                            rr = runs(d.tolist())
                            print(f"iter {it} PROD rank {r}: {d.size} wrong, runs {rr[:10]}; got {got[d[:4]]} want {want[d[:4]]}", flush=True)
                            k = int(d[0])
                            print("   inputs at first:", [int(pin[j][r * count + k]) for j in range(n)], flush=True)
                    continue
                break
        if args.filler and not args.reinit:
            fill = [np.full(args.filler * n, 0x30 + r, dtype=np.uint8) for r in range(n)]
            bed.run(COLL_REDUCESCATTER, F8, MAX, args.filler, fill, args.filler)
        ins = [((((it * 7 + i) & 0xff) << (8 * j)) & 0xffffffff).astype(np.uint32).view(np.int32) for j in range(n)]
        outs = bed.run(COLL_REDUCESCATTER, I32, SUM, count, ins, count * 4, in_place=True)
        for r in range(n):
            want = np.zeros(count, dtype=np.uint32)
            for j in range(n):
                want += ins[j][r * count:(r + 1) * count].view(np.uint32)
            got = outs[r].view(np.uint32)
            d = np.nonzero(got != want)[0]
            if d.size:
                bad += 1
                lanes = [int(np.count_nonzero(((got[d] ^ want[d]) >> (8 * j)) & 0xff)) for j in range(n)]
                rr = runs(d.tolist())
                print(f"iter {it} rank {r}: {d.size} wrong elements, runs {rr[:8]}{' ...' if len(rr) > 8 else ''}; "
                      f"wrong per source rank {lanes}", flush=True)
                k = int(d[0])
                print(f"   first: elem {k} got {got[k]:08x} want {want[k]:08x} prev-iter {((((it - 1) * 7 + r * count + k) & 0xff) * 0x01010101):08x}",
                      flush=True)
        if it % 10 == 9:
            print(f"iter {it + 1}: {bad} bad results so far", flush=True)
    bed.close()
    print("DONE bad =", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
