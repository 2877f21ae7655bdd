#!/usr/bin/env python3
"""Every (element type, operator) kernel instantiation through every schedule once, at sizes of several rounds per
channel, each result compared bit for bit with the CPU oracle.

    python tools/matrix_sweep.py [--ranks 8] [--bytes 3145728] [--configs chain] [--dtypes 0,1] [--repeat 3] [--in-place] [--pad 3]

The reducing kernels are compiled once per element type and operator, so a fault that belongs to one instantiation
(a wrong result, a device-side wait that never ends) shows only when that pair meets the schedule in question; the
randomised run (tools/fuzz_collectives.py) samples the pairs, this one walks them all.  Multi-process test bed, ranks
share one GPU.  Waits are limited to 5 s so that a deadlock ends as a reported failure; the run stops at the first rank that fails.
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
from testbed import COLL_ALLGATHER, COLL_ALLREDUCE, COLL_BROADCAST, COLL_REDUCE, COLL_REDUCESCATTER, TestBed  # noqa: E402

DTYPES = [orc.I8, orc.U8, orc.I32, orc.U32, orc.I64, orc.U64, orc.F16, orc.F32, orc.F64, orc.BF16, orc.F8E4M3, orc.F8E5M2]
OPS = [orc.SUM, orc.PROD, orc.MAX, orc.MIN, orc.AVG]
FIFO = {"RCCL_AMD_ONESHOT_MAX_BYTES": "0", "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": "0", "NCCL_PROTO": None}
CONFIGS = [
    ("all-pairs FIFOs", {"NCCL_ALGO": "Direct", **FIFO}, (COLL_ALLREDUCE, COLL_REDUCESCATTER, COLL_REDUCE)),
    ("all-pairs FIFOs, one workgroup per channel", {"NCCL_ALGO": "Direct", "RCCL_AMD_FUSED_ROLES": "1", **FIFO}, (COLL_ALLREDUCE,)),
    ("ring", {"NCCL_ALGO": "Ring", **FIFO}, (COLL_ALLREDUCE, COLL_REDUCESCATTER)),
    ("chain", {"NCCL_ALGO": "Tree", **FIFO}, (COLL_ALLREDUCE,)),
    ("default choice", {"NCCL_ALGO": None, "RCCL_AMD_ONESHOT_MAX_BYTES": None, "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": None, "NCCL_PROTO": None},
     (COLL_ALLREDUCE, COLL_REDUCESCATTER)),
    # small payloads: the low-latency line protocol (and whatever the planner picks for the other collectives there)
    ("default choice, 160 KiB", {"NCCL_ALGO": None, "RCCL_AMD_ONESHOT_MAX_BYTES": None, "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": None, "NCCL_PROTO": None},
     (COLL_ALLREDUCE, COLL_REDUCESCATTER, COLL_REDUCE), 160 << 10),
    # NCCL_ALGO=Tree with NCCL_PROTO=LL: the line protocol along the chain
    ("chain of lines, 160 KiB", {"NCCL_ALGO": "Tree", "NCCL_PROTO": "LL", "RCCL_AMD_ONESHOT_MAX_BYTES": None, "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": None},
     (COLL_ALLREDUCE,), 160 << 10),
    ("ring of lines, 160 KiB", {"NCCL_ALGO": "Ring", "NCCL_PROTO": "LL", "RCCL_AMD_ONESHOT_MAX_BYTES": None, "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": None},
     (COLL_ALLREDUCE,), 160 << 10),
]
NAMES = {COLL_ALLREDUCE: "all-reduce", COLL_REDUCESCATTER: "reduce-scatter", COLL_REDUCE: "reduce",
         COLL_ALLGATHER: "all-gather", COLL_BROADCAST: "broadcast"}


def es(dt):
    return np.dtype(orc.NP_OF[dt]).itemsize


def one(bed, coll, dt, op, nbytes, in_place=False, pad=None):
    n = bed.nranks
    e = es(dt)
    if coll == COLL_REDUCESCATTER:
        count = nbytes // (e * 2) + 48 // e                  # per rank
        ins = [orc.fill_pattern(dt, count * n, r) for r in range(n)]
        outs = bed.run(coll, dt, op, count, ins, count * e, in_place=in_place, pad=pad)
        plan = bed.plan(coll, count, dt)
        return [ex.check(dt, outs[r], ex.reduce_scatter(dt, op, ins, r, plan)) for r in range(n)], plan
    count = nbytes // e + 48 // e
    ins = [orc.fill_pattern(dt, count, r) for r in range(n)]
    if coll == COLL_ALLREDUCE:
        outs = bed.run(coll, dt, op, count, ins, count * e, in_place=in_place, pad=pad)
        plan = bed.plan(coll, count, dt)
        want = ex.allreduce(dt, op, ins, plan)
        return [ex.check(dt, outs[r], want) for r in range(n)], plan
    root = n - 1
    outs = bed.run(coll, dt, op, count, ins, count * e, root=root, in_place=in_place, pad=pad)
    return [ex.check(dt, outs[root], orc.device_fold(dt, op, ins))], bed.plan(coll, count, dt)


def run(bed, nbytes, configs="", dtypes=None, repeat=1, log=None, in_place=False, pad_elems=0, trace=None):
    """Walks the matrix on `bed`; returns the list of failures (strings).  Stops at the first rank that fails (a
    timeout or a fault): nothing more is started on the GPU after that.  `trace(line)` is called before every case, so
    that a run that dies names the case it died in."""
    failures, done, t0 = [], 0, time.time()
    for name, cfg, colls, *size in [c for c in CONFIGS if configs in c[0]] * repeat:
        bed.init_comms(cfg)
        for coll in colls:
            if coll == COLL_REDUCE and bed.nranks > 8:
                continue
            for dt in dtypes or DTYPES:
                for op in OPS:
                    what = f"{name}: {NAMES[coll]} {orc.DT_NAMES[dt]} {orc.OP_NAMES[op]}"
                    done += 1
                    if trace:
                        trace(what)
                    try:
                        bad, plan = one(bed, coll, dt, op, size[0] if size else nbytes, in_place,
                                        (pad_elems * es(dt), pad_elems * es(dt)) if pad_elems else None)
                    except Exception as exc:  # noqa: BLE001
                        failures.append(f"STOPPED at {what}: {repr(exc)[:300]}")
                        return failures
                    bad = [b for b in bad if b]
                    if bad:
                        failures.append(f"MISMATCH {what} {plan} -> {bad[0][:300]}")
                        if log:
                            log(failures[-1])
            if log:
                log(f"{name}: {NAMES[coll]} done; {done} cases, {len(failures)} failures, {time.time() - t0:.0f} s")
    return failures


def tracer_to(path):
    """A trace function that appends each line to `path` and flushes it to the disk at once."""
    f = open(path, "a")

    def put(line):
        f.write(line + "\n")
        f.flush()
        os.fsync(f.fileno())
    return put


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--bytes", type=int, default=3 << 20, help="payload per rank (reduce-scatter: half of it)")
    ap.add_argument("--configs", default="", help="only the configurations whose name contains this")
    ap.add_argument("--dtypes", default="", help="comma-separated ncclDataType_t values (default: all)")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--in-place", action="store_true")
    ap.add_argument("--pad", type=int, default=0, help="elements in front of both user buffers (unaligned pointers)")
    ap.add_argument("--trace", default="", help="file that receives one line before every case")
    args = ap.parse_args()
    bed = TestBed(args.ranks, env={"RCCL_AMD_SPIN_TIMEOUT_MS": "5000"})
    tracer = tracer_to(args.trace) if args.trace else None
    try:
        failures = run(bed, args.bytes, args.configs, [int(x) for x in args.dtypes.split(",")] if args.dtypes else None,
                       args.repeat, log=lambda m: print(m, flush=True), in_place=args.in_place, pad_elems=args.pad, trace=tracer)
    finally:
        bed.close()
    print(f"DONE, {len(failures)} failures")
    for f in failures:
        print(f)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
