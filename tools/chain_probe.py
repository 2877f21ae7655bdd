#!/usr/bin/env python3
"""Two-rank chain (NCCL_ALGO=Tree) all-reduce of a one-byte type, repeated: where do wrong bytes land, on which rank,
and what are they (an input, the other rank's input, a value of the previous launch)?
usage: chain_probe.py [DTYPE] [OP] [COUNT] [REPEATS]"""
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
    dt = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    op = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 3145776
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    n = 2
    bed = TestBed(n, env={"RCCL_AMD_SPIN_TIMEOUT_MS": "5000"})
    try:
        bed.init_comms({"NCCL_ALGO": os.environ.get("PROBE_ALGO", "Tree"), "RCCL_AMD_ONESHOT_MAX_BYTES": "0", "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": "0"})
        e = np.dtype(orc.NP_OF[dt]).itemsize
        for it in range(reps):
            ins = [orc.fill_pattern(dt, count, r + 2 * it) for r in range(n)]
            outs = bed.run(COLL_ALLREDUCE, dt, op, count, ins, count * e)
            plan = bed.plan(COLL_ALLREDUCE, count, dt)
            want = ex.allreduce(dt, op, ins, plan)
            for r in range(n):
                got = np.frombuffer(outs[r], dtype=np.uint8)[:want.nbytes].view(want.dtype)
                bad = np.nonzero(got != want)[0]
                if bad.size:
                    i = bad[:8]
                    print(f"run {it} rank {r}: {bad.size} wrong elements, first {int(bad[0])} last {int(bad[-1])}; chunk {plan['chunkElems']} -> round {int(bad[0]) // plan['chunkElems']} offset in round {int(bad[0]) % plan['chunkElems']}")
                    print("   idx ", i.tolist())
                    print("   got ", got[i].tolist(), " want", want[i].tolist(), " in0", ins[0][i].tolist(), " in1", ins[1][i].tolist())
                    b0 = int(bad[0]) * e // 4 * 4
                    word = np.frombuffer(outs[r], dtype=np.uint8)[b0:b0 + 4].view(np.uint32)[0]
                    for name, arr in (("in0", ins[0]), ("in1", ins[1]), ("want", want)):
                        w32 = arr.view(np.uint8)[:arr.nbytes // 4 * 4].view(np.uint32)
                        hits = np.nonzero(w32 == word)[0]
                        print(f"   wrong word {int(word):#010x} at byte {b0} occurs in {name} at dwords {hits[:6].tolist()} ({hits.size} hits)")
                    # bytes around the first wrong element
                    lo = max(0, b0 - 16)
                    print("   out bytes", np.frombuffer(outs[r], dtype=np.uint8)[lo:lo + 48].tolist())
                    print("   want     ", want.view(np.uint8)[lo:lo + 48].tolist())
                else:
                    print(f"run {it} rank {r}: ok")
    finally:
        bed.close()


if __name__ == "__main__":
    main()
