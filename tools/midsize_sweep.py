#!/usr/bin/env python3
"""Time per all-reduce over a range of sizes, two ranks of one process on ONE GPU (loop-back: HBM only, no xGMI),
once per setting of RCCL_AMD_ONESHOT_MAX_BYTES.  Shows where the protocols hand over (low-latency lines, one exchange
through the FIFOs, three-phase all-pairs schedule) and the step in time between adjacent sizes.

    python3 tools/midsize_sweep.py [--iters 200] [--variants default,0]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--variants", default="default,0", help="comma-separated RCCL_AMD_ONESHOT_MAX_BYTES settings")
    ap.add_argument("--sizes", default="262144,524288,1048576,2097152,4194304,8388608,16777216,33554432,67108864,134217728")
    a = ap.parse_args()
    os.environ["RCCL_AMD_MULTI_RANK_GPU"] = "1"
    os.environ.setdefault("RCCL_AMD_SPIN_TIMEOUT_MS", "20000")
    os.environ.setdefault("RCCL_AMD_NCHANNELS", "64")
    import torch
    import rccl_amd as ra
    n = 2
    sizes = [int(x) for x in a.sizes.split(",")]
    streams = [torch.cuda.Stream() for _ in range(n)]
    big = max(sizes) // 4
    ins = [torch.rand(big, device="cuda") for _ in range(n)]
    outs = [torch.zeros(big, device="cuda") for _ in range(n)]
    table = {}
    for v in a.variants.split(","):
        if v == "default":
            os.environ.pop("RCCL_AMD_ONESHOT_MAX_BYTES", None)
        else:
            os.environ["RCCL_AMD_ONESHOT_MAX_BYTES"] = v
        comms = ra.Communicator.init_all(n, [0] * n)
        row = []
        for nbytes in sizes:
            count = nbytes // 4
            iters = max(10, min(a.iters, (1 << 33) // nbytes))

            def step():
                ra.group_start()
                for r in range(n):
                    comms[r].all_reduce(ins[r], outs[r], count, ra.ncclFloat32, ra.ncclSum, streams[r])
                ra.group_end()

            for _ in range(5):
                step()
            torch.cuda.synchronize()
            e0 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
            e1 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
            for r in range(n):
                e0[r].record(streams[r])
            for _ in range(iters):
                step()
            for r in range(n):
                e1[r].record(streams[r])
            torch.cuda.synchronize()
            us = max(e0[r].elapsed_time(e1[r]) for r in range(n)) / iters * 1000
            want = ins[0][:count] + ins[1][:count]
            ok = all(torch.equal(o[:count], want) for o in outs) and all(c.async_error() == 0 for c in comms)
            plan = comms[0].plan_info(0, count, ra.ncclFloat32)
            row.append({"bytes": nbytes, "us": round(us, 1), "proto": plan["proto"], "ch": plan["nChannels"], "exact": ok})
        for c in comms:
            c.destroy()
        table[v] = row
        prev = None
        for e in row:
            ratio = "" if prev is None else f"  x{e['us'] / prev:.2f}"
            print(f"oneshot_max={v:>9} {e['bytes']:>10} B  {e['us']:>8.1f} us  proto {e['proto']} ch {e['ch']:>3} exact {e['exact']}{ratio}")
            prev = e["us"]
    print(json.dumps(table))


if __name__ == "__main__":
    main()
