#!/usr/bin/env python3
"""Several ranks of one communicator inside ONE process on ONE GPU (ncclCommInitAll with a repeated
device, ncclGroupStart/End, one stream per rank).  No xGMI is involved: all traffic is HBM, so this shows
what the collective kernels and the FIFO protocol cost by themselves, and it can run under rocprofv3
(one process, no launcher).  Not a scaling number.

    python3 tools/loopback_inproc.py --ranks 2 --bytes 1073741824 --iters 10
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--bytes", type=int, default=1 << 30)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--register", action="store_true", help="register the receive buffers (zero-copy gather phase)")
    a = ap.parse_args()
    if a.ranks > 2:
        # every rank's kernel must be running at once; a process gets 4 hardware queues per GPU (GPU_MAX_HW_QUEUES)
        # and streams share them, so with more ranks one rank's kernel can queue behind a kernel that waits for it
        sys.exit("more than 2 ranks inside one process on one GPU can deadlock on shared hardware queues; "
                 "use bench.py under torch.distributed.run (one process per rank) for more")
    os.environ["RCCL_AMD_MULTI_RANK_GPU"] = "1"
    os.environ.setdefault("RCCL_AMD_SPIN_TIMEOUT_MS", "20000")
    # two workgroups per channel and rank on the all-pairs schedule, one on the ring (which two ranks take by default from 24 MiB)
    ring = (os.environ.get("NCCL_ALGO", "").lower() == "ring"
            or (not os.environ.get("NCCL_ALGO") and os.environ.get("RCCL_AMD_TWO_RANK_RING", "1") != "0" and a.bytes >= 24 << 20))
    os.environ.setdefault("RCCL_AMD_NCHANNELS", str(max(2, (256 if ring else 128) // a.ranks)))
    import torch
    import rccl_amd as ra
    n, count = a.ranks, a.bytes // 4
    comms = ra.Communicator.init_all(n, [0] * n)
    streams = [torch.cuda.Stream() for _ in range(n)]
    ins = [torch.rand(count, device="cuda") for _ in range(n)]
    outs = [torch.zeros(count, device="cuda") for _ in range(n)]
    torch.cuda.synchronize()
    handles = [comms[r].register(outs[r].data_ptr(), a.bytes) for r in range(n)] if a.register else []

    def step():
        ra.group_start()
        for r in range(n):
            comms[r].all_reduce(ins[r], outs[r], count, ra.ncclFloat32, ra.ncclSum, streams[r])
        ra.group_end()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for r in range(n):
        e0[r].record(streams[r])
    for _ in range(a.iters):
        step()
    for r in range(n):
        e1[r].record(streams[r])
    torch.cuda.synchronize()
    ms = max(e0[r].elapsed_time(e1[r]) for r in range(n)) / a.iters
    want = ins[0].clone()
    for r in range(1, n):
        want += ins[r]
    ok = all(torch.equal(o, want) for o in outs) and all(c.async_error() == 0 for c in comms)
    # bytes all ranks move through HBM per all-reduce: input read and output written once, every scattered piece
    # written to a window and read back; the same again for the gathered pieces unless they are written directly
    hbm = (2 + (2 if a.register else 4) * (n - 1) / n) * a.bytes * n
    for r, h in enumerate(handles):
        comms[r].deregister(h)
    print(json.dumps({"ranks_on_one_gpu": n, "registered": bool(handles), "gather_phase_pieces": comms[0].stats(), "step_timers": comms[0].step_timers(), "bytes": a.bytes, "ms_per_allreduce": round(ms, 4), "exact": ok,
                      "plan": comms[0].plan_info(0, count, ra.ncclFloat32),
                      "hbm_bytes_all_ranks": hbm, "hbm_GBps": round(hbm / ms / 1e6, 1)}))
    for c in comms:
        c.destroy()


if __name__ == "__main__":
    main()
