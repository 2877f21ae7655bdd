#!/usr/bin/env python3
"""Platform probe, no rccl_amd involved: does `fill -> device copy over it -> synchronize -> read back`
always return the copied data when several processes share one GPU?

    python tools/torch_fill_copy_probe.py [--procs 4] [--iters 3000] [--with-lib]

Each process repeats: inp = host bytes .cuda(); buf = full(0xEE); buf[:n] = inp; synchronize; compare.
--with-lib additionally creates a communicator and runs collectives between the probes.
"""
import argparse
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(rank, nprocs, iters, q, big):
    import torch
    torch.cuda.set_device(0)
    rng = np.random.default_rng(rank)
    sizes = [867760, 51552, 1536, 384, 216940, 419428, 104857, 12 << 20, 20 << 20 if big else 4096]
    bad_total = 0
    for it in range(iters):
        n = sizes[it % len(sizes)]
        host = rng.integers(0, 200, n, dtype=np.uint8)          # never 0xEE
        inp = torch.from_numpy(host).cuda()
        buf = torch.full((n,), 0xEE, dtype=torch.uint8, device="cuda")
        buf[:n] = inp
        torch.cuda.synchronize()
        back = buf.cpu().numpy()
        bad = np.nonzero(back != host)[0]
        if bad.size:
            bad_total += 1
            print(f"proc {rank} iter {it} n {n}: {bad.size} bytes differ, first {bad[0]} last {bad[-1]}, "
                  f"values {back[bad[:6]]}, 128-byte lines {sorted(set((bad // 128).tolist()))[:10]}", flush=True)
    q.put((rank, bad_total))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=4)
    ap.add_argument("--iters", type=int, default=3000)
    ap.add_argument("--big", action="store_true")
    args = ap.parse_args()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=child, args=(r, args.procs, args.iters, q, args.big)) for r in range(args.procs)]
    for p in ps:
        p.start()
    tot = 0
    for _ in ps:
        r, b = q.get(timeout=600)
        tot += b
    for p in ps:
        p.join(timeout=30)
    print("DONE iterations with stale bytes:", tot)
    return 0


if __name__ == "__main__":
    sys.exit(main())
