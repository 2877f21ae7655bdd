#!/usr/bin/env python3
"""Local N-source reduce-copy (rcclAmdReduceCopy) rate per element type: output bytes per second and HBM bytes per
second ((N + 1) x output), 256 MiB of output, sum.  Shows what the software 8-bit float conversion costs next to the
types the hardware adds natively."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import rccl_amd as ra
    nbytes = 256 << 20
    rows = []
    for name, dt, es in (("fp32", ra.ncclFloat32, 4), ("bf16", ra.ncclBfloat16, 2), ("fp16", ra.ncclFloat16, 2),
                         ("int8", ra.ncclInt8, 1), ("fp8e4m3", ra.ncclFp8E4M3, 1), ("fp8e5m2", ra.ncclFp8E5M2, 1)):
        for nsrc in (2, 8):
            srcs = [torch.randint(0, 120, (nbytes,), dtype=torch.uint8, device="cuda") for _ in range(nsrc)]
            dst = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
            count = nbytes // es
            for _ in range(3):
                ra.reduce_copy(dst, srcs, count, dt, ra.ncclSum)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 10
            e0.record()
            for _ in range(iters):
                ra.reduce_copy(dst, srcs, count, dt, ra.ncclSum)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            rows.append({"dtype": name, "sources": nsrc, "ms": round(ms, 4), "out_GBps": round(nbytes / ms / 1e6, 1),
                         "hbm_GBps": round((nsrc + 1) * nbytes / ms / 1e6, 1)})
            print(rows[-1])
            del srcs, dst
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
