#!/usr/bin/env python3
"""Platform probe, no rccl_amd involved: when another process on the SAME GPU writes into a hipMalloc'ed buffer
through a HIP IPC mapping, does the owner's next read return what was written?

    python tools/ipc_write_probe.py [--iters 40] [--bytes 8388608] [--pairs 1]

Per iteration the owner uploads 0xA5, overwrites the buffer with 0x5A in a fill kernel and synchronises; the writer
process then copies a pattern (never 0xA5 or 0x5A) over it with an ordinary torch copy kernel through its IPC mapping
and synchronises; the owner reads the buffer back.  Bytes that still show 0x5A or 0xA5 are stale.  This is the access
pattern of several ranks sharing one GPU with registered receive buffers; with one rank per GPU the writer sits on
another GPU and the path through the caches is a different one.
"""
import argparse
import ctypes
import multiprocessing as mp
import sys

import numpy as np


class _Raw:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class _Handle(ctypes.Structure):
    _fields_ = [("reserved", ctypes.c_char * 64)]


def _hip():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipIpcGetMemHandle.argtypes = [ctypes.POINTER(_Handle), ctypes.c_void_p]
    hip.hipIpcOpenMemHandle.argtypes = [ctypes.POINTER(ctypes.c_void_p), _Handle, ctypes.c_uint]     # handle by value
    hip.hipIpcCloseMemHandle.argtypes = [ctypes.c_void_p]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    return hip


def _managed_prelude(hip, torch, nbytes):
    """hipMallocManaged buffers used by a kernel and freed, then every cached torch segment released: what precedes the
    registered-buffer tests when the managed-memory and graph tests ran first in the same processes."""
    for size in (2500, 20000, nbytes // 2, nbytes // 2 + 64):
        p = ctypes.c_void_p()
        assert hip.hipMallocManaged(ctypes.byref(p), ctypes.c_size_t(size), ctypes.c_uint(1)) == 0
        ctypes.memset(p.value, 7, size)
        t = torch.as_tensor(_Raw(p.value, size), device="cuda")
        dev = torch.full((size,), 3, dtype=torch.uint8, device="cuda")
        t.add_(dev)
        torch.cuda.synchronize()
        del t, dev
        hip.hipFree(p)
    torch.cuda.empty_cache()


def _owner(conn, nbytes, iters, managed_first=False, use_torch_alloc=False):
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    hip = _hip()
    if managed_first:
        _managed_prelude(hip, torch, nbytes)
    p = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)) == 0
    handle = _Handle()
    assert hip.hipIpcGetMemHandle(ctypes.byref(handle), p) == 0
    conn.send(bytes(handle))
    own = torch.as_tensor(_Raw(p.value, nbytes), device="cuda")
    first = torch.from_numpy(np.full(nbytes, 0xA5, dtype=np.uint8))
    pattern = (np.arange(nbytes, dtype=np.int64) % 80).astype(np.uint8)           # 0..79: neither 0xA5 nor 0x5A
    stale_iters, stale_bytes, values = 0, 0, set()
    for _ in range(iters):
        own.copy_(first)
        own.fill_(0x5A)
        torch.cuda.synchronize()
        conn.send("go")
        assert conn.recv() == "done"
        back = own.cpu().numpy()
        bad = back != pattern
        if bad.any():
            stale_iters += 1
            stale_bytes += int(bad.sum())
            values.update(np.unique(back[bad]).tolist()[:4])
    conn.send("stop")
    conn.send((stale_iters, stale_bytes, sorted(values)))
    hip.hipFree(p)


def _writer(conn, nbytes, managed_first=False):
    import torch
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    hip = _hip()
    if managed_first:
        _managed_prelude(hip, torch, nbytes)
    # a torch segment of this process, used and handed back to the driver: the IPC mapping below may take over its addresses
    scratch = torch.full((16 << 20,), 9, dtype=torch.uint8, device="cuda")
    scratch.add_(1)
    torch.cuda.synchronize()
    freed = scratch.data_ptr()
    del scratch
    torch.cuda.empty_cache()
    handle = _Handle.from_buffer_copy(conn.recv())
    mapped = ctypes.c_void_p()
    rc = hip.hipIpcOpenMemHandle(ctypes.byref(mapped), handle, 1)
    assert rc == 0, f"hipIpcOpenMemHandle failed with {rc}"
    print(f"writer: mapping at {mapped.value:#x}, the freed torch segment was at {freed:#x}", flush=True)
    dst = torch.as_tensor(_Raw(mapped.value, nbytes), device="cuda")
    src = torch.from_numpy((np.arange(nbytes, dtype=np.int64) % 80).astype(np.uint8)).cuda()
    while conn.recv() == "go":
        dst.copy_(src)
        torch.cuda.synchronize()
        conn.send("done")
    del dst
    hip.hipIpcCloseMemHandle(mapped)


def run_probe(iters=40, nbytes=8 << 20, pairs=1, timeout=40, managed_first=False):
    """Returns (iterations with stale bytes, stale bytes, the stale values seen) summed over `pairs` owner/writer pairs."""
    ctx = mp.get_context("spawn")
    procs, ends = [], []
    for _ in range(pairs):
        o_parent, o_child = ctx.Pipe()
        w_parent, w_child = ctx.Pipe()
        po = ctx.Process(target=_owner, args=(o_child, nbytes, iters, managed_first), daemon=True)
        pw = ctx.Process(target=_writer, args=(w_child, nbytes, managed_first), daemon=True)
        po.start()
        pw.start()
        procs += [po, pw]
        ends.append((o_parent, w_parent))
    for o, w in ends:
        if not o.poll(timeout):
            raise TimeoutError("probe owner did not start")
        w.send(o.recv())
    total = [0, 0, set()]
    live = list(ends)
    while live:
        for o, w in list(live):
            if not o.poll(timeout):
                raise TimeoutError("probe owner stalled")
            msg = o.recv()
            w.send(msg)
            if msg == "stop":
                it, nb, vals = o.recv()
                total[0] += it
                total[1] += nb
                total[2].update(vals)
                live.remove((o, w))
            else:
                if not w.poll(timeout):
                    raise TimeoutError("probe writer stalled")
                o.send(w.recv())
    for p in procs:
        p.join(timeout=30)
    return total[0], total[1], sorted(total[2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--bytes", type=int, default=8 << 20)
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--managed-first", action="store_true")
    args = ap.parse_args()
    it, nb, vals = run_probe(args.iters, args.bytes, args.pairs, managed_first=args.managed_first)
    print(f"PROBE iterations with stale bytes: {it} of {args.iters * args.pairs}; stale bytes {nb}; values {[hex(v) for v in vals]}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
