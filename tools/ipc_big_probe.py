#!/usr/bin/env python3
"""Platform probe, no rccl_amd involved: how long does hipIpcOpenMemHandle take for large uncached allocations?

    python tools/ipc_big_probe.py [GiB ...]          (default 1 1.5 2 3)

The owner allocates, the opener (another process on the same GPU) maps the handle, touches the first and the last byte
through a kernel and closes it.  Every size has 40 s; a size that does not return ends the probe."""
import ctypes
import multiprocessing as mp
import sys
import time


class _Handle(ctypes.Structure):
    _fields_ = [("reserved", ctypes.c_char * 64)]


def _hip():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipIpcGetMemHandle.argtypes = [ctypes.POINTER(_Handle), ctypes.c_void_p]
    hip.hipIpcOpenMemHandle.argtypes = [ctypes.POINTER(ctypes.c_void_p), _Handle, ctypes.c_uint]
    hip.hipIpcCloseMemHandle.argtypes = [ctypes.c_void_p]
    hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    return hip


def opener(conn):
    hip = _hip()
    hip.hipSetDevice(0)
    while True:
        msg = conn.recv()
        if msg is None:
            return
        raw, nbytes = msg
        h = _Handle()
        ctypes.memmove(ctypes.byref(h), raw, 64)
        p = ctypes.c_void_p()
        t0 = time.time()
        rc = hip.hipIpcOpenMemHandle(ctypes.byref(p), h, 1)
        t1 = time.time()
        ok = None
        if rc == 0:
            buf = (ctypes.c_ubyte * 2)()
            hip.hipMemcpy(ctypes.cast(buf, ctypes.c_void_p), ctypes.c_void_p(p.value), 1, 2)
            hip.hipMemcpy(ctypes.c_void_p(ctypes.addressof(buf) + 1), ctypes.c_void_p(p.value + nbytes - 1), 1, 2)
            ok = (buf[0], buf[1])
            hip.hipIpcCloseMemHandle(p)
        conn.send((rc, t1 - t0, ok))


def main():
    sizes = [float(x) for x in sys.argv[1:]] or [1, 1.5, 2, 3]
    ctx = mp.get_context("spawn")
    parent, child = ctx.Pipe()
    proc = ctx.Process(target=opener, args=(child,), daemon=True)
    proc.start()
    hip = _hip()
    hip.hipSetDevice(0)
    for g in sizes:
        nbytes = int(g * (1 << 30))
        p = ctypes.c_void_p()
        rc = hip.hipExtMallocWithFlags(ctypes.byref(p), nbytes, 3)        # hipDeviceMallocUncached
        if rc != 0:
            print(f"{g} GiB: allocation failed ({rc})", flush=True)
            break
        hip.hipMemset(p, 0x5A, nbytes)
        hip.hipDeviceSynchronize()
        h = _Handle()
        rc = hip.hipIpcGetMemHandle(ctypes.byref(h), p)
        parent.send((bytes(h.reserved) if False else ctypes.string_at(ctypes.byref(h), 64), nbytes))
        if not parent.poll(40):
            print(f"{g} GiB: hipIpcOpenMemHandle did not return within 40 s (get handle rc {rc}); stopping", flush=True)
            proc.kill()
            return 1
        orc, dt, ok = parent.recv()
        print(f"{g} GiB: get handle rc {rc}, open rc {orc} in {dt * 1e3:.1f} ms, first/last byte {ok}", flush=True)
        hip.hipFree(p)
    parent.send(None)
    proc.join(10)
    return 0


if __name__ == "__main__":
    sys.exit(main())
