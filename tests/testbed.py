"""Multi-process test bed: one child process per rank, driven over pipes by a command enum.

Modelled on the reference's own test bed (test/common/TestBed.cpp:614-765 and
test/common/TestBedChild.cpp:119-130: the parent forks one child per rank up front and
sends GET_UNIQUE_ID / INIT_COMMS / SET_COLL_ARGS / EXECUTE_COLL / VALIDATE commands).
Every child owns one communicator created with ncclCommInitRank; several ranks may share
one GPU (the reference's ranks-per-GPU mode), which is how the inter-rank protocol is
exercised on a one-GPU box.

Children import torch and the HIP library; the parent only needs numpy.
"""
import ctypes
import multiprocessing as mp
import os
import sys
import traceback

import numpy as np

COLL_ALLREDUCE, COLL_REDUCESCATTER, COLL_ALLGATHER, COLL_BROADCAST, COLL_REDUCE = 0, 1, 2, 3, 4


def ctypes_hip():
    if _DevBuf._hip is None:
        _DevBuf._hip = ctypes.CDLL("libamdhip64.so")
    return _DevBuf._hip


class _DevBuf:
    """A device buffer that stays at one address across commands.  `fine` = ncclMemAlloc (fine-grained memory, what a
    peer may write into), moved with hipMemcpy / hipMemset; otherwise an ordinary torch tensor (coarse-grained memory
    from torch's caching allocator), moved by torch."""

    _hip = None
    staging = None          # diagnostic only (RCCL_AMD_BED_CHECK_READBACK): a device buffer as old as the process
    # Coarse-grained test buffers are cut from one tensor that is as old as the process.  A coarse buffer that torch has
    # just obtained from hipMalloc, in a process that has allocated and freed other buffers before, intermittently reads
    # back through the copy engine with zeros or an earlier fill in 128-byte-aligned runs while a kernel reads the right
    # bytes at the same addresses, or the other way round (RCCL_AMD_BED_CHECK_READBACK=1 shows the two views side by
    # side; tools/fuzz_collectives.py with and without --arena-mib: 3 mismatches against 0 in the same 750 cases, in
    # three different collectives and schedules).  Allocations that live as long as the process have never shown it.
    arena = None
    arena_used = 0

    def __init__(self, ra, nbytes, fine=False, fill=None, image=None, arena=False):
        import torch
        ctypes_hip()
        if image is not None:
            image = np.ascontiguousarray(image).view(np.uint8).ravel()
            nbytes = image.size
        self.ra, self.n, self.fine, self.torch = ra, int(nbytes), fine, torch
        if fine:
            self.p = ra.mem_alloc(max(self.n, 16))
            self.t = None
        elif arena and _DevBuf.arena is not None and _DevBuf.arena_used + self.n + 4096 <= _DevBuf.arena.numel():
            lo = _DevBuf.arena_used
            _DevBuf.arena_used = (lo + max(self.n, 16) + 4095) // 4096 * 4096
            self.t = _DevBuf.arena[lo:lo + max(self.n, 16)]
            self.p = self.t.data_ptr()
        elif image is not None and self.n:
            self.t = torch.from_numpy(image).cuda()       # one host-to-device copy
            torch.cuda.synchronize()
            self.p = self.t.data_ptr()
            return
        else:
            self.t = torch.empty(max(self.n, 16), dtype=torch.uint8, device="cuda")
            self.p = self.t.data_ptr()
        if fill is not None:
            self.fill(fill)
        if image is not None and self.n:
            self.write(0, image)

    def data_ptr(self):
        return self.p

    def numel(self):
        return self.n

    def fill(self, value, off=0, n=None):
        n = self.n - off if n is None else n
        if self.t is not None:
            self.t[off:off + n].fill_(value)
            self.torch.cuda.synchronize()
            return
        rc = self._hip.hipMemset(ctypes.c_void_p(self.p + off), ctypes.c_int(value), ctypes.c_size_t(n))
        rc = rc or self._hip.hipDeviceSynchronize()
        if rc != 0:
            raise RuntimeError(f"hipMemset failed with {rc}")

    def write(self, off, image):
        image = np.ascontiguousarray(image).view(np.uint8).ravel()
        if self.t is not None:
            self.t[off:off + image.size].copy_(self.torch.from_numpy(image))
            self.torch.cuda.synchronize()
            return
        rc = self._hip.hipMemcpy(ctypes.c_void_p(self.p + off), ctypes.c_void_p(image.ctypes.data),
                                 ctypes.c_size_t(image.size), ctypes.c_int(1))
        if rc != 0:
            raise RuntimeError(f"hipMemcpy to the device failed with {rc}")

    def read(self):
        if self.t is not None:
            self.torch.cuda.synchronize()
            host = self.t[:self.n].cpu().numpy()
            if os.environ.get("RCCL_AMD_BED_CHECK_READBACK") and _DevBuf.staging is not None and self.n <= _DevBuf.staging.numel():
                # diagnostic: the same bytes read by a compute kernel into a long-lived buffer (allocated when the process
                # started) against what the copy engine brought back from this buffer
                st = _DevBuf.staging[:self.n]
                self.torch.add(self.t[:self.n], 0, out=st)
                self.torch.cuda.synchronize()
                via_kernel = st.cpu().numpy()
                if (via_kernel != host).any():
                    bad = np.flatnonzero(via_kernel != host)
                    sys.stderr.write(f"READBACK DIFFERS: the copy engine returned {bad.size} bytes (first at {int(bad[0])}, values "
                                     f"{host[bad[:4]].tolist()}) that differ from what a kernel reads at the same addresses "
                                     f"({via_kernel[bad[:4]].tolist()})\n")
                    sys.stderr.flush()
            return host
        out = np.empty(self.n, dtype=np.uint8)
        rc = self._hip.hipMemcpy(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(self.p), ctypes.c_size_t(self.n),
                                 ctypes.c_int(2))
        if rc != 0:
            raise RuntimeError(f"hipMemcpy from the device failed with {rc}")
        return out

    def free(self):
        if self.t is not None:
            self.t = None
        elif self.p:
            self.ra.mem_free(self.p)
        self.p = 0


class _DevView:
    """A range of a _DevBuf (the in-place receive block of a reduce-scatter): same calls, nothing of its own to free."""

    def __init__(self, buf, off, n):
        self.buf, self.off, self.n = buf, int(off), int(n)

    def data_ptr(self):
        return self.buf.data_ptr() + self.off

    def numel(self):
        return self.n

    def fill(self, value):
        self.buf.fill(value, self.off, self.n)

    def read(self):
        return self.buf.read()[self.off:self.off + self.n]

    def free(self):
        pass


def _child(rank, nranks, device, conn, env):
    try:
        os.environ.update(env)
        import torch
        import rccl_amd as ra
        torch.cuda.set_device(device)
        if os.environ.get("RCCL_AMD_BED_CHECK_READBACK"):
            _DevBuf.staging = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
        if os.environ.get("RCCL_AMD_BED_ARENA_MIB"):
            _DevBuf.arena = torch.empty(int(os.environ["RCCL_AMD_BED_ARENA_MIB"]) << 20, dtype=torch.uint8, device="cuda")
        comm = None
        stream = torch.cuda.Stream()
        while True:
            cmd, args = conn.recv()
            if cmd == "stop":
                break
            if cmd == "ping":
                conn.send(("ok", None))
            elif cmd == "uid":
                conn.send(("ok", ra.get_unique_id()))
            elif cmd == "init":
                uid, init_env = args
                init_env = dict(init_env)
                blocking = init_env.pop("__blocking__", None)
                for k, v in init_env.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
                comm = ra.Communicator.init_rank(nranks, uid, rank, blocking=blocking)
                if blocking is False:
                    comm.wait_ready()
                conn.send(("ok", (comm.count(), comm.user_rank(), comm.cu_device())))
            elif cmd == "split":
                # ncclCommSplit of the bed's communicator, one int32 all-reduce on the new one, then it is destroyed
                color, key, count = args[:3]
                max_ctas = args[3] if len(args) > 3 else None
                sub = comm.split(color, key, max_ctas=max_ctas)
                if sub is None:
                    conn.send(("ok", None))
                else:
                    sub.wait_ready()
                    src_b = _DevBuf(ra, 0, image=(np.arange(count, dtype=np.int32) + 1000 * rank), arena=True)
                    dst_b = _DevBuf(ra, count * 4, fill=0, arena=True)
                    torch.cuda.synchronize()
                    sub.all_reduce(src_b.data_ptr(), dst_b.data_ptr(), count, ra.ncclInt32, ra.ncclSum, stream)
                    stream.synchronize()
                    dst = torch.from_numpy(dst_b.read().view(np.int32).copy())
                    src_b.free(); dst_b.free()
                    _DevBuf.arena_used = 0
                    # (the channels the new communicator would use for a large all-reduce: ncclConfig_t::maxCTAs bounds them)
                    big = sub.plan_info(COLL_ALLREDUCE, 1 << 28, ra.ncclFloat32)["nChannels"] if sub.count() > 1 else 0
                    res = (sub.count(), sub.user_rank(), sub.async_error(), dst.cpu().numpy().tobytes(), big)
                    sub.destroy()
                    conn.send(("ok", res))
            elif cmd == "coll" and args.get("skip"):
                conn.send(("ok", (0, b"")))            # this rank sits the collective out (failure tests)
            elif cmd == "abort":
                if comm is not None:
                    comm.abort()
                    comm = None
                conn.send(("ok", None))
            elif cmd == "try_coll":
                # a collective on a communicator that already failed must return its error code
                t = torch.zeros(64, dtype=torch.float32, device="cuda")
                try:
                    comm.all_reduce(t, t, 64, ra.ncclFloat32, ra.ncclSum, stream)
                    conn.send(("ok", 0))
                except ra.NcclError as e:
                    conn.send(("ok", e.code))
            elif cmd == "coll":
                # args: dict(coll, dtype, op, count, input(np bytes), out_bytes, in_place, premul(bytes|None), iters)
                a = args
                es = ra.TYPE_SIZE[a["dtype"]]
                host_in = np.frombuffer(a["input"], dtype=np.uint8)

                bufs = []

                def upload(image):
                    """One host-to-device copy per buffer (torch memory); it must read back as written."""
                    b = _DevBuf(ra, 0, image=image, arena=True)          # from the start-up arena when the bed has one
                    if not np.array_equal(b.read(), np.ascontiguousarray(image).view(np.uint8).ravel()):
                        raise RuntimeError("test bed: a host-to-device upload does not read back as written")
                    bufs.append(b)
                    return b

                # pad = (bytes in front of sendbuff, bytes in front of recvbuff): unaligned user pointers.  Every
                # buffer ends in a 64-byte guard that must come back untouched.
                pad_s, pad_r = a.get("pad") or (0, 0)
                GUARD = 64
                if a.get("managed"):
                    # hipMallocManaged buffers (the reference's managed-memory test mode, TestBedChild.cpp ALLOCATE_MEM):
                    # filled and read back through the host mapping.
                    hip = ctypes_hip()

                    class Managed:
                        def __init__(self, image):
                            self.n = image.size
                            p = ctypes.c_void_p()
                            rc = hip.hipMallocManaged(ctypes.byref(p), ctypes.c_size_t(max(self.n, 1)), ctypes.c_uint(1))
                            if rc != 0:
                                raise RuntimeError(f"hipMallocManaged failed with {rc}")
                            self.p = p.value
                            ctypes.memmove(self.p, image.ctypes.data, self.n)
                            bufs.append(self)

                        def data_ptr(self):
                            return self.p

                        def read(self):
                            out = np.empty(self.n, dtype=np.uint8)
                            ctypes.memmove(out.ctypes.data, self.p, self.n)
                            return out

                        def fill(self, value, off=0, n=None):
                            ctypes.memset(self.p + off, value, self.n - off if n is None else n)

                        def write(self, off, image):
                            image = np.ascontiguousarray(image).view(np.uint8).ravel()
                            ctypes.memmove(self.p + off, image.ctypes.data, image.size)

                        def free(self):
                            hip.hipFree(ctypes.c_void_p(self.p))

                    upload = Managed
                if a["in_place"]:
                    # buffer sized for the larger side; input placed where the API expects it
                    nbytes = max(host_in.size, a["out_bytes"])
                    image = np.full(pad_r + nbytes + GUARD, 0xA5, dtype=np.uint8)
                    image[pad_r:pad_r + nbytes] = 0
                    off = rank * a["count"] * es if a["coll"] == COLL_ALLGATHER else 0
                    image[pad_r + off:pad_r + off + host_in.size] = host_in
                    buf = upload(image)
                    base = buf.data_ptr() + pad_r
                    if a["coll"] == COLL_ALLGATHER:
                        send_ptr, recv_ptr = base + off, base
                    elif a["coll"] == COLL_REDUCESCATTER:
                        send_ptr, recv_ptr = base, base + rank * a["count"] * es
                    else:
                        send_ptr = recv_ptr = base
                    out_view = lambda: buf
                    guards = [(0, pad_r), (pad_r + nbytes, pad_r + nbytes + GUARD)]
                else:
                    inp = upload(np.concatenate([np.zeros(pad_s, dtype=np.uint8), host_in]))
                    out = upload(np.full(pad_r + a["out_bytes"] + GUARD, 0xA5, dtype=np.uint8))
                    send_ptr, recv_ptr = inp.data_ptr() + pad_s, out.data_ptr() + pad_r
                    out_view = lambda: out
                    guards = [(0, pad_r), (pad_r + a["out_bytes"], pad_r + a["out_bytes"] + GUARD)]
                op = a["op"]
                user_op = None
                scal_dev = scal_next = None
                if a.get("premul") is not None and a.get("premul_device"):
                    # the scalar lives in device memory (ncclScalarDevice): the kernel reads it when it runs
                    scal_dev = upload(np.frombuffer(a["premul"], dtype=np.uint8).copy())
                    if a.get("premul2") is not None:
                        scal_next = np.frombuffer(a["premul2"], dtype=np.uint8).copy()
                    user_op = comm.create_premul_sum(None, a["dtype"], residence=ra.ncclScalarDevice, device_ptr=scal_dev.data_ptr())
                    op = user_op
                elif a.get("premul") is not None:
                    user_op = comm.create_premul_sum(a["premul"], a["dtype"])
                    op = user_op
                torch.cuda.synchronize()

                def issue():
                    if a["coll"] == COLL_ALLREDUCE:
                        comm.all_reduce(send_ptr, recv_ptr, a["count"], a["dtype"], op, stream)
                    elif a["coll"] == COLL_REDUCESCATTER:
                        comm.reduce_scatter(send_ptr, recv_ptr, a["count"], a["dtype"], op, stream)
                    elif a["coll"] == COLL_BROADCAST:
                        comm.broadcast(send_ptr, recv_ptr, a["count"], a["dtype"], a["root"], stream)
                    elif a["coll"] == COLL_REDUCE:
                        comm.reduce(send_ptr, recv_ptr, a["count"], a["dtype"], op, a["root"], stream)
                    else:
                        comm.all_gather(send_ptr, recv_ptr, a["count"], a["dtype"], stream)

                if a.get("graph"):
                    # hipGraph mode of the reference's tests (TestBed.cpp:614-765): capture once, replay twice
                    if a["in_place"]:
                        raise ValueError("graph replays of an in-place reduction would re-reduce its output")
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=stream):
                        issue()
                    with torch.cuda.stream(stream):        # replays on the capture stream: no further hardware queue
                        g.replay()
                        stream.synchronize()
                        out_view().fill(0x3C, pad_r, a["out_bytes"])
                        if scal_next is not None:
                            scal_dev.write(0, scal_next)       # the second replay must pick the new scalar up
                        stream.synchronize()
                        g.replay()
                with torch.cuda.stream(stream):
                    for _ in range(0 if a.get("graph") else a.get("iters", 1)):
                        if a["in_place"] and a.get("iters", 1) > 1:
                            raise ValueError("in-place with iters>1 would re-reduce its own output")
                        issue()
                stream.synchronize()
                if user_op is not None:
                    comm.destroy_op(user_op)
                err = comm.async_error()
                torch.cuda.synchronize()
                res = out_view().read()
                for b in bufs:
                    b.free()
                _DevBuf.arena_used = 0
                for lo, hi in guards:
                    if not np.all(res[lo:hi] == 0xA5):
                        raise RuntimeError(f"rank {rank}: bytes outside the receive buffer were written (guard {lo}:{hi})")
                res = res[pad_r:]
                if a["in_place"]:
                    if a["coll"] == COLL_REDUCESCATTER:
                        off = rank * a["count"] * es
                        res = res[off:off + a["out_bytes"]]
                    else:
                        res = res[:a["out_bytes"]]
                else:
                    res = res[:a["out_bytes"]]
                conn.send(("ok", (err, res.tobytes())))
            elif cmd == "big":
                # A collective at full size.  args: dict(coll, dtype, op, count, seed, in_place, want(path|None), host(path|None)).
                # The bucket is made here from the oracle's hash pattern (element i of rank r depends on (seed, r, i) only),
                # nothing large crosses the pipe: the whole output is compared here, bit for bit, with the expectation the
                # parent computed with the oracle and saved (`want`; all-gather: built here from the same pattern), and
                # within the reference tolerance with the reference's host fold (`host`).
                import oracle as orc
                a = args
                dt, cnt, seed = a["dtype"], a["count"], a["seed"]
                es = ra.TYPE_SIZE[dt]
                n_in = cnt * (nranks if a["coll"] == COLL_REDUCESCATTER else 1)
                n_out = cnt * (nranks if a["coll"] == COLL_ALLGATHER else 1)
                GUARD = 256
                host_in = orc.fill_hashed(dt, n_in, rank, seed).view(np.uint8)
                out = torch.full((n_out * es + GUARD,), 0xA5, dtype=torch.uint8, device="cuda")
                if a.get("in_place"):
                    assert a["coll"] == COLL_ALLREDUCE
                    out[:n_in * es].copy_(torch.from_numpy(host_in))
                    inp = out
                else:
                    inp = torch.from_numpy(host_in).cuda()
                del host_in
                torch.cuda.synchronize()
                with torch.cuda.stream(stream):
                    if a["coll"] == COLL_ALLREDUCE:
                        comm.all_reduce(inp.data_ptr(), out.data_ptr(), cnt, dt, a["op"], stream)
                    elif a["coll"] == COLL_REDUCESCATTER:
                        comm.reduce_scatter(inp.data_ptr(), out.data_ptr(), cnt, dt, a["op"], stream)
                    else:
                        comm.all_gather(inp.data_ptr(), out.data_ptr(), cnt, dt, stream)
                stream.synchronize()
                err = comm.async_error()
                res = out.cpu().numpy()
                del inp, out
                torch.cuda.empty_cache()
                report = {"err": err, "guard_ok": bool(np.all(res[n_out * es:] == 0xA5)), "bad": 0, "first": -1, "host_bad": 0,
                          "host_first": -1, "elems": n_out}
                got = res[:n_out * es].view(orc.NP_OF[dt])
                if a["coll"] == COLL_ALLGATHER:
                    for q in range(nranks):
                        blk = orc.fill_hashed(dt, cnt, q, seed)
                        diff = np.nonzero(got[q * cnt:(q + 1) * cnt].view(np.uint8) != blk.view(np.uint8))[0]
                        if diff.size:
                            report["bad"] += int(diff.size)
                            if report["first"] < 0:
                                report["first"] = q * cnt + int(diff[0]) // es
                else:
                    lo = rank * cnt if a["coll"] == COLL_REDUCESCATTER else 0
                    bits = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[es]
                    want = np.load(a["want"], mmap_mode="r")[lo:lo + n_out]
                    same = np.array_equal(got.view(bits), want.view(bits))
                    diff = np.empty(0, dtype=np.int64) if same else np.nonzero(got.view(bits) != want.view(bits))[0]
                    if diff.size:
                        report["bad"], report["first"] = int(diff.size), int(diff[0])
                        report["got"], report["want"] = got[diff[:4]].tolist(), np.asarray(want[diff[:4]]).tolist()
                    if a.get("host"):
                        host = np.load(a["host"], mmap_mode="r")[lo:lo + n_out]
                        report["host_bad"], report["host_first"] = orc.count_mismatch(dt, got, np.ascontiguousarray(host))
                conn.send(("ok", report))
            elif cmd == "reg_prepare":
                # args: dict(coll, dtype, count, input(bytes), in_place, register_recv, fine): buffers that stay alive across
                # commands.  fine: the receive buffer comes from ncclMemAlloc (what peers may write into), else hipMalloc.
                a = args
                es = ra.TYPE_SIZE[a["dtype"]]
                host_in = np.frombuffer(a["input"], dtype=np.uint8)
                fine = a.get("fine", True)
                reg_what = None
                if a.get("coll", COLL_ALLREDUCE) == COLL_REDUCESCATTER:
                    # the SEND buffer (nranks blocks) is the registered one: the peers' reducers read it directly
                    src = _DevBuf(ra, host_in.size, fine, image=host_in, arena=True)
                    block = host_in.size // nranks
                    if a["in_place"]:
                        reg_out, reg_bufs = _DevView(src, rank * block, block), [src]
                    else:
                        reg_out = _DevBuf(ra, block, fine, fill=0xA5, arena=True)
                        reg_bufs = [reg_out, src]
                    reg_in, reg_what = src.data_ptr(), src
                elif a.get("coll", COLL_ALLREDUCE) == COLL_ALLGATHER:
                    reg_out = _DevBuf(ra, host_in.size * nranks, fine, fill=0xA5, arena=True)
                    if a["in_place"]:
                        reg_out.write(rank * host_in.size, host_in)
                        reg_in = reg_out.data_ptr() + rank * host_in.size
                        reg_bufs = [reg_out]
                    else:
                        src = _DevBuf(ra, host_in.size, False, image=host_in, arena=True)
                        reg_in, reg_bufs = src.data_ptr(), [reg_out, src]
                elif a["in_place"]:
                    reg_out = _DevBuf(ra, host_in.size, fine, image=host_in, arena=True)
                    reg_in, reg_bufs = reg_out.data_ptr(), [reg_out]
                else:
                    src = _DevBuf(ra, host_in.size, False, image=host_in, arena=True)
                    reg_out = _DevBuf(ra, host_in.size, fine, fill=0xA5, arena=True)
                    reg_in, reg_bufs = src.data_ptr(), [reg_out, src]
                torch.cuda.synchronize()
                reg_what = reg_what or reg_out
                reg_handle = comm.register(reg_what.data_ptr(), reg_what.numel()) if a["register_recv"] else None
                reg_state = (a, reg_in, reg_out, reg_handle, reg_bufs)
                conn.send(("ok", None))
            elif cmd == "reg_run":
                a, reg_in, reg_out, reg_handle, reg_bufs = reg_state
                before = comm.stats()
                gather = a.get("coll", COLL_ALLREDUCE) == COLL_ALLGATHER
                scatter = a.get("coll", COLL_ALLREDUCE) == COLL_REDUCESCATTER
                if not a["in_place"]:
                    reg_out.fill(0x5A)          # the owner's poison; no cache sweep follows: peers must still win

                def launch():
                    if scatter:
                        comm.reduce_scatter(reg_in, reg_out.data_ptr(), a["count"], a["dtype"], a["op"], stream)
                    elif gather:
                        comm.all_gather(reg_in, reg_out.data_ptr(), a["count"], a["dtype"], stream)
                    else:
                        comm.all_reduce(reg_in, reg_out.data_ptr(), a["count"], a["dtype"], a["op"], stream)

                if args and args.get("graph"):
                    # capture the launch on the registered buffers and replay it twice (a framework's steady state);
                    # the registrations were mapped by an earlier eager launch
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=stream):
                        launch()
                    with torch.cuda.stream(stream):
                        g.replay()
                        stream.synchronize()
                        if not a["in_place"]:
                            reg_out.fill(0x3C)
                        g.replay()
                else:
                    launch()
                stream.synchronize()
                after = comm.stats()
                conn.send(("ok", (comm.async_error(), reg_out.read().tobytes(),
                                  {k: after[k] - before[k] for k in after})))
            elif cmd == "reg_release":
                a, reg_in, reg_out, reg_handle, reg_bufs = reg_state
                if reg_handle is not None:
                    comm.deregister(reg_handle)
                for bf in reg_bufs:
                    bf.free()
                _DevBuf.arena_used = 0
                reg_state = None
                conn.send(("ok", None))
            elif cmd == "probe_poison":
                # Diagnostic, no communicator involved: fill a buffer with poison, overwrite it with the library's local
                # copy kernel on the side stream, read it back; variants of how the buffer is allocated and how it is
                # read.  Returns {variant: (runs with wrong bytes, wrong bytes, of which poison)}.
                import time
                nbytes, reps = args["nbytes"], args.get("reps", 8)
                rng = np.random.default_rng(1234 + rank)
                image = rng.integers(0, 200, nbytes, dtype=np.uint8)
                image[image == 0x5A] = 0x11
                hip = ctypes_hip()
                src_t = torch.from_numpy(image).cuda()
                torch.cuda.synchronize()
                found = {}

                class _Raw:                       # hipMalloc / ncclMemAlloc with nothing else attached
                    def __init__(self, fine):
                        self.fine = fine
                        if fine:
                            self.p = ra.mem_alloc(nbytes)
                        else:
                            q = ctypes.c_void_p()
                            if hip.hipMalloc(ctypes.byref(q), ctypes.c_size_t(nbytes)) != 0:
                                raise RuntimeError("hipMalloc failed")
                            self.p = q.value

                    def data_ptr(self):
                        return self.p

                    def free(self):
                        if self.fine:
                            ra.mem_free(self.p)
                        else:
                            hip.hipFree(ctypes.c_void_p(self.p))

                kept = _Raw(False)
                for variant in args.get("variants", ("fresh/dma", "fresh/dma-late", "fresh/kernel", "fresh-nofree/dma", "reuse/dma", "torch/dma", "fine-fresh/dma",
                                "fresh/dma/writer-null", "fresh/dma/plainstores")):
                    parts = variant.split("/")
                    alloc, read = parts[0], parts[1]
                    extra = parts[2] if len(parts) > 2 else ""
                    bad_runs = bad_bytes = poison_bytes = 0
                    leak = []
                    hist = []
                    for _ in range(reps):
                        t = None
                        if alloc == "torch":
                            t = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
                            ptr, buf = t.data_ptr(), None
                        elif alloc == "reuse":
                            ptr, buf = kept.data_ptr(), None
                        else:
                            buf = _Raw(alloc.startswith("fine"))
                            ptr = buf.data_ptr()
                        if extra == "settle":
                            time.sleep(0.1)
                        hip.hipMemset(ctypes.c_void_p(ptr), ctypes.c_int(0x5A), ctypes.c_size_t(nbytes))
                        hip.hipDeviceSynchronize()
                        if extra == "plainstores":
                            # a torch kernel as the writer (plain stores) through a tensor that aliases nothing: copy into a
                            # torch tensor is not possible for raw memory, so use hipMemcpyAsync D2D on the side stream
                            hip.hipMemcpyAsync(ctypes.c_void_p(ptr), ctypes.c_void_p(src_t.data_ptr()), ctypes.c_size_t(nbytes), ctypes.c_int(3),
                                               ctypes.c_void_p(stream.cuda_stream))
                        else:
                            ra.reduce_copy(ptr, [src_t], nbytes, ra.ncclInt8, ra.ncclSum, 0 if extra == "writer-null" else stream)
                        stream.synchronize()
                        torch.cuda.synchronize()
                        if read == "dma-late":
                            time.sleep(0.05)
                        if read == "dma-after-sweep":
                            sw = torch.empty(1 << 29, dtype=torch.uint8, device="cuda")
                            sw.fill_(1)
                            int(sw.sum().item())
                            del sw
                            torch.cuda.synchronize()
                        if read == "kernel":
                            out_t = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
                            ra.reduce_copy(out_t, [ptr], nbytes, ra.ncclInt8, ra.ncclSum, stream)
                            stream.synchronize()
                            back = out_t.cpu().numpy()
                        else:
                            back = np.empty(nbytes, dtype=np.uint8)
                            hip.hipMemcpy(ctypes.c_void_p(back.ctypes.data), ctypes.c_void_p(ptr), ctypes.c_size_t(nbytes), ctypes.c_int(2))
                        wrong = back != image
                        nb = int(wrong.sum())
                        bad_runs += nb > 0
                        bad_bytes += nb
                        poison_bytes += int((back[wrong] == 0x5A).sum())
                        if nb:
                            vals, cnts = np.unique(back[wrong], return_counts=True)
                            top = sorted(zip(cnts.tolist(), vals.tolist()), reverse=True)[:3]
                            idx = np.nonzero(wrong)[0]
                            hist.append((top, int(idx[0]), int(idx[-1])))
                        if buf is not None:
                            if alloc.endswith("nofree"):
                                leak.append(buf)
                            else:
                                buf.free()
                    for bf in leak:
                        bf.free()
                    found[variant] = (bad_runs, bad_bytes, poison_bytes, hist[:3])
                kept.free()
                conn.send(("ok", found))
            elif cmd == "second_comm":
                # a second communicator over the same ranks, alive next to the first (a framework's DP + TP groups)
                comm2 = ra.Communicator.init_rank(nranks, args, rank)
                conn.send(("ok", comm2.count()))
            elif cmd == "interleave":
                # args: dict(dtype, op, count, a(bytes), b(bytes)): first(a) second(b) first(b) second(a) on one stream
                a = args
                ta = _DevBuf(ra, 0, image=np.frombuffer(a["a"], dtype=np.uint8), arena=True)
                tb = _DevBuf(ra, 0, image=np.frombuffer(a["b"], dtype=np.uint8), arena=True)
                outs = [_DevBuf(ra, ta.numel(), fill=0xA5, arena=True) for _ in range(4)]
                torch.cuda.synchronize()
                for k, (c, src) in enumerate([(comm, ta), (comm2, tb), (comm, tb), (comm2, ta)]):
                    c.all_reduce(src.data_ptr(), outs[k].data_ptr(), a["count"], a["dtype"], a["op"], stream)
                stream.synchronize()
                errs = (comm.async_error(), comm2.async_error())
                got = [o.read().tobytes() for o in outs]
                for b in [ta, tb] + outs:
                    b.free()
                _DevBuf.arena_used = 0
                conn.send(("ok", (errs, got)))
            elif cmd == "second_destroy":
                comm2.destroy()
                comm2 = None
                conn.send(("ok", None))
            elif cmd == "plan":
                conn.send(("ok", comm.plan_info(*args)))
            elif cmd == "destroy":
                if comm is not None:
                    comm.destroy()
                    comm = None
                conn.send(("ok", None))
            else:
                conn.send(("err", f"unknown command {cmd}"))
    except Exception:  # noqa: BLE001 - report everything to the parent
        try:
            conn.send(("err", traceback.format_exc()))
        except Exception:  # noqa: BLE001
            pass


class TestBed:
    """nranks child processes, rank r on devices[r % len(devices)]."""

    __test__ = False

    def __init__(self, nranks, devices=(0,), env=None, timeout=120):
        self.nranks = nranks
        self.timeout = timeout
        ctx = mp.get_context("spawn")
        # Ranks that share a GPU share its 256 CUs: keep ranks x roles x channels workgroups resident.
        per_gpu = -(-nranks // len(devices))
        # (with 8 processes on one GPU single launches were seen to stall for 0.5-3 s, in steps of about half a second,
        # while the driver rotated the processes' queues: the device-side wait limit leaves room for that)
        # (every rank process makes one 200 MiB tensor when it starts and cuts its test buffers from it: _DevBuf.arena)
        base_env = {"RCCL_AMD_SPIN_TIMEOUT_MS": "30000", "RCCL_AMD_BOOTSTRAP_TIMEOUT_S": "60",
                    "RCCL_AMD_NCHANNELS": str(max(2, min(16, 64 // per_gpu))), "RCCL_AMD_BED_ARENA_MIB": "200"}
        base_env.update(env or {})
        self.base_env = dict(base_env)
        self.conns, self.procs = [], []
        for r in range(nranks):
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_child, args=(r, nranks, devices[r % len(devices)], child, base_env), daemon=True)
            p.start()
            self.conns.append(parent)
            self.procs.append(p)
        self.inited = False
        self.answered = set()

    def _recv(self, r):
        # the first answer of a child includes its `import torch`, which can take minutes on a box that has just started
        limit = self.timeout if r in self.answered else max(self.timeout, 400)
        if not self.conns[r].poll(limit):
            raise TimeoutError(f"rank {r} did not answer within {limit}s")
        self.answered.add(r)
        status, val = self.conns[r].recv()
        if status != "ok":
            raise RuntimeError(f"rank {r} failed:\n{val}")
        return val

    def _all(self, cmd, args_per_rank):
        for r in range(self.nranks):
            self.conns[r].send((cmd, args_per_rank[r]))
        return [self._recv(r) for r in range(self.nranks)]

    def init_comms(self, env=None, blocking=None):
        """(Re)create the communicators; `env` entries are set (None: unset) in every child first.  blocking=False:
        ncclCommInitRankConfig with config.blocking = 0, then ncclCommGetAsyncError until the creation has ended."""
        if blocking is not None:
            env = dict(env or {})
            env["__blocking__"] = blocking
        env = {"RCCL_AMD_ONESHOT_MAX_BYTES": None, "RCCL_AMD_TWO_EXCHANGE_MAX_BYTES": None, "RCCL_AMD_FUSED_ROLES": None, **(env or {})}     # limits set for one test do not leak into the next
        self.destroy_comms()
        if len(self.answered) < self.nranks:
            self._all("ping", [None] * self.nranks)      # every child is past its imports before any rank waits for the others
        self.conns[0].send(("uid", None))
        uid = self._recv(0)
        info = self._all("init", [(uid, env or {})] * self.nranks)
        for r, (count, rank, _dev) in enumerate(info):
            assert count == self.nranks and rank == r
        self.inited = True
        return info

    def run(self, coll, dtype, op, count, inputs, out_bytes, in_place=False, premul=None, iters=1, graph=False,
            skip_ranks=(), expect_errors=False, root=0, pad=None, managed=False, premul_device=False, premul2=None):
        """inputs: one numpy array per rank.  Returns one uint8 array (out_bytes) per rank."""
        if not self.inited:
            self.init_comms()
        args = [dict(coll=coll, dtype=dtype, op=op, count=count, input=np.ascontiguousarray(inputs[r]).tobytes(),
                     out_bytes=out_bytes, in_place=in_place, premul=None if premul is None else premul[r], iters=iters,
                     graph=graph, skip=r in skip_ranks, root=root, pad=pad, managed=managed, premul_device=premul_device,
                     premul2=None if premul2 is None else premul2[r])
                for r in range(self.nranks)]
        res = self._all("coll", args)
        if expect_errors:
            return [err for err, _ in res]
        for r, (err, _) in enumerate(res):
            assert err == 0, f"rank {r}: asynchronous error {err}"
        return [np.frombuffer(b, dtype=np.uint8) for _, b in res]

    def run_big(self, coll, dtype, op, count, seed, want=None, host=None, in_place=False):
        """One collective at full size on buckets every rank makes itself from the oracle's hash pattern; every rank
        compares its whole output with the saved expectation(s) and returns a report dict."""
        if not self.inited:
            self.init_comms()
        return self._all("big", [dict(coll=coll, dtype=dtype, op=op, count=count, seed=seed, want=want, host=host,
                                      in_place=in_place)] * self.nranks)

    def run_registered(self, dtype, op, count, inputs, in_place=False, register_ranks=None, runs=1, coll=COLL_ALLREDUCE,
                       graph=False, fine=True):
        """All-reduce (or all-gather) on buffers that stay allocated (and registered on `register_ranks`, default all) across `runs`
        launches.  Returns, per run, (outputs per rank, stats deltas per rank).  `graph`: every run after the first is
        captured into a hipGraph and replayed twice instead of being launched directly.  `fine`: receive buffers come
        from ncclMemAlloc (fine-grained memory, the type peers may write into); False: ordinary hipMalloc memory, whose
        registration must stay local.  Every out-of-place launch starts from a receive buffer the owner has just filled
        with poison, with no cache maintenance of any kind in between."""
        if not self.inited:
            self.init_comms()
        regs = set(range(self.nranks)) if register_ranks is None else set(register_ranks)
        self._all("reg_prepare", [dict(coll=coll, dtype=dtype, op=op, count=count, input=np.ascontiguousarray(inputs[r]).tobytes(),
                                       in_place=in_place, register_recv=r in regs, fine=fine) for r in range(self.nranks)])
        results = []
        for k in range(runs):
            res = self._all("reg_run", [dict(graph=graph and k > 0)] * self.nranks)
            for r, (err, _, _) in enumerate(res):
                assert err == 0, f"rank {r}: asynchronous error {err}"
            outs = [np.frombuffer(b, dtype=np.uint8) for _, b, _ in res]
            results.append((outs, [st for _, _, st in res]))
            if in_place and coll in (COLL_ALLREDUCE, COLL_REDUCESCATTER):
                break          # a second in-place run would reduce the first one's output
        self._all("reg_release", [None] * self.nranks)
        return results

    def run_two_comms(self, dtype, op, count, ins_a, ins_b):
        """Four all-reduces alternating between the bed's communicator and a second one over the same ranks.
        Returns per rank the four outputs [first(a), second(b), first(b), second(a)]."""
        if not self.inited:
            self.init_comms()
        self.conns[0].send(("uid", None))
        uid = self._recv(0)
        assert self._all("second_comm", [uid] * self.nranks) == [self.nranks] * self.nranks
        res = self._all("interleave", [dict(dtype=dtype, op=op, count=count, a=np.ascontiguousarray(ins_a[r]).tobytes(),
                                            b=np.ascontiguousarray(ins_b[r]).tobytes()) for r in range(self.nranks)])
        self._all("second_destroy", [None] * self.nranks)
        for r, (errs, _) in enumerate(res):
            assert errs == (0, 0), f"rank {r}: asynchronous errors {errs}"
        return [[np.frombuffer(b, dtype=np.uint8) for b in outs] for _, outs in res]

    def split(self, colors, keys, count=4096, max_ctas=None):
        """ncclCommSplit on every rank (colour -1: NCCL_SPLIT_NOCOLOR), then an int32 all-reduce of rank * 1000 + i on each
        new communicator.  Returns per rank None or (new size, new rank, async error, result as int32 array); with
        max_ctas (ncclConfig_t::maxCTAs of the new communicators) also the channels they plan for a 1 GiB all-reduce."""
        if not self.inited:
            self.init_comms()
        res = self._all("split", [(colors[r], keys[r], count, max_ctas) for r in range(self.nranks)])
        if max_ctas is not None:
            return [None if x is None else (x[0], x[1], x[2], np.frombuffer(x[3], dtype=np.int32), x[4]) for x in res]
        return [None if x is None else (x[0], x[1], x[2], np.frombuffer(x[3], dtype=np.int32)) for x in res]

    def command(self, rank, cmd, args=None):
        self.conns[rank].send((cmd, args))
        return self._recv(rank)

    def plan(self, coll, count, dtype):
        self.conns[0].send(("plan", (coll, count, dtype)))
        return self._recv(0)

    def destroy_comms(self):
        if self.inited:
            self._all("destroy", [None] * self.nranks)
            self.inited = False

    def close(self):
        try:
            self.destroy_comms()
        except Exception:  # noqa: BLE001
            pass
        for c in self.conns:
            try:
                c.send(("stop", None))
            except Exception:  # noqa: BLE001
                pass
        for p in self.procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
