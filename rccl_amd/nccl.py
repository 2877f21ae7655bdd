"""ctypes mirror of include/rccl_amd.h — same names, argument order and error convention as
the reference's public header (ROCm/rccl src/nccl.h.in).

There is no fallback: if librccl_amd.so is missing or does not load, importing this module
raises.  Buffers are raw device pointers (int) or torch tensors (their data_ptr is used);
streams are hipStream_t handles (int) or torch streams.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RCCL_AMD_LIB") or os.path.join(_HERE, "librccl_amd.so")   # override: A/B runs of two builds

# ncclResult_t (nccl.h.in:50-60)
ncclSuccess, ncclUnhandledCudaError, ncclSystemError, ncclInternalError = 0, 1, 2, 3
ncclInvalidArgument, ncclInvalidUsage, ncclRemoteError, ncclInProgress = 4, 5, 6, 7
# ncclRedOp_t (nccl.h.in:353-358)
ncclSum, ncclProd, ncclMax, ncclMin, ncclAvg, ncclNumOps = 0, 1, 2, 3, 4, 5
# ncclDataType_t (nccl.h.in:364-377)
ncclInt8, ncclUint8, ncclInt32, ncclUint32, ncclInt64, ncclUint64 = 0, 1, 2, 3, 4, 5
ncclFloat16, ncclFloat32, ncclFloat64, ncclBfloat16, ncclFp8E4M3, ncclFp8E5M2 = 6, 7, 8, 9, 10, 11
ncclChar, ncclInt, ncclHalf, ncclFloat, ncclDouble = ncclInt8, ncclInt32, ncclFloat16, ncclFloat32, ncclFloat64
ncclScalarDevice, ncclScalarHostImmediate = 0, 1
NCCL_UNIQUE_ID_BYTES = 128

TYPE_SIZE = {ncclInt8: 1, ncclUint8: 1, ncclInt32: 4, ncclUint32: 4, ncclInt64: 8, ncclUint64: 8,
             ncclFloat16: 2, ncclFloat32: 4, ncclFloat64: 8, ncclBfloat16: 2, ncclFp8E4M3: 1, ncclFp8E5M2: 1}

EXPORTS = [
    "ncclGetVersion", "ncclGetErrorString", "ncclGetLastError", "ncclCommGetAsyncError", "ncclGetUniqueId",
    "ncclCommInitRankConfig", "ncclCommInitRank", "ncclCommInitAll", "ncclCommFinalize", "ncclCommDestroy",
    "ncclCommAbort", "ncclCommCount", "ncclCommCuDevice", "ncclCommUserRank", "ncclMemAlloc", "ncclMemFree",
    "ncclCommRegister", "ncclCommDeregister", "rcclAmdCommStats", "rcclAmdCommStepTimers", "rcclAmdCommNetStats",
    "ncclRedOpCreatePreMulSum", "ncclRedOpDestroy", "ncclAllReduce", "ncclReduceScatter", "ncclAllGather",
    "ncclGroupStart", "ncclGroupEnd", "ncclReduce", "ncclBroadcast", "ncclBcast", "ncclSend", "ncclRecv",
    "ncclCommSplit", "ncclGather", "ncclScatter", "ncclAllToAll", "ncclAllToAllv", "mscclLoadAlgo", "mscclRunAlgo",
    "mscclUnloadAlgo", "rcclAmdReduceCopy", "rcclAmdTopoRings", "rcclAmdTopoParseXml", "rcclAmdPlanInfo",
    "rcclAmdPlanForConfig", "rcclAmdPlanProtoForConfig", "rcclAmdBootstrapAllGather", "rcclAmdAllReduceNoAlloc",
    "rcclAmdNoAllocAsyncError", "rcclAmdTopoFromLinks", "rcclAmdWindowBytesFor",
]
EXPORTS += ["p" + name for name in EXPORTS if name.startswith(("nccl", "msccl"))]      # profiling aliases


class ncclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * NCCL_UNIQUE_ID_BYTES)]


class ncclConfig_t(ctypes.Structure):
    _fields_ = [("size", ctypes.c_size_t), ("magic", ctypes.c_uint), ("version", ctypes.c_uint),
                ("blocking", ctypes.c_int), ("cgaClusterSize", ctypes.c_int), ("minCTAs", ctypes.c_int),
                ("maxCTAs", ctypes.c_int), ("netName", ctypes.c_char_p), ("splitShare", ctypes.c_int)]


class NcclError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().ncclGetErrorString(code).decode()
        last = lib().ncclGetLastError(None).decode()
        super().__init__(f"{where}: {msg} [{code}] {last}")


_lib = None


def lib():
    """The loaded library.  Raises OSError when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} is missing: build it with `make -C rccl_amd/csrc -j8` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_LOCAL)
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L.ncclGetErrorString.restype = ctypes.c_char_p
        L.ncclGetErrorString.argtypes = [ci]
        L.ncclGetLastError.restype = ctypes.c_char_p
        L.ncclGetLastError.argtypes = [vp]
        L.ncclGetVersion.argtypes = [ctypes.POINTER(ci)]
        L.ncclGetUniqueId.argtypes = [ctypes.POINTER(ncclUniqueId)]
        L.ncclCommInitRank.argtypes = [ctypes.POINTER(vp), ci, ncclUniqueId, ci]
        L.ncclCommInitRankConfig.argtypes = [ctypes.POINTER(vp), ci, ncclUniqueId, ci, ctypes.POINTER(ncclConfig_t)]
        L.ncclCommInitAll.argtypes = [ctypes.POINTER(vp), ci, ctypes.POINTER(ci)]
        for f in ("ncclCommFinalize", "ncclCommDestroy", "ncclCommAbort"):
            getattr(L, f).argtypes = [vp]
        for f in ("ncclCommCount", "ncclCommCuDevice", "ncclCommUserRank", "ncclCommGetAsyncError"):
            getattr(L, f).argtypes = [vp, ctypes.POINTER(ci)]
        L.ncclMemAlloc.argtypes = [ctypes.POINTER(vp), sz]
        L.ncclMemFree.argtypes = [vp]
        L.ncclCommRegister.argtypes = [vp, vp, sz, ctypes.POINTER(vp)]
        L.ncclCommDeregister.argtypes = [vp, vp]
        L.rcclAmdCommStepTimers.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
        L.rcclAmdCommNetStats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
        L.rcclAmdCommStats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
        L.ncclRedOpCreatePreMulSum.argtypes = [ctypes.POINTER(ci), vp, ci, ci, vp]
        L.ncclRedOpDestroy.argtypes = [ci, vp]
        L.ncclAllReduce.argtypes = [vp, vp, sz, ci, ci, vp, vp]
        L.ncclReduceScatter.argtypes = [vp, vp, sz, ci, ci, vp, vp]
        L.ncclAllGather.argtypes = [vp, vp, sz, ci, vp, vp]
        L.ncclGroupStart.argtypes = []
        L.ncclGroupEnd.argtypes = []
        L.ncclReduce.argtypes = [vp, vp, sz, ci, ci, ci, vp, vp]
        L.ncclBroadcast.argtypes = [vp, vp, sz, ci, ci, vp, vp]
        L.ncclBcast.argtypes = [vp, sz, ci, ci, vp, vp]
        L.ncclSend.argtypes = [vp, sz, ci, ci, vp, vp]
        L.ncclRecv.argtypes = [vp, sz, ci, ci, vp, vp]
        L.ncclCommSplit.argtypes = [vp, ci, ci, ctypes.POINTER(vp), vp]
        L.rcclAmdReduceCopy.argtypes = [vp, ctypes.POINTER(vp), ci, sz, ci, ci, vp]
        L.rcclAmdAllReduceNoAlloc.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ci), ci, sz, ci, ci,
                                              ctypes.POINTER(vp)]
        L.rcclAmdTopoRings.argtypes = [ci, ctypes.POINTER(ci), ci, ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.rcclAmdTopoParseXml.argtypes = [ctypes.c_char_p, ci, ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.rcclAmdPlanInfo.argtypes = [vp, ci, sz, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci),
                                      ctypes.POINTER(sz)]
        L.rcclAmdPlanForConfig.argtypes = [ci, ci, sz, ci, ci, ci, sz, sz, ctypes.POINTER(ci), ctypes.POINTER(ci),
                                           ctypes.POINTER(sz), ctypes.POINTER(sz), ctypes.POINTER(ci)]
        L.rcclAmdPlanProtoForConfig.argtypes = [ci, ci, sz, ci, ci, ci, sz, sz, ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.rcclAmdBootstrapAllGather.argtypes = [ctypes.POINTER(ncclUniqueId), ci, ci, vp, vp, sz]
        L.rcclAmdNoAllocAsyncError.argtypes = [ci, ctypes.POINTER(ci)]
        L.rcclAmdTopoFromLinks.argtypes = [ci, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.rcclAmdWindowBytesFor.argtypes = [ci, ci, ctypes.POINTER(sz), ctypes.POINTER(sz), ctypes.POINTER(sz), ctypes.POINTER(ci)]
        _lib = L
    return _lib


def _check(code, where):
    if code != ncclSuccess:
        raise NcclError(code, where)


def _ptr(x):
    if x is None:
        return None
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    return ctypes.c_void_p(x.data_ptr())      # torch tensor


def _stream(s):
    if s is None:
        import torch
        s = torch.cuda.current_stream()
    if isinstance(s, int):
        return ctypes.c_void_p(s)
    return ctypes.c_void_p(s.cuda_stream)     # torch.cuda.Stream


def get_version():
    v = ctypes.c_int()
    _check(lib().ncclGetVersion(ctypes.byref(v)), "ncclGetVersion")
    return v.value


def get_unique_id():
    """ncclGetUniqueId -> 128 opaque bytes to hand to every rank."""
    uid = ncclUniqueId()
    _check(lib().ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
    return ctypes.string_at(ctypes.byref(uid), NCCL_UNIQUE_ID_BYTES)


def mem_alloc(size):
    """ncclMemAlloc on the current device: fine-grained (uncached) device memory, the type a registered receive
    buffer must have for peers to write into it directly.  Returns the device pointer as an int."""
    p = ctypes.c_void_p()
    _check(lib().ncclMemAlloc(ctypes.byref(p), size), "ncclMemAlloc")
    return p.value


def mem_free(ptr):
    _check(lib().ncclMemFree(ctypes.c_void_p(ptr)), "ncclMemFree")


def group_start():
    _check(lib().ncclGroupStart(), "ncclGroupStart")


def group_end():
    _check(lib().ncclGroupEnd(), "ncclGroupEnd")


class Communicator:
    """One rank's ncclComm_t."""

    def __init__(self, handle):
        self.handle = handle

    @classmethod
    def init_rank(cls, nranks, unique_id, rank, blocking=None, max_ctas=None, min_ctas=None):
        """ncclCommInitRank / ncclCommInitRankConfig on the current HIP device."""
        uid = ncclUniqueId()
        ctypes.memmove(ctypes.byref(uid), unique_id, NCCL_UNIQUE_ID_BYTES)
        h = ctypes.c_void_p()
        if blocking is None and max_ctas is None and min_ctas is None:
            _check(lib().ncclCommInitRank(ctypes.byref(h), nranks, uid, rank), "ncclCommInitRank")
        else:
            undef = -2**31
            cfg = ncclConfig_t(ctypes.sizeof(ncclConfig_t), 0xcafebeef, 22105, undef if blocking is None else int(blocking), undef,
                               undef if min_ctas is None else min_ctas, undef if max_ctas is None else max_ctas, None, undef)
            blocking = True if blocking is None else blocking
            rc = lib().ncclCommInitRankConfig(ctypes.byref(h), nranks, uid, rank, ctypes.byref(cfg))
            if not (rc == ncclInProgress and not blocking):     # non-blocking: poll wait_ready() / async_error()
                _check(rc, "ncclCommInitRankConfig")
        return cls(h)

    def wait_ready(self, timeout_s=120.0):
        """Polls ncclCommGetAsyncError until a non-blocking creation has ended (test/common/TestBedChild.cpp:40-55).
        Returns the number of polls that said ncclInProgress."""
        import time
        t0, polls = time.time(), 0
        while True:
            st = self.async_error()
            if st != ncclInProgress:
                _check(st, "ncclCommGetAsyncError")
                return polls
            polls += 1
            if time.time() - t0 > timeout_s:
                raise TimeoutError("communicator creation still in progress")
            time.sleep(0.0005)

    def split(self, color, key, max_ctas=None):
        """ncclCommSplit: the new communicator, or None for NCCL_SPLIT_NOCOLOR (-1).  max_ctas: ncclConfig_t::maxCTAs of the
        new communicator (bounds its channels, and with them its window)."""
        h = ctypes.c_void_p()
        cfg = None
        if max_ctas is not None:
            undef = -2**31
            cfg = ctypes.byref(ncclConfig_t(ctypes.sizeof(ncclConfig_t), 0xcafebeef, 22105, undef, undef, undef, max_ctas, None, undef))
        rc = lib().ncclCommSplit(self.handle, color, key, ctypes.byref(h), cfg)
        if rc != ncclInProgress:
            _check(rc, "ncclCommSplit")
        return Communicator(h) if h.value else None

    @classmethod
    def init_all(cls, ndev, devlist=None):
        """ncclCommInitAll: one communicator per entry of devlist, all in this process."""
        hs = (ctypes.c_void_p * ndev)()
        dl = None if devlist is None else (ctypes.c_int * ndev)(*devlist)
        _check(lib().ncclCommInitAll(hs, ndev, dl), "ncclCommInitAll")
        return [cls(ctypes.c_void_p(h)) for h in hs]

    def _int(self, fn):
        v = ctypes.c_int()
        _check(getattr(lib(), fn)(self.handle, ctypes.byref(v)), fn)
        return v.value

    def count(self):
        return self._int("ncclCommCount")

    def user_rank(self):
        return self._int("ncclCommUserRank")

    def cu_device(self):
        return self._int("ncclCommCuDevice")

    def async_error(self):
        return self._int("ncclCommGetAsyncError")

    def all_reduce(self, sendbuff, recvbuff, count, datatype, op=ncclSum, stream=None):
        _check(lib().ncclAllReduce(_ptr(sendbuff), _ptr(recvbuff), count, datatype, op, self.handle, _stream(stream)),
               "ncclAllReduce")

    def reduce_scatter(self, sendbuff, recvbuff, recvcount, datatype, op=ncclSum, stream=None):
        _check(lib().ncclReduceScatter(_ptr(sendbuff), _ptr(recvbuff), recvcount, datatype, op, self.handle,
                                       _stream(stream)), "ncclReduceScatter")

    def all_gather(self, sendbuff, recvbuff, sendcount, datatype, stream=None):
        _check(lib().ncclAllGather(_ptr(sendbuff), _ptr(recvbuff), sendcount, datatype, self.handle, _stream(stream)),
               "ncclAllGather")

    def broadcast(self, sendbuff, recvbuff, count, datatype, root, stream=None):
        _check(lib().ncclBroadcast(_ptr(sendbuff), _ptr(recvbuff), count, datatype, root, self.handle, _stream(stream)),
               "ncclBroadcast")

    def reduce(self, sendbuff, recvbuff, count, datatype, op, root, stream=None):
        _check(lib().ncclReduce(_ptr(sendbuff), _ptr(recvbuff), count, datatype, op, root, self.handle, _stream(stream)),
               "ncclReduce")

    def create_premul_sum(self, scalar_bytes, datatype, residence=ncclScalarHostImmediate, device_ptr=None):
        """ncclRedOpCreatePreMulSum; scalar_bytes holds the scalar in `datatype`."""
        op = ctypes.c_int()
        if residence == ncclScalarHostImmediate:
            buf = ctypes.create_string_buffer(bytes(scalar_bytes), 8)
            p = ctypes.cast(buf, ctypes.c_void_p)
        else:
            p = ctypes.c_void_p(device_ptr)
        _check(lib().ncclRedOpCreatePreMulSum(ctypes.byref(op), p, datatype, residence, self.handle),
               "ncclRedOpCreatePreMulSum")
        return op.value

    def destroy_op(self, op):
        _check(lib().ncclRedOpDestroy(op, self.handle), "ncclRedOpDestroy")

    def plan_info(self, coll, count, datatype):
        a, p, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        ch = ctypes.c_size_t()
        _check(lib().rcclAmdPlanInfo(self.handle, coll, count, datatype, ctypes.byref(a), ctypes.byref(p),
                                     ctypes.byref(c), ctypes.byref(ch)), "rcclAmdPlanInfo")
        return {"algo": a.value, "proto": p.value, "nChannels": c.value, "chunkElems": ch.value}

    def register(self, buff, size):
        """ncclCommRegister: returns the opaque handle to pass to deregister()."""
        h = ctypes.c_void_p()
        _check(lib().ncclCommRegister(self.handle, _ptr(buff), size, ctypes.byref(h)), "ncclCommRegister")
        return h.value

    def deregister(self, handle):
        _check(lib().ncclCommDeregister(self.handle, ctypes.c_void_p(handle)), "ncclCommDeregister")

    def stats(self):
        d, f = ctypes.c_uint64(), ctypes.c_uint64()
        _check(lib().rcclAmdCommStats(self.handle, ctypes.byref(d), ctypes.byref(f)), "rcclAmdCommStats")
        return {"direct_pieces": d.value, "fifo_pieces": f.value}

    def net_stats(self):
        """rcclAmdCommNetStats: proxy-thread timers of the host-staged transport (ms) and counters; {} without it."""
        v = (ctypes.c_uint64 * 12)()
        _check(lib().rcclAmdCommNetStats(self.handle, v), "rcclAmdCommNetStats")
        if not any(v):
            return {}
        return {"tx_send_data_ms": v[0] / 1e6, "tx_send_credit_ms": v[1] / 1e6, "tx_idle_ms": v[2] / 1e6, "tx_total_ms": v[3] / 1e6,
                "tx_messages": v[4], "payload_bytes": v[5], "tx_channel_scans": v[6], "tx_scans_skipped": v[7],
                "rx_recv_data_ms": v[8] / 1e6, "rx_poll_ms": v[9] / 1e6, "rx_total_ms": v[10] / 1e6, "rx_messages": v[11]}

    def step_timers(self):
        """With RCCL_AMD_TUNE_FLAGS=32: microseconds per step and role spent waiting / in the body / draining / posting."""
        t = (ctypes.c_uint64 * 15)()
        _check(lib().rcclAmdCommStepTimers(self.handle, t), "rcclAmdCommStepTimers")
        out = {}
        for role, name in enumerate(("role0", "role1", "role2")):
            steps = t[role * 5 + 4]
            if steps:
                out[name] = {k: round(t[role * 5 + i] / steps / 100.0, 2) for i, k in enumerate(("wait_us", "body_us", "drain_us", "post_us"))}
                out[name]["steps"] = steps
        return out

    def finalize(self):
        _check(lib().ncclCommFinalize(self.handle), "ncclCommFinalize")

    def destroy(self):
        if self.handle:
            _check(lib().ncclCommDestroy(self.handle), "ncclCommDestroy")
            self.handle = None

    def abort(self):
        if self.handle:
            _check(lib().ncclCommAbort(self.handle), "ncclCommAbort")
            self.handle = None


def reduce_copy(dst, srcs, count, datatype, op=ncclSum, stream=None):
    """rcclAmdReduceCopy: dst = fold of srcs (device pointers or tensors) on the current device."""
    arr = (ctypes.c_void_p * len(srcs))(*[_ptr(s).value for s in srcs])
    _check(lib().rcclAmdReduceCopy(_ptr(dst), arr, len(srcs), count, datatype, op, _stream(stream)),
           "rcclAmdReduceCopy")


NO_ALLOC_MAX_BYTES = 256 << 10      # payload limit per rank of rcclAmdAllReduceNoAlloc
NO_ALLOC_MAX_RANKS = 8


def all_reduce_no_alloc(sendbuffs, recvbuffs, devices, count, datatype, op=ncclSum, streams=None):
    """rcclAmdAllReduceNoAlloc: rank r reduces sendbuffs[r] (on devices[r]) into recvbuffs[r] on streams[r]; no
    communicator, no allocation (the staging lines are a static array on each GPU)."""
    n = len(devices)
    sb = (ctypes.c_void_p * n)(*[_ptr(b).value for b in sendbuffs])
    rb = (ctypes.c_void_p * n)(*[_ptr(b).value for b in recvbuffs])
    dv = (ctypes.c_int * n)(*[int(d) for d in devices])
    st = (ctypes.c_void_p * n)(*[_stream(x).value for x in streams])
    _check(lib().rcclAmdAllReduceNoAlloc(sb, rb, dv, n, count, datatype, op, st), "rcclAmdAllReduceNoAlloc")


def no_alloc_async_error(device=0):
    """rcclAmdNoAllocAsyncError: ncclSuccess, or ncclRemoteError once a kernel of a no-alloc call gave up on `device`."""
    err = ctypes.c_int()
    _check(lib().rcclAmdNoAllocAsyncError(device, ctypes.byref(err)), "rcclAmdNoAllocAsyncError")
    return err.value


def topo_from_links(link_type, hops):
    """rcclAmdTopoFromLinks: (adjacency matrix, all pairs direct?) from n x n link types and hop counts."""
    n = len(link_type)
    t = (ctypes.c_uint * (n * n))(*[int(x) for row in link_type for x in row])
    h = (ctypes.c_uint * (n * n))(*[int(x) for row in hops for x in row])
    adj = (ctypes.c_int * (n * n))()
    allp = ctypes.c_int()
    _check(lib().rcclAmdTopoFromLinks(n, t, h, adj, ctypes.byref(allp)), "rcclAmdTopoFromLinks")
    return [[adj[i * n + j] for j in range(n)] for i in range(n)], bool(allp.value)


def window_bytes_for(nranks, channels):
    """rcclAmdWindowBytesFor: dict(total, fifo, exchange, channels) of the per-rank window."""
    tot, fifo, exch = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    ch = ctypes.c_int()
    _check(lib().rcclAmdWindowBytesFor(nranks, channels, ctypes.byref(tot), ctypes.byref(fifo), ctypes.byref(exch), ctypes.byref(ch)),
           "rcclAmdWindowBytesFor")
    return dict(total=tot.value, fifo=fifo.value, exchange=exch.value, channels=ch.value)


def topo_rings(ngpus, adj, max_rings=64):
    """rcclAmdTopoRings: list of rings (each a list of ngpus ranks) for a 0/1 adjacency matrix."""
    flat = (ctypes.c_int * (ngpus * ngpus))(*[int(x) for row in adj for x in row])
    out = (ctypes.c_int * (max_rings * ngpus))()
    n = ctypes.c_int()
    _check(lib().rcclAmdTopoRings(ngpus, flat, max_rings, out, ctypes.byref(n)), "rcclAmdTopoRings")
    return [[out[i * ngpus + j] for j in range(ngpus)] for i in range(n.value)]


def topo_parse_xml(text, max_gpus=16):
    """rcclAmdTopoParseXml: xGMI adjacency matrix of a topology XML in the reference's format."""
    adj = (ctypes.c_int * (max_gpus * max_gpus))()
    n = ctypes.c_int()
    _check(lib().rcclAmdTopoParseXml(text.encode(), max_gpus, adj, ctypes.byref(n)), "rcclAmdTopoParseXml")
    k = n.value
    return [[adj[i * k + j] for j in range(k)] for i in range(k)]


def plan_for_config(nranks, coll, count, datatype, algo=-1, max_channels=16, slot_bytes=512 << 10, min_chunk_bytes=32 << 10):
    """rcclAmdPlanForConfig: the planner's decision for explicit settings (no GPU needed)."""
    a, c, g = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    ch, last = ctypes.c_size_t(), ctypes.c_size_t()
    _check(lib().rcclAmdPlanForConfig(nranks, coll, count, datatype, algo, max_channels, slot_bytes, min_chunk_bytes,
                                      ctypes.byref(a), ctypes.byref(c), ctypes.byref(ch), ctypes.byref(last),
                                      ctypes.byref(g)), "rcclAmdPlanForConfig")
    return {"algo": a.value, "nChannels": c.value, "chunkElems": ch.value, "lastChunkElems": last.value, "grid": g.value}


def plan_proto_for_config(nranks, coll, count, datatype, algo=-1, allow_ll=1, one_shot_max_bytes=0, two_exchange_max_bytes=0):
    """rcclAmdPlanProtoForConfig: (algo, proto) a collective takes for explicit settings (no GPU needed)."""
    a, p = ctypes.c_int(), ctypes.c_int()
    _check(lib().rcclAmdPlanProtoForConfig(nranks, coll, count, datatype, algo, allow_ll, one_shot_max_bytes,
                                           two_exchange_max_bytes, ctypes.byref(a), ctypes.byref(p)), "rcclAmdPlanProtoForConfig")
    return a.value, p.value


def bootstrap_all_gather(unique_id, rank, nranks, mine):
    """rcclAmdBootstrapAllGather: every rank's `mine` (bytes), in rank order."""
    uid = ncclUniqueId()
    ctypes.memmove(ctypes.byref(uid), unique_id, NCCL_UNIQUE_ID_BYTES)
    out = ctypes.create_string_buffer(len(mine) * nranks)
    _check(lib().rcclAmdBootstrapAllGather(ctypes.byref(uid), rank, nranks, mine, out, len(mine)),
           "rcclAmdBootstrapAllGather")
    return [out.raw[i * len(mine):(i + 1) * len(mine)] for i in range(nranks)]
