"""rcclAmdAllReduceNoAlloc, the all-reduce that allocates nothing, against ncclAllReduce on a communicator made by
ncclCommInitAll over the same (repeated) device: same inputs, bit-identical outputs, two calls in a row with the
second smaller.  Both go through the C ABI via ctypes."""
import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


def _rand(dt, n, seed):
    rng = np.random.default_rng(seed)
    if dt in (orc.I32, orc.U8, orc.I64):
        info = np.iinfo(orc.NP_OF[dt])
        return rng.integers(info.min, info.max, size=n, dtype=orc.NP_OF[dt], endpoint=True)
    x = rng.uniform(-1, 1, n).astype(np.float32)
    if dt == orc.F32:
        return x
    if dt == orc.F16:
        return x.astype(np.float16)
    return orc.f32_to_bf16_bits(x)


def _to_dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).copy()).cuda()


def test_no_alloc_all_reduce_matches_comm_all_reduce(monkeypatch):
    import torch
    import rccl_amd as ra
    monkeypatch.setenv("RCCL_AMD_MULTI_RANK_GPU", "1")
    monkeypatch.setenv("RCCL_AMD_SPIN_TIMEOUT_MS", "20000")
    n = 2
    comms = ra.Communicator.init_all(n, [0] * n)
    streams = [torch.cuda.Stream() for _ in range(n)]
    try:
        # (type, op, elements): the first call fills the limit, every later one is smaller than the one before
        cases = [(orc.F32, ra.ncclSum, 65536), (orc.F32, ra.ncclSum, 12345), (orc.BF16, ra.ncclAvg, 9001),
                 (orc.I32, ra.ncclAvg, 4097), (orc.U8, ra.ncclMax, 1001), (orc.F16, ra.ncclProd, 513),
                 (orc.I64, ra.ncclMin, 1)]
        for k, (dt, op, count) in enumerate(cases):
            es = np.dtype(orc.NP_OF[dt]).itemsize
            ins = [_rand(dt, count, 1000 + 10 * k + r) for r in range(n)]
            d_in = [_to_dev(a) for a in ins]
            ref = [torch.full((count * es + 64,), 0x5A, dtype=torch.uint8, device="cuda") for _ in range(n)]
            got = [torch.full((count * es + 64,), 0x5A, dtype=torch.uint8, device="cuda") for _ in range(n)]
            torch.cuda.synchronize()
            ra.group_start()
            for r in range(n):
                comms[r].all_reduce(d_in[r], ref[r], count, dt, op, streams[r])
            ra.group_end()
            torch.cuda.synchronize()
            ra.all_reduce_no_alloc(d_in, got, [0] * n, count, dt, op, streams)
            torch.cuda.synchronize()
            for r in range(n):
                a, b = got[r].cpu().numpy(), ref[r].cpu().numpy()
                assert np.array_equal(a, b), (orc.DT_NAMES[dt], op, count, r)       # poison tail included
            if op in (ra.ncclSum, ra.ncclProd, ra.ncclMax, ra.ncclMin):
                want = orc.device_fold(dt, op, ins).view(np.uint8)
                assert np.array_equal(got[0].cpu().numpy()[:count * es], want)
        # in place
        x = [_rand(orc.F32, 777, 50 + r) for r in range(n)]
        d = [_to_dev(a) for a in x]
        ra.all_reduce_no_alloc(d, d, [0] * n, 777, ra.ncclFloat32, ra.ncclSum, streams)
        torch.cuda.synchronize()
        want = orc.device_fold(orc.F32, orc.SUM, x).view(np.uint8)
        for r in range(n):
            assert np.array_equal(d[r].cpu().numpy(), want)
        # beyond the limit: refused with ncclInvalidArgument, nothing launched
        big = torch.zeros(ra.NO_ALLOC_MAX_BYTES + 4, dtype=torch.uint8, device="cuda")
        with pytest.raises(ra.NcclError) as e:
            ra.all_reduce_no_alloc([big] * n, [big] * n, [0] * n, ra.NO_ALLOC_MAX_BYTES // 4 + 1, ra.ncclFloat32,
                                   ra.ncclSum, streams)
        assert e.value.code == ra.ncclInvalidArgument
        with pytest.raises(ra.NcclError):
            ra.all_reduce_no_alloc([big] * 9, [big] * 9, [0] * 9, 4, ra.ncclFloat32, ra.ncclSum, (streams * 5)[:9])
        # two ranks on one stream would never meet: refused, nothing launched
        small = torch.zeros(64, dtype=torch.uint8, device="cuda")
        with pytest.raises(ra.NcclError) as e:
            ra.all_reduce_no_alloc([small] * n, [small] * n, [0] * n, 16, ra.ncclFloat32, ra.ncclSum, [streams[0]] * n)
        assert e.value.code == ra.ncclInvalidUsage
        # ranks on different devices: refused (the staging array is not memory a peer GPU may write)
        with pytest.raises(ra.NcclError) as e:
            ra.all_reduce_no_alloc([small] * n, [small] * n, [0, 1], 16, ra.ncclFloat32, ra.ncclSum, streams)
        assert e.value.code == (ra.ncclInvalidUsage if torch.cuda.device_count() > 1 else ra.ncclInvalidArgument)
        assert ra.no_alloc_async_error(0) == ra.ncclSuccess
        assert all(c.async_error() == 0 for c in comms)
    finally:
        for c in comms:
            c.destroy()


@pytest.mark.parametrize("n", [3, 4])
def test_no_alloc_all_reduce_more_than_two_ranks_on_one_device(n, monkeypatch):
    """3 and 4 ranks of one process on device 0.  Their kernels wait for one another, so each must sit on a hardware
    queue of its own: the streams are created with hipStreamCreateWithPriority, spread over the priority levels.
    Two calls in a row, the second smaller; results against the oracle's rank-order fold.  More than 4 ranks are
    refused (8 ranks of one process on one device were tried: their kernels do not all run at the same time)."""
    import ctypes
    import torch
    import rccl_amd as ra
    monkeypatch.setenv("RCCL_AMD_SPIN_TIMEOUT_MS", "20000")
    hip = ctypes.CDLL("libamdhip64.so")
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
    lo, hi = ctypes.c_int(), ctypes.c_int()
    assert hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)) == 0
    levels = list(range(hi.value, lo.value + 1))          # greatest priority first (numerically lowest)
    streams = []
    for r in range(n):
        s = ctypes.c_void_p()
        assert hip.hipStreamCreateWithPriority(ctypes.byref(s), ctypes.c_uint(1), ctypes.c_int(levels[r % len(levels)])) == 0
        streams.append(s.value)
    try:
        small = torch.zeros(64, dtype=torch.uint8, device="cuda")
        with pytest.raises(ra.NcclError) as e:
            ra.all_reduce_no_alloc([small] * 5, [small] * 5, [0] * 5, 16, ra.ncclFloat32, ra.ncclSum, streams + [0, 1][:5 - n])
        assert e.value.code == ra.ncclInvalidUsage
        for dt, op, oop, count in ((orc.F32, ra.ncclSum, orc.SUM, 65536), (orc.I32, ra.ncclMax, orc.MAX, 3001)):
            es = np.dtype(orc.NP_OF[dt]).itemsize
            ins = [_rand(dt, count, 7000 + 10 * n + r) for r in range(n)]
            d_in = [_to_dev(a) for a in ins]
            got = [torch.full((count * es + 64,), 0x5A, dtype=torch.uint8, device="cuda") for _ in range(n)]
            torch.cuda.synchronize()
            ra.all_reduce_no_alloc(d_in, got, [0] * n, count, dt, op, streams)
            torch.cuda.synchronize()
            assert ra.no_alloc_async_error(0) == ra.ncclSuccess
            want = orc.device_fold(dt, oop, ins).view(np.uint8)
            for r in range(n):
                out = got[r].cpu().numpy()
                assert np.array_equal(out[:count * es], want), (n, orc.DT_NAMES[dt], count, r)
                assert (out[count * es:] == 0x5A).all()
    finally:
        torch.cuda.synchronize()
        for s in streams:
            hip.hipStreamDestroy(ctypes.c_void_p(s))
