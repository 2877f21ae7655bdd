"""GPU parity of the reduce-copy core and the one-rank path against the CPU oracle.
All calls go through the C ABI (rcclAmdReduceCopy, ncclAllReduce ...) via ctypes."""
import os

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu

INT_TYPES = [orc.I8, orc.U8, orc.I32, orc.U32, orc.I64, orc.U64]
FLT_TYPES = [orc.F16, orc.F32, orc.F64, orc.BF16, orc.F8E4M3, orc.F8E5M2]


def _rand(dt, n, seed):
    rng = np.random.default_rng(seed)
    if dt in INT_TYPES:
        info = np.iinfo(orc.NP_OF[dt])
        return rng.integers(info.min, info.max, size=n, dtype=orc.NP_OF[dt], endpoint=True)
    x = rng.uniform(-1, 1, n).astype(np.float32)
    if dt == orc.F32:
        return x
    if dt == orc.F64:
        return rng.uniform(-1, 1, n)
    if dt == orc.F16:
        return x.astype(np.float16)
    if dt in (orc.F8E4M3, orc.F8E5M2):
        codes = rng.integers(0, 256, size=n, dtype=np.uint8)
        codes[codes == 0x80] = 0                     # 0x80 is NaN in this encoding
        return codes
    return orc.f32_to_bf16_bits(x)


def _to_dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).copy()).cuda()


@pytest.mark.parametrize("dt", INT_TYPES + FLT_TYPES)
@pytest.mark.parametrize("op", [orc.SUM, orc.PROD, orc.MAX, orc.MIN])
def test_reduce_copy_matches_oracle(dt, op):
    import torch
    import rccl_amd as ra
    es = np.dtype(orc.NP_OF[dt]).itemsize
    for nsrc, n in ((1, 1000), (2, 4099), (3, 70001), (4, 65536), (8, 300007), (5, 17)):
        ins = [_rand(dt, n, 100 * nsrc + r) for r in range(nsrc)]
        dev = [_to_dev(a) for a in ins]
        out = torch.full((n * es,), 0x5A, dtype=torch.uint8, device="cuda")
        ra.reduce_copy(out, dev, n, dt, op)
        torch.cuda.synchronize()
        want = orc.device_fold(dt, op, ins)
        got = out.cpu().numpy().view(orc.NP_OF[dt])
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (orc.DT_NAMES[dt], orc.OP_NAMES[op], nsrc, n)


@pytest.mark.parametrize("dt", [orc.U8, orc.F16, orc.F32, orc.I64])
def test_reduce_copy_unaligned_pointers(dt):
    """Pointers off 16-byte alignment take the element path (common_kernel.h:214-239 in the reference)."""
    import torch
    import rccl_amd as ra
    es = np.dtype(orc.NP_OF[dt]).itemsize
    n = 10007
    ins = [_rand(dt, n, 7 + r) for r in range(3)]
    pad = es  # one element of misalignment
    dev = []
    for a in ins:
        t = torch.zeros(n * es + 64, dtype=torch.uint8, device="cuda")
        t[pad:pad + n * es] = _to_dev(a)
        dev.append(t)
    out = torch.zeros(n * es + 64, dtype=torch.uint8, device="cuda")
    ra.reduce_copy(out.data_ptr() + pad, [t.data_ptr() + pad for t in dev], n, dt, orc.SUM)
    torch.cuda.synchronize()
    want = orc.device_fold(dt, orc.SUM, ins)
    host = out.cpu().numpy()
    assert np.array_equal(host[pad:pad + n * es], want.view(np.uint8))
    assert not host[:pad].any() and not host[pad + n * es:].any()   # nothing written outside


def test_one_rank_paths():
    """nRanks == 1: copy out of place, nothing in place, pre-multiply still scales (onerank.cu:55-90)."""
    import torch
    import rccl_amd as ra
    comm = ra.Communicator.init_all(1, [0])[0]
    try:
        assert comm.count() == 1 and comm.user_rank() == 0 and comm.cu_device() == 0
        x = _rand(orc.F32, 1 << 20, 3)
        d_in = _to_dev(x)
        d_out = torch.zeros_like(d_in)
        for op in (ra.ncclSum, ra.ncclMax, ra.ncclAvg):
            d_out.zero_()
            comm.all_reduce(d_in, d_out, x.size, ra.ncclFloat32, op)
            torch.cuda.synchronize()
            assert np.array_equal(d_out.cpu().numpy().view(np.float32), x)
        comm.all_reduce(d_in, d_in, x.size, ra.ncclFloat32, ra.ncclSum)      # in place: untouched
        comm.reduce_scatter(d_in, d_out, x.size, ra.ncclFloat32, ra.ncclSum)
        comm.all_gather(d_in, d_out, x.size, ra.ncclFloat32)
        torch.cuda.synchronize()
        assert np.array_equal(d_in.cpu().numpy().view(np.float32), x)
        assert np.array_equal(d_out.cpu().numpy().view(np.float32), x)
        # custom pre-multiply-sum: out = in * 0.5
        op = comm.create_premul_sum(np.float32(0.5).tobytes(), ra.ncclFloat32)
        comm.all_reduce(d_in, d_out, x.size, ra.ncclFloat32, op)
        torch.cuda.synchronize()
        comm.destroy_op(op)
        assert np.array_equal(d_out.cpu().numpy().view(np.float32), x * np.float32(0.5))
        # count == 0 is a successful no-op (src/enqueue.cc:2078)
        comm.all_reduce(d_in, d_out, 0, ra.ncclFloat32, ra.ncclSum)
        # odd byte count
        comm.all_gather(d_in, d_out, 12345, ra.ncclUint8)
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy()[:12345], x.view(np.uint8)[:12345])
    finally:
        comm.destroy()


def test_argument_errors():
    """Error convention of the boundary (src/misc/argcheck.cc:36-84)."""
    import torch
    import rccl_amd as ra
    comm = ra.Communicator.init_all(1, [0])[0]
    try:
        t = torch.zeros(64, dtype=torch.uint8, device="cuda")
        for bad in (lambda: comm.all_reduce(t, t, 4, 99, ra.ncclSum),
                    lambda: comm.all_reduce(t, t, 4, ra.ncclFloat32, 77),
                    lambda: comm.all_reduce(None, t, 4, ra.ncclFloat32, ra.ncclSum)):
            with pytest.raises(ra.NcclError) as e:
                bad()
            assert e.value.code == ra.ncclInvalidArgument
        assert ra.lib().ncclAllReduce(None, None, 4, ra.ncclFloat32, ra.ncclSum, None, None) == ra.ncclInvalidArgument
        assert ra.lib().ncclSend(None, 0, 0, 0, None, None) == ra.ncclInvalidUsage
    finally:
        comm.destroy()


def test_register_cache_semantics():
    """ncclCommRegister / ncclCommDeregister keep a reference-counted cache per allocation
    (src/register.cc:106-136,163-185): a range inside a registered allocation returns the same handle,
    the entry goes away with the last deregistration, an unknown handle is ncclInvalidUsage."""
    import ctypes
    import torch
    import rccl_amd as ra
    comm = ra.Communicator.init_all(1, [0])[0]
    try:
        t = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
        h1 = comm.register(t.data_ptr(), 4096)
        h2 = comm.register(t.data_ptr() + 8192, 1000)          # same allocation: the same cache entry
        assert h1 and h1 == h2
        comm.deregister(h1)
        comm.deregister(h2)
        with pytest.raises(ra.NcclError) as e:
            comm.deregister(h1)                                # gone with the last reference
        assert e.value.code == ra.ncclInvalidUsage
        with pytest.raises(ra.NcclError) as e:
            comm.deregister(0x1234)
        assert e.value.code == ra.ncclInvalidUsage
        host = ctypes.create_string_buffer(4096)
        with pytest.raises(ra.NcclError) as e:
            comm.register(ctypes.addressof(host), 4096)        # not device memory
        assert e.value.code == ra.ncclInvalidArgument
        with pytest.raises(ra.NcclError) as e:
            comm.register(t.data_ptr(), 0)
        assert e.value.code == ra.ncclInvalidArgument
        # registered buffers do not change the one-rank path
        x = torch.arange(1000, dtype=torch.float32, device="cuda")
        y = torch.zeros_like(x)
        h = comm.register(y.data_ptr(), 4000)
        comm.all_reduce(x, y, 1000, ra.ncclFloat32, ra.ncclSum)
        torch.cuda.synchronize()
        assert torch.equal(x, y)
        comm.deregister(h)
        assert ra.lib().ncclCommRegister(None, None, 0, None) == ra.ncclInvalidArgument
    finally:
        comm.destroy()


def test_launch_mode_group():
    """NCCL_LAUNCH_MODE=GROUP (src/group.cc:128-190): sibling communicators of one process launch in rounds behind a
    barrier, whichever threads drive them.  Runs in a process of its own (the mode is read once per process)."""
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "launch_mode_child.py")
    r = subprocess.run([sys.executable, child], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "launch mode group ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
