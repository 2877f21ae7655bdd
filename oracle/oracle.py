"""ctypes/numpy front-end of oracle/oracle.c (test infrastructure only).

Arrays are numpy arrays whose dtype carries the element type; f16 uses
np.float16, bf16 is carried as np.uint16 bit patterns (numpy has no bf16) and
tagged by passing dt=BF16 explicitly.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

# ncclDataType_t (reference src/nccl.h.in:364-377)
I8, U8, I32, U32, I64, U64, F16, F32, F64, BF16, F8E4M3, F8E5M2 = range(12)
# ncclRedOp_t (reference src/nccl.h.in:353-358)
SUM, PROD, MAX, MIN, AVG = range(5)

NP_OF = {I8: np.int8, U8: np.uint8, I32: np.int32, U32: np.uint32, I64: np.int64,
         U64: np.uint64, F16: np.float16, F32: np.float32, F64: np.float64, BF16: np.uint16,
         F8E4M3: np.uint8, F8E5M2: np.uint8}
DT_NAMES = {I8: "int8", U8: "uint8", I32: "int32", U32: "uint32", I64: "int64", U64: "uint64",
            F16: "fp16", F32: "fp32", F64: "fp64", BF16: "bf16", F8E4M3: "fp8e4m3", F8E5M2: "fp8e5m2"}
OP_NAMES = {SUM: "sum", PROD: "prod", MAX: "max", MIN: "min", AVG: "avg"}


def build(force=False):
    """Compile liboracle.so with gcc if it is missing or older than its source."""
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        L.orc_type_size.restype = sz
        L.orc_type_size.argtypes = [ci]
        L.orc_fill_pattern.argtypes = [ci, sz, ci, vp]
        L.orc_reduce_into.argtypes = [ci, ci, sz, vp, vp]
        L.orc_divide_by_int.argtypes = [ci, sz, vp, ci]
        L.orc_premul.argtypes = [ci, sz, vp, ci]
        L.orc_host_fold.argtypes = [ci, ci, ci, sz, ctypes.POINTER(vp), vp]
        L.orc_device_fold.argtypes = [ci, ci, ci, sz, ctypes.POINTER(vp), ctypes.POINTER(ci), ci, vp]
        L.orc_ring_allreduce.argtypes = [ci, ci, ci, sz, ctypes.POINTER(vp), ctypes.POINTER(ci), sz, sz, vp]
        L.orc_chain_allreduce.argtypes = [ci, ci, ci, sz, ctypes.POINTER(vp), ctypes.POINTER(ci), vp]
        L.orc_first_mismatch.restype = ctypes.c_long
        L.orc_first_mismatch.argtypes = [ci, sz, vp, vp]
        L.orc_fill_hashed.argtypes = [ci, sz, sz, ci, ctypes.c_uint32, vp]
        L.orc_count_mismatch.restype = ctypes.c_long
        L.orc_count_mismatch.argtypes = [ci, sz, vp, vp, ctypes.POINTER(ctypes.c_long)]
        L.orc_omp_sum_f32.argtypes = [ci, ctypes.POINTER(vp), vp, sz]
        L.orc_ref_style_sum_f32.argtypes = [ci, ctypes.POINTER(vp), vp, sz]
        L.orc_omp_threads.restype = ci
        for name in ("orc_f32_to_bf16", "orc_f32_to_f16"):
            getattr(L, name).restype = ctypes.c_uint16
            getattr(L, name).argtypes = [ctypes.c_float]
        for name in ("orc_bf16_to_f32", "orc_f16_to_f32"):
            getattr(L, name).restype = ctypes.c_float
            getattr(L, name).argtypes = [ctypes.c_uint16]
        for name in ("orc_f32_to_f8e4m3", "orc_f32_to_f8e5m2"):
            getattr(L, name).restype = ctypes.c_uint8
            getattr(L, name).argtypes = [ctypes.c_float]
        for name in ("orc_f8e4m3_to_f32", "orc_f8e5m2_to_f32"):
            getattr(L, name).restype = ctypes.c_float
            getattr(L, name).argtypes = [ctypes.c_uint8]
        L.orc_cvt_f32_to_bf16.argtypes = [sz, vp, vp]
        L.orc_cvt_f32_to_f16.argtypes = [sz, vp, vp]
        L.orc_cvt_f16_to_f32.argtypes = [sz, vp, vp]
        _lib = L
    return _lib


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def _ptr_array(arrs):
    return (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def _int_array(xs):
    return (ctypes.c_int * len(xs))(*[int(x) for x in xs])


def _check(rc, what):
    if rc != 0:
        raise ValueError(f"oracle: {what} failed (unsupported dtype/op?)")


def empty(dt, n):
    return np.empty(int(n), dtype=NP_OF[dt])


def fill_pattern(dt, n, rank):
    """Reference test input of `rank` (test/common/PtrUnion.cpp:135-168)."""
    out = empty(dt, n)
    _check(lib().orc_fill_pattern(dt, n, rank, _ptr(out)), "fill_pattern")
    return out


def fill_hashed(dt, n, rank, seed, first=0):
    """Synthetic bucket for runs at full size: element first+i of `rank` is a hash of (seed, rank, first+i) on a grid the
    element type holds exactly (not a reference pattern: see orc_fill_hashed)."""
    out = empty(dt, n)
    _check(lib().orc_fill_hashed(dt, n, first, rank, seed, _ptr(out)), "fill_hashed")
    return out


def count_mismatch(dt, got, want):
    """(elements outside the reference tolerance, index of the first one or -1)."""
    got = np.ascontiguousarray(got)
    want = np.ascontiguousarray(want)
    assert got.size == want.size and got.dtype.itemsize == want.dtype.itemsize
    first = ctypes.c_long(-1)
    bad = int(lib().orc_count_mismatch(dt, got.size, _ptr(got), _ptr(want), ctypes.byref(first)))
    return bad, int(first.value)


def host_fold(dt, op, inputs):
    """Reference host expected value of an all-reduce (PrepDataFuncs.cpp:95-146)."""
    inputs = [np.ascontiguousarray(a) for a in inputs]
    out = empty(dt, inputs[0].size)
    _check(lib().orc_host_fold(dt, op, len(inputs), out.size, _ptr_array(inputs), _ptr(out)), "host_fold")
    return out


def device_fold(dt, op, inputs, order=None, rank_first=False):
    """Fold in device semantics (float Avg = pre-multiply, int Avg = post-divide)."""
    inputs = [np.ascontiguousarray(a) for a in inputs]
    n = len(inputs)
    order = list(range(n)) if order is None else list(order)
    out = empty(dt, inputs[0].size)
    _check(lib().orc_device_fold(dt, op, n, out.size, _ptr_array(inputs), _int_array(order),
                                 1 if rank_first else 0, _ptr(out)), "device_fold")
    return out


def ring_allreduce(dt, op, inputs, ring, chunk, last_chunk=None):
    """One-channel ring all-reduce in the reference association order (all_reduce.h:71-199)."""
    inputs = [np.ascontiguousarray(a) for a in inputs]
    out = empty(dt, inputs[0].size)
    last_chunk = chunk if last_chunk is None else last_chunk
    _check(lib().orc_ring_allreduce(dt, op, len(inputs), out.size, _ptr_array(inputs), _int_array(ring),
                                    chunk, last_chunk, _ptr(out)), "ring_allreduce")
    return out


def chain_allreduce(dt, op, inputs, chain):
    """Chain-tree all-reduce order, chain[0] = root (all_reduce.h:269-289)."""
    inputs = [np.ascontiguousarray(a) for a in inputs]
    out = empty(dt, inputs[0].size)
    _check(lib().orc_chain_allreduce(dt, op, len(inputs), out.size, _ptr_array(inputs), _int_array(chain),
                                     _ptr(out)), "chain_allreduce")
    return out


def reduce_scatter_expected(dt, op, inputs, rank, fold=host_fold):
    """Rank's block of the full reduction (PrepDataFuncs.cpp:188-263)."""
    n = len(inputs)
    full = fold(dt, op, inputs)
    per = full.size // n
    return full[rank * per:(rank + 1) * per].copy()


def all_gather_expected(inputs):
    """Concatenation in rank order (PrepDataFuncs.cpp:150-186)."""
    return np.concatenate([np.ascontiguousarray(a) for a in inputs])


def first_mismatch(dt, got, want):
    """Index of the first element outside the reference tolerance, or -1 (PtrUnion.cpp:313-342)."""
    got = np.ascontiguousarray(got)
    want = np.ascontiguousarray(want)
    assert got.size == want.size and got.dtype.itemsize == want.dtype.itemsize
    return int(lib().orc_first_mismatch(dt, got.size, _ptr(got), _ptr(want)))


def omp_sum_f32(inputs, out=None):
    inputs = [np.ascontiguousarray(a, dtype=np.float32) for a in inputs]
    if out is None:
        out = np.empty_like(inputs[0])
    lib().orc_omp_sum_f32(len(inputs), _ptr_array(inputs), _ptr(out), out.size)
    return out


def ref_style_sum_f32(inputs, out=None):
    inputs = [np.ascontiguousarray(a, dtype=np.float32) for a in inputs]
    if out is None:
        out = np.empty_like(inputs[0])
    lib().orc_ref_style_sum_f32(len(inputs), _ptr_array(inputs), _ptr(out), out.size)
    return out


def omp_threads():
    return int(lib().orc_omp_threads())


def f32_to_bf16_bits(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, dtype=np.uint16)
    lib().orc_cvt_f32_to_bf16(x.size, _ptr(x), _ptr(out))
    return out


def f32_to_f16_bits(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape, dtype=np.uint16)
    lib().orc_cvt_f32_to_f16(x.size, _ptr(x), _ptr(out))
    return out


def f16_bits_to_f32(b):
    b = np.ascontiguousarray(b, dtype=np.uint16)
    out = np.empty(b.shape, dtype=np.float32)
    lib().orc_cvt_f16_to_f32(b.size, _ptr(b), _ptr(out))
    return out


def bf16_bits_to_f32(b):
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return (b.astype(np.uint32) << 16).view(np.float32)
