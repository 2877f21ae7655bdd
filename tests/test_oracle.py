"""Pins the CPU oracle (oracle/oracle.c) before anything is checked against it.

Sources of truth used here:
  * tests/golden/hip_f16_bf16_*.csv — produced by oracle/gen_golden_hip.cc from the ROCm HIP
    host headers, the 16-bit float arithmetic the reference calls (reduce_kernel.h:166,260-262;
    test/common/PtrUnion.cpp:181,185,271,275);
  * the reference's known-answer test (test/StandaloneTests.cpp:191-246);
  * the reference's input pattern and tolerances (test/common/PtrUnion.cpp:149-154,313-342).
"""
import csv
import os

import numpy as np
import pytest

import oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rows(name):
    with open(os.path.join(GOLD, name)) as f:
        r = csv.reader(f)
        next(r)
        return [[int(x, 16) for x in row] for row in r]


def test_golden_convert_matches_hip_headers():
    rows = _rows("hip_f16_bf16_convert.csv")
    assert len(rows) > 6000
    f32 = np.array([r[0] for r in rows], dtype=np.uint32).view(np.float32)
    bf = orc.f32_to_bf16_bits(f32)
    fh = orc.f32_to_f16_bits(f32)
    assert np.array_equal(bf, np.array([r[1] for r in rows], dtype=np.uint16))
    assert np.array_equal(fh, np.array([r[2] for r in rows], dtype=np.uint16))
    # numpy's own IEEE binary16 conversion agrees too
    with np.errstate(over="ignore"):
        ok = ~np.isnan(f32)  # NaN payload conventions differ; the HIP goldens above cover them
        assert np.array_equal(f32.astype(np.float16).view(np.uint16)[ok], fh[ok])


def test_golden_add_mul_matches_hip_headers():
    rows = _rows("hip_f16_bf16_add.csv")
    a = np.array([r[0] for r in rows], dtype=np.uint16)
    b = np.array([r[1] for r in rows], dtype=np.uint16)
    for dt, cs, cp in ((orc.F16, 2, 3), (orc.BF16, 4, 5)):
        for op, col in ((orc.SUM, cs), (orc.PROD, cp)):
            got = orc.host_fold(dt, op, [a.view(orc.NP_OF[dt]), b.view(orc.NP_OF[dt])]).view(np.uint16)
            want = np.array([r[col] for r in rows], dtype=np.uint16)
            assert np.array_equal(got, want), (dt, op)


def test_golden_fp8_matches_reference_header():
    """Both 8-bit formats, code by code, against vectors produced from the reference's rccl_float8.h."""
    L = orc.lib()
    for code, e4, e5 in _rows("ref_fp8_decode.csv"):
        for fn, want in ((L.orc_f8e4m3_to_f32, e4), (L.orc_f8e5m2_to_f32, e5)):
            got = np.array([fn(code)], np.float32).view(np.uint32)[0]
            assert got == want or (code == 0x80 and np.isnan(np.array([got], np.uint32).view(np.float32)[0])), (code, hex(got), hex(want))
    rows = _rows("ref_fp8_encode.csv")
    assert len(rows) > 7000
    f32 = np.array([r[0] for r in rows], dtype=np.uint32).view(np.float32)
    for i, x in enumerate(f32):
        assert L.orc_f32_to_f8e4m3(float(x)) == rows[i][1], (hex(rows[i][0]), "e4m3")
        assert L.orc_f32_to_f8e5m2(float(x)) == rows[i][2], (hex(rows[i][0]), "e5m2")


def test_sixteen_bit_known_values():
    # 1/3 + 1/7 in the 16-bit formats, computed with the HIP host headers: f16 0x379e, bf16 0x3ef4
    for dt, want in ((orc.F16, 0x379E), (orc.BF16, 0x3EF4)):
        cvt = orc.f32_to_f16_bits if dt == orc.F16 else orc.f32_to_bf16_bits
        a = cvt(np.array([1.0 / 3], np.float32)).view(orc.NP_OF[dt])
        b = cvt(np.array([1.0 / 7], np.float32)).view(orc.NP_OF[dt])
        assert orc.host_fold(dt, orc.SUM, [a, b]).view(np.uint16)[0] == want


def test_known_answer_standalone():
    # test/StandaloneTests.cpp:191-246: 2 ranks, ncclInt sum, N=1250, in[i]=i -> 2*i
    n = 1250
    x = np.arange(n, dtype=np.int32)
    want = 2 * x
    assert np.array_equal(orc.host_fold(orc.I32, orc.SUM, [x, x]), want)
    assert np.array_equal(orc.device_fold(orc.I32, orc.SUM, [x, x]), want)
    assert np.array_equal(orc.ring_allreduce(orc.I32, orc.SUM, [x, x], [0, 1], 512, 128), want)
    assert np.array_equal(orc.chain_allreduce(orc.I32, orc.SUM, [x, x], [0, 1]), want)


@pytest.mark.parametrize("dt", list(orc.NP_OF))
def test_pattern(dt):
    n, rank = 700, 3
    got = orc.fill_pattern(dt, n, rank)
    vi = (rank + np.arange(n)) % 256
    if dt in (orc.I8, orc.U8, orc.I32, orc.U32, orc.I64, orc.U64):
        assert np.array_equal(got, vi.astype(orc.NP_OF[dt]))
    else:
        vf = 1.0 / (vi.astype(np.float64) + 1.0)
        if dt == orc.F64:
            assert np.array_equal(got, vf)
        elif dt == orc.F32:
            assert np.array_equal(got, vf.astype(np.float32))
        elif dt == orc.F16:
            assert np.array_equal(got, vf.astype(np.float32).astype(np.float16))
        elif dt in (orc.F8E4M3, orc.F8E5M2):
            fn = orc.lib().orc_f32_to_f8e4m3 if dt == orc.F8E4M3 else orc.lib().orc_f32_to_f8e5m2
            assert np.array_equal(got, np.array([fn(float(np.float32(v))) for v in vf], np.uint8))
        else:
            assert np.array_equal(got, orc.f32_to_bf16_bits(vf.astype(np.float32)))


@pytest.mark.parametrize("dt", list(orc.NP_OF))
@pytest.mark.parametrize("op", [orc.SUM, orc.PROD, orc.MAX, orc.MIN, orc.AVG])
@pytest.mark.parametrize("n", [2, 4, 8])
def test_orders_agree_within_reference_tolerance(dt, op, n):
    """Ring / chain / rank-order association differ only by float rounding; ints are exact
    (wraparound arithmetic is associative), except integer Avg which is exact too."""
    count = 1031
    ins = [orc.fill_pattern(dt, count, r) for r in range(n)]
    host = orc.host_fold(dt, op, ins)
    dev = orc.device_fold(dt, op, ins)
    ring = orc.ring_allreduce(dt, op, ins, list(range(n)), 64, 16)
    chain = orc.chain_allreduce(dt, op, ins, list(range(n)))
    is_int = dt in (orc.I8, orc.U8, orc.I32, orc.U32, orc.I64, orc.U64)
    is_f8 = dt in (orc.F8E4M3, orc.F8E5M2)
    for other in (dev, ring, chain):
        if is_f8 and op in (orc.SUM, orc.PROD, orc.AVG):
            continue   # 2-3 mantissa bits: orders differ by whole ulps; the reference tests fp8 with min/max only
        if is_int or is_f8:
            assert np.array_equal(other, host)
        else:
            assert orc.first_mismatch(dt, other, host) == -1
    if op != orc.AVG:  # rank-order left fold is bit-identical to the host fold for every type
        assert np.array_equal(dev.view(np.uint8), host.view(np.uint8))


def test_numpy_cross_check_fp32_sum():
    rng = np.random.default_rng(1)
    ins = [rng.uniform(-1, 1, 4099).astype(np.float32) for _ in range(8)]
    want = ins[0].copy()
    for x in ins[1:]:
        want = want + x
    assert np.array_equal(orc.host_fold(orc.F32, orc.SUM, ins), want)
    assert np.array_equal(orc.omp_sum_f32(ins), want)
    assert np.array_equal(orc.ref_style_sum_f32(ins), want)
    # ring order: chunk c finishes at position c after starting at c+1
    ring = orc.ring_allreduce(orc.F32, orc.SUM, ins, list(range(8)), 512, 512)
    c = 2
    seg = slice(c * 512, (c + 1) * 512)
    acc = ins[(c + 1) % 8][seg].copy()
    for k in range(2, 9):
        acc = ins[(c + k) % 8][seg] + acc
    assert np.array_equal(ring[seg], acc)


def test_tolerance_rule():
    a = np.array([1.0, 2.0], np.float32)
    assert orc.first_mismatch(orc.F32, a, a + np.float32(5e-6)) == -1
    assert orc.first_mismatch(orc.F32, a, a + np.array([0, 2e-5], np.float32)) == 1
    i = np.array([1, 2, 3], np.int32)
    assert orc.first_mismatch(orc.I32, i, i) == -1
    assert orc.first_mismatch(orc.I32, i, i + np.array([0, 0, 1], np.int32)) == 2
