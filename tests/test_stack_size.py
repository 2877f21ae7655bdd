"""Every device kernel of the library runs without a private segment (no spill, no stack object), read from the code
objects' metadata.  The reference gates its own device code at 360 bytes of stack per function
(tools/scripts/check_stack_size.py, CMakeLists.txt ENABLE_STACK_SIZE_CHECK); here the bound is held for whole kernels, and
the collective kernels are held to zero: a kernel with no private segment asks the runtime for no scratch memory at
launch, whatever number of processes share the GPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_meta  # noqa: E402

LIB = os.path.join(ROOT, "rccl_amd", "librccl_amd.so")
FAMILIES = ("raCollKernel", "raOneShotKernel", "raTwoExchangeKernel", "raGatherKernel", "raLL", "raLocalKernel", "raNoAlloc")
# kernels allowed a private segment, by demangled name: none
SMALL = {}


@pytest.fixture(scope="module")
def rows():
    if not os.path.exists(LIB):
        pytest.fail("librccl_amd.so is not built")
    rs = kernel_meta.kernels(LIB)
    names = kernel_meta.demangle([r["symbol"] for r in rs])
    return [(n.split("(")[0].replace("void ", ""), r) for n, r in zip(names, rs)]


def _in_family(name):
    # (names the demangler does not know, the _Float16 instantiations, stay mangled: "_Z12raCollKernelIDF16_...")
    return name.startswith(FAMILIES) or (name.startswith("_Z") and any(f in name for f in FAMILIES))


def test_every_family_is_present(rows):
    for fam in FAMILIES:
        assert any(n.startswith(fam) for n, _ in rows), fam
    assert len(rows) > 500          # 12 types x 5 operators x the schedules
    assert all(_in_family(n) for n, _ in rows), [n for n, _ in rows if not _in_family(n)][:5]


def test_no_kernel_has_a_stack_beyond_the_gate(rows):
    over = [(n, r["private_segment_fixed_size"]) for n, r in rows if r["private_segment_fixed_size"] > 360]
    assert not over, over
    assert not [n for n, r in rows if r.get("uses_dynamic_stack")]


def test_collective_kernels_have_no_private_segment(rows):
    bad = []
    for n, r in rows:
        if not _in_family(n):
            continue
        want = SMALL.get(n, 0)
        if r["private_segment_fixed_size"] > want or (want == 0 and r["vgpr_spill_count"]):
            bad.append((n, r["private_segment_fixed_size"], r["vgpr_spill_count"]))
    assert not bad, bad[:10]


def test_one_workgroup_of_1024_lanes_fits(rows):
    # 1024-lane workgroups leave 128 VGPRs a lane (512 per SIMD lane / 4 waves); accumulation registers count
    for n, r in rows:
        if _in_family(n):
            assert r["vgpr_count"] + r.get("agpr_count", 0) <= 128, (n, r)


def test_no_array_ends_up_in_lds(rows):
    # The kernels' own LDS is the per-channel protocol state (below 2 KiB).  An argument array indexed at run time would
    # be moved to LDS by the compiler (64 B per lane: 64 KiB per workgroup) instead of showing up as a stack frame.
    for n, r in rows:
        assert r["group_segment_fixed_size"] <= 4096, (n, r["group_segment_fixed_size"])
