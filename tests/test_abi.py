"""The C-ABI library loads on a CPU-only machine and exports every symbol include/rccl_amd.h declares.
No compute is launched here."""
import ctypes
import os
import re

import rccl_amd as ra

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "rccl_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:p?nccl|p?msccl|rcclAmd)[A-Za-z0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    names = _declared()
    assert len(names) >= 30
    L = ctypes.CDLL(ra.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(ra.EXPORTS) == names          # the Python mirror binds exactly the declared ABI


def test_every_public_entry_point_of_the_reference_is_exported():
    """The 38 functions the reference's header declares (list in tests/golden/nccl_public_symbols.txt, taken from
    src/nccl.h.in:105-818) and their p-prefixed twins are dynamic symbols of the library: a program linked against
    the reference's librccl loads against this one."""
    import subprocess
    names = [l.strip() for l in open(os.path.join(ROOT, "tests", "golden", "nccl_public_symbols.txt")) if l.strip() and not l.startswith("#")]
    assert len(names) == 38
    out = subprocess.run(["nm", "-D", "--defined-only", ra.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    missing = [n for n in names + ["p" + n for n in names] if n not in exported]
    assert not missing, missing


def test_version_macros_and_forwarding_header(tmp_path):
    """NCCL_MAJOR/MINOR/PATCH, NCCL_VERSION_CODE, NCCL_VERSION(), NCCL_COMM_NULL, NCCL_SPLIT_NOCOLOR
    (src/nccl.h.in:15-21,37,65) through <rccl/rccl.h>, compiled as C."""
    import subprocess
    src = tmp_path / "v.c"
    src.write_text("""#include <rccl/rccl.h>
#if NCCL_VERSION_CODE != NCCL_VERSION(NCCL_MAJOR, NCCL_MINOR, NCCL_PATCH) || NCCL_VERSION_CODE != 22105
#error version macros disagree
#endif
#if NCCL_VERSION(2, 8, 3) != 2803 || NCCL_VERSION(2, 21, 5) != 22105
#error NCCL_VERSION
#endif
static ncclComm_t c = NCCL_COMM_NULL;
static int nocolor = NCCL_SPLIT_NOCOLOR;
static ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
int main(void) { return (c == 0 && nocolor == -1 && cfg.blocking == NCCL_CONFIG_UNDEF_INT) ? 0 : 1; }
""")
    exe = tmp_path / "v"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    assert subprocess.run([str(exe)]).returncode == 0


def test_version_and_error_strings():
    assert ra.get_version() == 22105              # NCCL 2.21.5 API level, as the reference snapshot
    L = ra.lib()
    assert L.ncclGetErrorString(0) == b"no error"
    for code in range(1, 8):
        assert len(L.ncclGetErrorString(code)) > 5
    assert L.ncclGetErrorString(99) == b"unknown result code"


def test_null_communicator_is_invalid_argument():
    # src/misc/argcheck.cc:36-43: a NULL communicator is ncclInvalidArgument, never a crash
    L = ra.lib()
    assert L.ncclAllReduce(None, None, 16, ra.ncclFloat32, ra.ncclSum, None, None) == ra.ncclInvalidArgument
    assert L.ncclReduceScatter(None, None, 16, ra.ncclFloat32, ra.ncclSum, None, None) == ra.ncclInvalidArgument
    assert L.ncclAllGather(None, None, 16, ra.ncclFloat32, None, None) == ra.ncclInvalidArgument
    n = ctypes.c_int()
    assert L.ncclCommCount(None, ctypes.byref(n)) == ra.ncclInvalidArgument
    assert L.ncclCommDestroy(None) == ra.ncclSuccess          # destroying NULL is a no-op
    assert b"NULL" in L.ncclGetLastError(None)
    assert L.ncclGroupEnd() == ra.ncclInvalidUsage             # not inside a group
    assert L.ncclGroupStart() == ra.ncclSuccess and L.ncclGroupEnd() == ra.ncclSuccess


def test_out_of_scope_entry_points_say_so():
    L = ra.lib()
    assert L.ncclSend(None, 0, 0, 0, None, None) == ra.ncclInvalidUsage
    assert L.ncclRecv(None, 0, 0, 0, None, None) == ra.ncclInvalidUsage
    for name in ("ncclGather", "ncclScatter", "ncclAllToAll", "mscclUnloadAlgo"):
        assert getattr(L, name)(None, None, 0, 0, 0, None, None) == ra.ncclInvalidUsage, name
    assert b"outside" in L.ncclGetLastError(None)
    assert L.ncclBroadcast(None, None, 0, 0, 0, None, None) == ra.ncclInvalidArgument   # on the path: NULL comm


def test_unique_ids_differ():
    a, b = ra.get_unique_id(), ra.get_unique_id()
    assert len(a) == 128 and a != b


def test_product_never_imports_the_oracle():
    """The shipped package and library must not reach into oracle/ (checker only)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rccl_amd")):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cc", ".h", ".hip")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().lower(), f
