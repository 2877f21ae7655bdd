"""A C++ program written only against include/rccl_amd.h (the way code written against <rccl/rccl.h>
is) replays the reference's known-answer test through ncclCommInitAll / ncclGroupStart / ncclAllReduce."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_cxx_known_answer():
    exe = os.path.join(HERE, "cxx", "known_answer")
    assert os.path.exists(exe), "build it with `make -C tests/cxx` (done by __graft_entry__.build())"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "known-answer test passed" in r.stdout


@pytest.mark.gpu
def test_cxx_multi_comm():
    """Four communicators driven by one thread (grouped launches), the reference's ncclCommSplit tests
    (test/StandaloneTests.cpp:18-153) and non-blocking creation (test/NonBlockingTests.cpp), all through the C ABI."""
    exe = os.path.join(HERE, "cxx", "multi_comm")
    assert os.path.exists(exe), "build it with `make -C tests/cxx` (done by __graft_entry__.build())"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "multi-communicator test passed" in r.stdout


def test_header_compiles_as_c():
    root = os.path.dirname(HERE)
    subprocess.check_call(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(HERE, "cxx", "header_is_c.c")])
