"""Static check of the built library's gfx950 code for the store-data hazard hipcc (ROCm 7.2) does not cover.

A 16-byte (or 12-byte) buffer store whose scalar-offset field holds a REGISTER is treated by the compiler as free of
the store-data hazard, so it may schedule a VALU write of the data registers directly behind the store; on gfx950 the
store then ships the overwritten dwords of some lanes (found as rare wrong dwords in 8-bit integer reductions).  The
kernels therefore keep the constant 0 in that field (kernels_impl.h: stSysAt), and this test holds them to it: it
disassembles every gfx950 code object of librccl_amd.so and fails on a wide buffer store with a register there, and on
any wide store of any kind whose data registers the very next instruction overwrites.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("RCCL_AMD_LIB") or os.path.join(ROOT, "rccl_amd", "librccl_amd.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not present")
def test_no_wide_store_with_unprotected_data_registers(tmp_path):
    assert os.path.exists(LIB), "build the library first (make -C rccl_amd/csrc)"
    lib = shutil.copy(LIB, tmp_path / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)   # writes the bundles beside the copy
    objs = sorted(p for p in os.listdir(tmp_path) if p.endswith("gfx950"))
    assert objs, "no gfx950 code object found in the library"
    wide = re.compile(r"^(buffer|global|flat|scratch)_store_dwordx[34]$")
    n_stores, bad = 0, []
    for obj in objs:
        text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(tmp_path / obj)], check=True, capture_output=True,
                              text=True).stdout
        ins = []
        for line in text.split("\n"):
            line = line.split("//")[0].strip()
            if line and not line.endswith(":") and not line.startswith(("Disassembly", obj)) and not line.startswith("<"):
                ins.append(line)
        for i, line in enumerate(ins):
            parts = line.replace(",", " ").split()
            if not wide.match(parts[0]):
                continue
            n_stores += 1
            data = set()
            for tok in parts[1:3]:
                r = _regs(tok)
                if len(r) >= 3:
                    data = r
            if parts[0].startswith("buffer_store"):
                # buffer_store_dwordx4 vdata, voffset|off, srsrc, soffset ...
                soffset = parts[4]
                if re.match(r"s\d+$", soffset):
                    bad.append(f"{obj}: register in the scalar-offset field: {line}")
            if i + 1 < len(ins):
                nxt = ins[i + 1].replace(",", " ").split()
                if nxt[0].startswith("v_") and not nxt[0].startswith(("v_cmp", "v_readfirstlane", "v_readlane")) and len(nxt) > 1:
                    if _regs(nxt[1]) & data:
                        bad.append(f"{obj}: data registers overwritten right behind the store: {line}  ->  {ins[i + 1]}")
    assert n_stores > 1000, f"only {n_stores} wide stores seen: the disassembly was not parsed"
    assert not bad, bad[:5]
