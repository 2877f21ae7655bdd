#!/usr/bin/env python3
"""Per-kernel resource figures of the gfx950 code objects inside a library or object file: private segment (stack /
spill bytes per lane), VGPRs, spills, LDS.  Usage: tools/kernel_meta.py [file] [substring-of-kernel-name]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
KEYS = ("private_segment_fixed_size", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
        "group_segment_fixed_size", "uses_dynamic_stack")


def code_objects(path, tmp):
    """Unbundles every gfx950 code object of `path` into tmp and returns their paths."""
    local = shutil.copy(path, os.path.join(tmp, "in.bin"))
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=tmp)
    return sorted(os.path.join(tmp, p) for p in os.listdir(tmp) if p.endswith("gfx950"))


def kernels(path):
    """[{symbol, private_segment_fixed_size, ...}] for every kernel in the file's gfx950 code objects."""
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in code_objects(path, tmp):
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", obj], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in txt.split("\n"):
                m = re.match(r"\s+(- )?\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                first, key, val = m.group(1), m.group(2), m.group(3).strip()
                if first and key == "agpr_count":
                    cur = {}
                    rows.append(cur)
                if cur is None:
                    continue
                if key == "symbol":
                    cur["symbol"] = val[:-3] if val.endswith(".kd") else val
                elif key in KEYS:
                    cur[key] = val == "true" if key == "uses_dynamic_stack" else int(val)
    return [r for r in rows if "symbol" in r]


def demangle(names):
    tool = shutil.which("c++filt") or os.path.join(LLVM, "llvm-cxxfilt")
    try:
        out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True).stdout
    except OSError:
        return names
    return out.split("\n")[:len(names)]


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "rccl_amd", "librccl_amd.so")
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = kernels(path)
    for r, name in zip(rows, demangle([r["symbol"] for r in rows])):
        if pat in name:
            print(f"{r['private_segment_fixed_size']:5d} B stack  {r['vgpr_count']:3d} vgpr  {r['vgpr_spill_count']:3d} vspill  "
                  f"{r['sgpr_spill_count']:3d} sspill  {r['group_segment_fixed_size']:5d} lds  {name.split('(')[0]}")
