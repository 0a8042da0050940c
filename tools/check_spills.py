#!/usr/bin/env python3
"""Kernels of the built library whose register allocation spilled (code-object metadata, no GPU needed).

A spill in a hot kernel is not only slow code: scratch is HBM-backed, so every lane's spill stores become write
traffic (the first LDS-window compose wrote 134 MB per launch that way, the geometry-once splat 335 MB).

    python tools/check_spills.py [path/to/liblagomorph_hip.so]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(lib):
    """{kernel name: (vgprs, sgprs, spilled vgprs, scratch bytes)} for every gfx950 kernel in `lib`."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, capture_output=True, cwd=tmp)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True,
                                   capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                get = lambda key: re.search(rf"\.{key}:\s+(\S+)", blk)
                name = get("name")
                if not name:
                    continue
                out[name.group(1)] = tuple(int(get(k).group(1)) for k in
                                           ("vgpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size"))
    return out


def demangle(names):
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt", path=LLVM)
    if not tool or not names:
        return list(names)
    r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else list(names)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "lagomorph_amd", "_lib", "liblagomorph_hip.so")
    res = kernel_resources(lib)
    bad = {k: v for k, v in res.items() if v[2] or v[3]}
    print(f"{len(res)} kernels, {len(bad)} with spills or scratch")
    for name, pretty in zip(bad, demangle(list(bad))):
        v = bad[name]
        print(f"  {v[2]:4d} spilled VGPRs, {v[3]:5d} B scratch, {v[0]:3d} VGPRs: {pretty[:150]}")
