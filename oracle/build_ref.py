"""TEST INFRASTRUCTURE ONLY -- build oracle/_ref from the reference's own sources.

The only part of the reference that is plain C++ (everything else is CUDA,
which this image cannot compile: no nvcc, no cuda headers) is
``lagomorph/extension/cpu/affine.cpp``.  It is compiled here from where it lies
under /root/reference into ``oracle/_ref/lagomorph_ref_cpu.so`` (git-ignored,
travels to the GPU box with the snapshot).  No reference source is copied.

Usage: python oracle/build_ref.py        (no-op with a message when the
reference tree is absent, e.g. on the GPU box)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("LAGOMORPH_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "lagomorph", "extension", "cpu", "affine.cpp")
INC = os.path.join(REF, "lagomorph", "extension", "include")
OUT = os.path.join(HERE, "_ref")
NAME = "lagomorph_ref_cpu"


def built_path():
    return os.path.join(OUT, NAME + ".so")


def build(verbose=False):
    if not os.path.exists(SRC):
        print(f"[oracle/_ref] reference tree not present ({SRC}); keeping prebuilt files if any")
        return os.path.exists(built_path())
    if os.path.exists(built_path()) and os.path.getmtime(built_path()) >= max(
        os.path.getmtime(SRC), os.path.getmtime(os.path.join(HERE, "ref_cpu_binding.cpp"))
    ):
        return True
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load

    load(
        name=NAME,
        sources=[os.path.join(HERE, "ref_cpu_binding.cpp")],
        extra_include_paths=[INC],
        extra_cflags=["-O2", "-w", "-ffp-contract=off", f'-DLAGOMORPH_REF_CPU_AFFINE=\\"{SRC}\\"'],
        build_directory=OUT,
        with_cuda=False,
        verbose=verbose,
    )
    return os.path.exists(built_path())


def load_ref():
    """Import the prebuilt module (does not need /root/reference)."""
    import importlib.util

    import torch  # noqa: F401  (the extension links libtorch)

    p = built_path()
    if not os.path.exists(p):
        return None
    spec = importlib.util.spec_from_file_location(NAME, p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    ok = build(verbose="-v" in sys.argv)
    print("[oracle/_ref]", "ok" if ok else "not built")
    sys.exit(0 if ok else 1)
