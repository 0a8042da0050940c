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
        os.path.getmtime(SRC), os.path.getmtime(os.path.join(HERE, "ref_cpu_binding.cpp")), os.path.getmtime(os.path.join(INC, "interp.h"))
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


EXT_SRC = os.path.join(REF, "lagomorph", "extension", "extension.cpp")
FWD_SRC = os.path.join(HERE, "..", "tests", "native", "lagomorph_ext_forwarders.cpp")
NAME_B = "lagomorph_ext_optionb"


def built_path_b():
    return os.path.join(OUT, NAME_B + ".so")


def build_option_b(verbose=False):
    """INTEGRATION.md option B as a real build: the reference's OWN extension.cpp (argument checks +
    PYBIND11_MODULE, compiled from where it lies) + its cpu/affine.cpp + tests/native/
    lagomorph_ext_forwarders.cpp in place of the four cuda/*.cu files, linked against
    liblagomorph_hip.so.  The result is the reference's `lagomorph_ext` pybind module running on the HIP
    kernels; tests/test_option_b.py imports it on the GPU box.  Test infrastructure (it contains compiled
    reference code), hence under oracle/_ref."""
    if not os.path.exists(EXT_SRC):
        print(f"[oracle/_ref] reference tree not present ({EXT_SRC}); keeping prebuilt files if any")
        return os.path.exists(built_path_b())
    lib = os.path.abspath(os.path.join(HERE, "..", "lagomorph_amd", "_lib", "liblagomorph_hip.so"))
    if not os.path.exists(lib):
        print("[oracle/_ref] liblagomorph_hip.so not built yet: option-B module skipped")
        return False
    deps = [EXT_SRC, SRC, FWD_SRC, os.path.join(HERE, "ref_cpu_binding.cpp"),
            os.path.join(HERE, "..", "include", "lagomorph_hip.h")]
    if os.path.exists(built_path_b()) and os.path.getmtime(built_path_b()) >= max(os.path.getmtime(d) for d in deps):
        return True
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load

    load(
        name=NAME_B,
        sources=[EXT_SRC, FWD_SRC, os.path.join(HERE, "ref_cpu_binding.cpp")],
        extra_include_paths=[INC, os.path.join(HERE, "..", "include"), "/opt/rocm/include"],
        extra_cflags=["-O2", "-w", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DLAGOMORPH_REF_NO_MODULE",
                      f'-DLAGOMORPH_REF_CPU_AFFINE=\\"{SRC}\\"'],
        # the library is linked by absolute path (/root/repo/... also exists on the GPU box, as a symlink)
        extra_ldflags=[lib, "-L/opt/rocm/lib", "-lamdhip64", "-lc10_hip", "-ltorch_hip"],
        build_directory=OUT,
        with_cuda=False,
        verbose=verbose,
    )
    return os.path.exists(built_path_b())


def load_option_b():
    import importlib.util

    import torch  # noqa: F401

    p = built_path_b()
    if not os.path.exists(p):
        return None
    import ctypes

    ctypes.CDLL(os.path.abspath(os.path.join(HERE, "..", "lagomorph_amd", "_lib", "liblagomorph_hip.so")),
                mode=ctypes.RTLD_GLOBAL)
    spec = importlib.util.spec_from_file_location(NAME_B, p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


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
    okb = build_option_b(verbose="-v" in sys.argv)
    print("[oracle/_ref] option-B module", "ok" if okb else "not built")
    sys.exit(0 if ok else 1)
