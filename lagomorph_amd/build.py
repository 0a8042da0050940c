"""Build liblagomorph_hip.so (the C-ABI HIP library) for gfx950 with hipcc.

In-tree build: objects under lagomorph_amd/_build/, the shared library at
lagomorph_amd/_lib/liblagomorph_hip.so (git-ignored; travels to the GPU box with
the gpurun snapshot).  hipcc cross-compiles without a GPU.

    python -m lagomorph_amd.build [-f] [-v] [--profiling]

--profiling builds lagomorph_amd/_lib/liblagomorph_hip_prof.so with -DLAGO_PROFILING: the same library plus the
`lago_debug_*` knobs that skip stages of kernels (results are WRONG with them set) for the ablation scripts under
tools/ (select it with LAGO_HIP_LIBRARY=<path>).  The product library has no result-changing knob.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "liblagomorph_hip.so")
LIB_PROF = os.path.join(LIBDIR, "liblagomorph_hip_prof.so")
SOURCES = ["api.hip", "interp.hip", "splat.hip", "diff.hip", "metric.hip", "affine.hip", "fused.hip", "fft.hip", "fftx.hip", "fft3.hip", "fft3x.hip", "fft3b.hip", "fftg.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# extra compiler flags for A/B builds of tools/ (e.g. LAGO_HIPCC_EXTRA="-DLAGO_NT_X_LD=1"; use -f: objects are not keyed on flags)
EXTRA = os.environ.get("LAGO_HIPCC_EXTRA", "").split()
FLAGS = [
    "-O3",
    "--offload-arch=gfx950",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",       # bit-parity with the strict-IEEE oracle where no atomic is involved
    "-munsafe-fp-atomics",     # hardware global_atomic_add_f32/f64 and ds_add_f32, no CAS loops
    "-Wall",
    "-Wno-unused-function",
]


def _deps():
    return [os.path.join(CSRC, h) for h in ("common.hpp", "fft_lds.hpp", "fft3_sizes.hpp", "stencil_tile.hpp", "gather_window.hpp", "fluid_bin.hpp")] + [
        os.path.join(HERE, "..", "include", "lagomorph_hip.h")]


def _stale(target, srcs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def _compile(src, force, verbose, profiling=False):
    obj = os.path.join(OBJ, "prof" if profiling else "", os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if not force and not _stale(obj, [path] + _deps()):
        return obj
    cmd = [HIPCC] + FLAGS + EXTRA + (["-DLAGO_PROFILING"] if profiling else []) + ["-c", path, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    return obj


def build(force=False, verbose=False, profiling=False):
    os.makedirs(os.path.join(OBJ, "prof") if profiling else OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    lib = LIB_PROF if profiling else LIB
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, verbose, profiling), srcs))
    if force or _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lhipfft"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose="-v" in sys.argv, profiling="--profiling" in sys.argv))
