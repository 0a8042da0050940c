"""CPU: the C-ABI library loads and exports every symbol include/lagomorph_hip.h declares (no
compute calls without a GPU), the Python shim fails loudly on CPU tensors, and the host-side
logic that mirrors the reference (LUTs, regrid argument rules, expmap flags) behaves like it."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lagomorph_hip.h")).read()
    names = set()
    for m in re.finditer(r"\b(lago_\w+)##SUF\s*\(", text):
        names.update({m.group(1) + "_f32", m.group(1) + "_f64"})
    for m in re.finditer(r"^\s*(?:int|void|long long|const char \*)\s*\*?(lago_\w+)\s*\(", text, re.M):
        names.add(m.group(1))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import lagomorph_amd

    lib = ctypes.CDLL(lagomorph_amd.lagomorph_ext.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 35, names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/lagomorph_hip.h but not exported: {missing}"
    assert lib.lago_abi_version() == lagomorph_amd.lagomorph_ext.ABI_VERSION == 5
    lib.lago_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.lago_version()


def test_library_exports_nothing_undeclared():
    """The other direction: every `lago_*` symbol the product library exports is declared in the header -- in
    particular no `lago_debug_*` profiling knob (those exist only in the -DLAGO_PROFILING build for tools/)."""
    import subprocess

    import lagomorph_amd

    out = subprocess.run(["nm", "-D", "--defined-only", lagomorph_amd.lagomorph_ext.LIB_PATH], capture_output=True,
                         text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TW"
                       and ln.split()[-1].startswith("lago_")})
    assert len(exported) >= 35
    extra = [n for n in exported if n not in set(declared_symbols())]
    assert not extra, f"exported by the library but not declared in include/lagomorph_hip.h: {extra}"
    assert not [n for n in exported if n.startswith("lago_debug")]


def test_loader_refuses_a_stale_abi(tmp_path):
    """a library of another ABI version must be refused at import, not called with shifted arguments."""
    import subprocess
    import sys

    src = tmp_path / "stale.c"
    src.write_text("int lago_abi_version(void) { return 2; }\n")
    so = tmp_path / "libstale.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    r = subprocess.run([sys.executable, "-c", "import lagomorph_amd"], capture_output=True, text=True, cwd=ROOT,
                       env={**os.environ, "LAGO_HIP_LIBRARY": str(so)})
    assert r.returncode != 0 and "C-ABI version 2" in r.stderr and "lagomorph_amd.build" in r.stderr


def test_shim_covers_the_reference_surface():
    """The 13 names of extension.cpp:175-189."""
    import lagomorph_amd

    ext = lagomorph_amd.lagomorph_ext
    for name in ("set_debug_mode", "affine_interp_forward", "affine_interp_backward", "regrid_forward",
                 "regrid_backward", "fluid_operator", "interp_forward", "interp_backward",
                 "interp_hessian_diagonal_image", "jacobian_times_vectorfield_forward",
                 "jacobian_times_vectorfield_backward", "jacobian_times_vectorfield_adjoint_forward",
                 "jacobian_times_vectorfield_adjoint_backward"):
        assert callable(getattr(ext, name)), name


def test_no_cpu_fallback():
    """Every compute entry point rejects CPU tensors (the product has no CPU path; the reference's
    cpu/affine.cpp fallback is deliberately not reproduced)."""
    import lagomorph_amd as lm

    I = torch.zeros((1, 1, 4, 4, 4))
    u = torch.zeros((1, 3, 4, 4, 4))
    A, T = torch.eye(3)[None], torch.zeros((1, 3))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        lm.interp(I, u)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        lm.jacobian_times_vectorfield(u, u)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        lm.affine_interp(I, A, T)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        lm.FluidMetric().sharp(u)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        lm.regrid(I, shape=(5, 5, 5))
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        lm.compose(u, u)


def test_fluid_luts_match_reference_formula():
    """metric.py:66-75: float64 numpy -> float32 (torch.Tensor) -> dtype."""
    from lagomorph_amd.metric import FluidMetric, fluid_luts

    luts = fluid_luts((6, 5, 8), torch.float64, "cpu")
    for d, (N, Nf) in enumerate(((6, 6), (5, 5), (8, 5))):
        k = np.arange(Nf)
        want_c = torch.Tensor(2.0 * (1.0 - np.cos(2 * np.pi * k / N))).type(torch.float64)
        want_s = torch.Tensor(np.sin(2.0 * np.pi * k / N)).type(torch.float64)
        assert torch.equal(luts["cos"][d], want_c) and torch.equal(luts["sin"][d], want_s)
    # the reference reuses stale LUTs when the shape changes (metric.py:63); here they are per shape
    m = FluidMetric([0.1, 0.0, 0.01])
    m.initialize_luts((1, 2, 8, 6), torch.float32, "cpu")
    a = m.luts
    m.initialize_luts((1, 2, 4, 6), torch.float32, "cpu")
    assert m.luts["cos"][0].numel() == 4 and a["cos"][0].numel() == 8
    assert m.complexshape == (1, 2, 4, 4)


def test_regrid_argument_rules():
    """affine.py:230-258: only `shape` alone is implemented; everything else raises as in the reference."""
    import lagomorph_amd as lm

    I = torch.zeros((1, 1, 4, 4))
    with pytest.raises(ValueError):
        lm.regrid(I)
    with pytest.raises(NotImplementedError):
        lm.regrid(I, spacing=1.0)
    with pytest.raises(NotImplementedError):
        lm.regrid(I, origin=1.0)
    with pytest.raises(ValueError):
        lm.regrid(I, origin=1.0, spacing=1.0)
    with pytest.raises(NotImplementedError):
        lm.regrid(I, shape=(4, 4), origin=(1.5, 1.5))


def test_expmap_rejects_the_broken_checkpoint_path():
    import lagomorph_amd as lm

    with pytest.raises(NotImplementedError):
        lm.expmap(lm.FluidMetric(), torch.zeros((1, 2, 4, 4)), checkpoints=2)
    with pytest.raises(NotImplementedError):
        lm.Ad(None, None)


def test_identity_grid():
    import lagomorph_amd as lm

    ix = lm.identity((2, 3, 2, 3, 4))
    assert ix.shape == (2, 3, 2, 3, 4) and ix.dtype == np.float32
    assert (ix[1, 0, 1] == 1).all() and (ix[0, 1, :, 2] == 2).all() and (ix[0, 2, :, :, 3] == 3).all()


def test_oracle_builds_agree():
    """Same source, contraction on/off: the two oracle builds differ by rounding only."""
    from oracle import lago_oracle as orc

    rng = np.random.default_rng(0)
    I = rng.standard_normal((2, 2, 6, 7, 8)).astype(np.float32)
    u = (1.5 * rng.standard_normal((2, 3, 6, 7, 8))).astype(np.float32)
    try:
        orc.set_strict(False)
        a = orc.interp_forward(I, u, 0.7)
        orc.set_strict(True)
        b = orc.interp_forward(I, u, 0.7)
    finally:
        orc.set_strict(False)
    assert np.abs(a - b).max() <= 2e-6 * np.abs(a).max()
    assert not np.array_equal(a, b)  # they really are two different evaluations


def test_header_is_plain_c_and_links_against_the_library(tmp_path):
    """include/lagomorph_hip.h is the drop-in boundary: it must compile as strict C99 (no C++, no torch
    types) and a C program must link against the shared library and read its housekeeping symbols
    (no GPU needed for those)."""
    import shutil
    import subprocess

    import lagomorph_amd.lagomorph_ext as ext

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    src = tmp_path / "abi.c"
    src.write_text(
        '#include <stdio.h>\n#include <string.h>\n#include "lagomorph_hip.h"\n'
        "int main(void) {\n"
        "    if (lago_abi_version() != LAGO_ABI_VERSION) return 1;\n"
        "    if (!lago_version() || !strlen(lago_version())) return 2;\n"
        "    lago_set_debug(1); if (lago_get_debug() != 1) return 3; lago_set_debug(0);\n"
        "    { lago_tuning t, d; t.struct_size = sizeof t; d.struct_size = sizeof d; lago_get_tuning(&t); lago_default_tuning(&d);\n"
        "      if (memcmp(&t, &d, sizeof t) || t.splat_shear[7] != 1024 || t.fluid_mode != 3) return 5;\n"
        "      t.gather_window = 0; if (lago_set_tuning(&t) != LAGO_OK) return 6; lago_get_tuning(&d); if (d.gather_window != 0) return 7;\n"
        "      t.struct_size = 6; if (lago_set_tuning(&t) != LAGO_ERR_INVALID) return 8;\n"
        "      t.struct_size = 8; t.splat_mode = 0; t.gather_window = 1; if (lago_set_tuning(&t) != LAGO_OK) return 9;   /* an older, shorter struct */\n"
        "      d.struct_size = sizeof d; lago_get_tuning(&d); if (d.splat_mode != 0 || d.gather_window != 0) return 10; }\n"
        "    /* argument validation happens before any HIP call */\n"
        "    if (lago_interp_forward_f32(0, 0, 0, 1.0, 5, 1, 1, 4, 4, 4, 0, 0) != LAGO_ERR_INVALID) return 4;\n"
        '    printf("%s\\n", lago_last_error());\n'
        "    return 0;\n}\n")
    exe = tmp_path / "abi"
    libdir = os.path.dirname(ext.LIB_PATH)
    inc = os.path.join(ROOT, "include")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", f"-I{inc}", str(src), "-o", str(exe),
                    f"-L{libdir}", "-llagomorph_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "two- and three-dimensional" in r.stdout


def test_builder_does_not_need_the_package():
    """A fresh checkout has no library, and `import lagomorph_amd` fails loudly without it -- so the
    builder must be loadable on its own (by path, as __graft_entry__.build() and conftest do) and
    `python -m lagomorph_amd.build` must get past the package import."""
    import subprocess
    import sys

    import __graft_entry__ as ge

    b = ge._load_builder()
    assert callable(b.build) and b.LIB.endswith("liblagomorph_hip.so")
    src = open(os.path.join(ROOT, "lagomorph_amd", "build.py")).read()
    assert "from ." not in src and "import lagomorph_amd" not in src
    # the package import is skipped when the interpreter was started as `-m lagomorph_amd.build`
    code = ("import sys; sys.orig_argv = ['python', '-m', 'lagomorph_amd.build']; import lagomorph_amd; "
            "print(lagomorph_amd._BUILDING, hasattr(lagomorph_amd, 'expmap'))")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0 and r.stdout.split() == ["True", "False"], (r.stdout, r.stderr)


def test_tuning_is_one_struct():
    """include/lagomorph_hip.h: the sixteen per-experiment setters of rounds 1-3 are ONE struct now.
    The Python shim reads, changes and writes it as a whole; unknown fields and wrong lengths are refused; the defaults
    are what the library starts with."""
    import lagomorph_amd.lagomorph_ext as ext

    d = ext.default_tuning()
    assert ext.get_tuning() == d
    assert d["splat_shear"] == [1, 8, 6, 0, 1, 1, 4, 1024] and d["splat_tile"] == [0, 8, 0, 1, 1, 4, 512]
    assert d["fluid_mode"] == 3 and d["splat_shear_mc"] == 2 and d["launch_order"] == 1
    try:
        ext.tune(gather_window=0, splat_shear=[1, 4, 8, 0, 2, 2, 8, 512])
        t = ext.get_tuning()
        assert t["gather_window"] == 0 and t["splat_shear"] == [1, 4, 8, 0, 2, 2, 8, 512]
        assert {k: v for k, v in t.items() if k not in ("gather_window", "splat_shear")} == \
               {k: v for k, v in d.items() if k not in ("gather_window", "splat_shear")}
        ext.set_fluid_mode(0)   # the convenience wrappers go through the same struct
        assert ext.get_tuning()["fluid_mode"] == 0
        with pytest.raises(KeyError):
            ext.tune(no_such_field=1)
        with pytest.raises(ValueError):
            ext.tune(splat_tile=[1, 2, 3])
    finally:
        ext.tune(**d)
    assert ext.get_tuning() == d


def test_operator_output_factor_on_the_host_form(oracle_ext):
    """`FluidMetric.sharp(m, out_scale=s)` (the GPU form folds s into the operator's last kernel,
    lago_fluid_metric_scaled): on the three-call host form it is `sharp(m) * s`, its backward s * sharp(grad), and the
    first Euler step of `expmap` from the identity uses it (-dt sharp(m0))."""
    import lagomorph_amd as lm
    from lagomorph_amd import lddmm

    g = torch.Generator().manual_seed(3)
    m = torch.randn((2, 3, 6, 8, 10), dtype=torch.float64, generator=g)
    met = lm.FluidMetric([0.1, 0.05, 0.01])
    v = met.sharp(m)
    assert torch.equal(met.sharp(m, out_scale=-0.25), v * -0.25)
    assert torch.equal(met.sharp(m, out_scale=1.0), v)
    mr = m.clone().requires_grad_(True)
    go = torch.randn(m.shape, dtype=torch.float64, generator=g)
    met.sharp(mr, out_scale=-0.25).backward(go)
    assert torch.equal(mr.grad, met.sharp(go) * -0.25)
    assert torch.equal(lddmm._first_step(met, m, 0.5), v * -0.5)
    assert torch.equal(lddmm._first_step(met, m, 0.5, v0=v), v * -0.5)
    assert torch.equal(lm.expmap(met, m, num_steps=1), v * -1.0)
