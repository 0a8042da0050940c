"""INTEGRATION.md option B, built and run: the reference's OWN extension.cpp (checks + PYBIND11_MODULE,
/root/reference/lagomorph/extension/extension.cpp, compiled from where it lies) with
tests/native/lagomorph_ext_forwarders.cpp in place of its four cuda/*.cu files, linked against
liblagomorph_hip.so (oracle/build_ref.py: build_option_b -> oracle/_ref/lagomorph_ext_optionb.so; the
prebuilt module travels to the GPU box).  CPU: the module builds (or, without the reference tree, the
forwarders at least compile against the installed torch headers), exposes the 13 names of
extension.cpp:175-189, raises the reference's own CHECK_CUDA errors and serves the reference's CPU path.
GPU: every function of the module gives bit for bit what the ctypes shim gives, and matches the oracle."""
import os
import subprocess
import sys
import sysconfig

import numpy as np
import pytest
import torch

from oracle import build_ref
from oracle import lago_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ("set_debug_mode", "affine_interp_forward", "affine_interp_backward", "regrid_forward", "regrid_backward",
         "fluid_operator", "interp_forward", "interp_backward", "interp_hessian_diagonal_image",
         "jacobian_times_vectorfield_forward", "jacobian_times_vectorfield_backward",
         "jacobian_times_vectorfield_adjoint_forward", "jacobian_times_vectorfield_adjoint_backward")


@pytest.fixture(scope="module")
def optb():
    try:
        build_ref.build_option_b()
    except Exception as e:  # pragma: no cover
        print(f"[tests] option-B module not built: {e}")
    m = build_ref.load_option_b()
    if m is None:
        pytest.skip("oracle/_ref/lagomorph_ext_optionb.so not built (needs /root/reference)")
    return m


def test_forwarders_compile_against_torch_headers():
    """All 12 host functions of extension.cpp:29-102 as forwarders: a syntax-only compile with the host compiler
    (skipped when the full module was just built from the same file -- that is the stronger check)."""
    if os.path.exists(build_ref.EXT_SRC) and os.path.exists(build_ref.built_path_b()) and \
            os.path.getmtime(build_ref.built_path_b()) >= os.path.getmtime(os.path.join(ROOT, "tests", "native", "lagomorph_ext_forwarders.cpp")):
        return
    from torch.utils.cpp_extension import include_paths

    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-w", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"]
    for p in include_paths() + ["/opt/rocm/include", sysconfig.get_paths()["include"], os.path.join(ROOT, "include")]:
        cmd += ["-I", p]
    cmd.append(os.path.join(ROOT, "tests", "native", "lagomorph_ext_forwarders.cpp"))
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_module_surface_and_cpu_behaviour(optb):
    for n in NAMES:
        assert callable(getattr(optb, n)), n
    I = torch.randn(2, 1, 8, 8)
    A = torch.eye(2)[None].repeat(2, 1, 1).contiguous()
    assert torch.equal(optb.affine_interp_forward(I, A, torch.zeros(2, 2)), I)  # the reference's cpu/affine.cpp
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):  # extension.cpp:8 CHECK_CUDA
        optb.interp_forward(I, torch.zeros(2, 2, 8, 8), 1.0)


@pytest.mark.gpu
def test_reference_argument_checks_survive(optb):
    with pytest.raises(RuntimeError, match="Must provide same number cosine LUTs"):  # extension.cpp:167-168
        optb.fluid_operator(torch.zeros(1, 2, 4, 3, 2).cuda(), True, [], [], .1, 0., .01)
    with pytest.raises(RuntimeError, match="must be contiguous"):  # extension.cpp:9 CHECK_CONTIGUOUS
        optb.interp_forward(torch.zeros(1, 1, 4, 6).cuda().transpose(2, 3), torch.zeros(1, 2, 6, 4).cuda(), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("sp", [(6, 7, 9), (10, 12)])
def test_option_b_module_equals_ctypes_shim(optb, dtype, sp):
    import lagomorph_amd as lm

    ext = lm.lagomorph_ext
    d = len(sp)
    g = torch.Generator(device="cuda").manual_seed(3)
    r = lambda *s: torch.randn(s, device="cuda", dtype=dtype, generator=g)
    I, u, go = r(2, 2, *sp), 1.5 * r(2, d, *sp), r(2, 2, *sp)
    v, w, gv = r(2, d, *sp), r(2, d, *sp), r(2, d, *sp)
    A = (torch.eye(d, device="cuda", dtype=dtype)[None] + 0.1 * r(2, d, d)).contiguous()
    T = r(2, d)
    optb.set_debug_mode(True)
    try:
        assert torch.equal(optb.interp_forward(I, u, 0.7), ext.interp_forward(I, u, 0.7))
        np_I, np_u = I.cpu().numpy(), u.cpu().numpy()
        assert np.array_equal(optb.interp_forward(I, u, 0.7).cpu().numpy(), orc.interp_forward(np_I, np_u, 0.7))
        a, b = optb.interp_backward(go, I, u, 0.7, True, True), ext.interp_backward(go, I, u, 0.7, True, True)
        assert torch.equal(a[1], b[1]) and torch.allclose(a[0], b[0], rtol=0, atol=1e-5 * float(b[0].abs().max()))
        for disp, tr in ((True, False), (False, True)):
            assert torch.equal(optb.jacobian_times_vectorfield_forward(v, w, disp, tr),
                               ext.jacobian_times_vectorfield_forward(v, w, disp, tr))
            for x, y in zip(optb.jacobian_times_vectorfield_backward(gv, v, w, disp, tr, True, True),
                            ext.jacobian_times_vectorfield_backward(gv, v, w, disp, tr, True, True)):
                assert torch.equal(x, y)
        assert torch.equal(optb.jacobian_times_vectorfield_adjoint_forward(v, w),
                           ext.jacobian_times_vectorfield_adjoint_forward(v, w))
        for x, y in zip(optb.jacobian_times_vectorfield_adjoint_backward(gv, v, w, True, True),
                        ext.jacobian_times_vectorfield_adjoint_backward(gv, v, w, True, True)):
            assert torch.equal(x, y)
        assert torch.equal(optb.affine_interp_forward(I, A, T), ext.affine_interp_forward(I, A, T))
        for x, y in zip(optb.affine_interp_backward(go, I, A, T, True, True, True),
                        ext.affine_interp_backward(go, I, A, T, True, True, True)):
            assert torch.allclose(x, y, rtol=0, atol=1e-5 * max(float(y.abs().max()), 1e-30))
        shape = [s + 3 for s in sp]
        origin = [(s - 1) * 0.5 for s in sp]
        spacing = [(s - 1) / (S - 1) for s, S in zip(sp, shape)]
        out = optb.regrid_forward(I, shape, origin, spacing)
        assert torch.equal(out, ext.regrid_forward(I, shape, origin, spacing))
        gb = torch.randn(out.shape, device="cuda", dtype=dtype, generator=g)
        a, b = optb.regrid_backward(gb, list(sp), shape, origin, spacing), ext.regrid_backward(gb, list(sp), shape, origin, spacing)
        assert torch.allclose(a, b, rtol=0, atol=1e-5 * float(b.abs().max()))
        met = lm.FluidMetric([0.1, 0.05, 0.01])
        met.initialize_luts((2, d) + sp, dtype, "cuda")
        csp = list(sp)
        csp[-1] = csp[-1] // 2 + 1
        F1 = r(2, d, *csp, 2)
        F2 = F1.clone()
        optb.fluid_operator(F1, True, met.luts["cos"], met.luts["sin"], 0.1, 0.05, 0.01)
        ext.fluid_operator(F2, True, met.luts["cos"], met.luts["sin"], 0.1, 0.05, 0.01)
        assert torch.equal(F1, F2)
        if d == 2:
            a, b = optb.interp_hessian_diagonal_image(I, u, 1.0), ext.interp_hessian_diagonal_image(I, u, 1.0)
            assert torch.allclose(a, b, rtol=0, atol=1e-5 * float(b.abs().max()))  # atomic accumulation order
    finally:
        optb.set_debug_mode(False)
        lm.set_debug_mode(False)
