"""GPU: the HIP path (through lagomorph_ext -> C ABI) against the CPU oracle on identical
seeded inputs.  Bit-exact wherever no atomic is involved (the library is built with
-ffp-contract=off and follows the reference's expression order); tolerance-based for
scatter-add results, whose summation order is unspecified in the reference too.
"""
import json
import os
import re

import numpy as np
import pytest
import torch

import kat
from oracle import lago_oracle as orc

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.float64]
# fp32 tolerance from BASELINE.json north_star: <=1e-5 relative for interp/metric outputs.  Every tolerance-based
# comparison of this file uses exactly this bound (no multipliers): the largest error observed on MI355X is 0.58 of it
# (affine d_I, float32; profiles/r02_tolerances.json lists every comparison), 0.016 of the float64 bound.
RTOL = {torch.float32: 1e-5, torch.float64: 1e-12}


@pytest.fixture(scope="module")
def ext():
    import lagomorph_amd

    lagomorph_amd.set_debug_mode(True)
    yield lagomorph_amd.lagomorph_ext
    out = os.environ.get("LAGO_TOL_REPORT")
    if out:  # observed errors of this run, in units of the north_star tolerance
        json.dump(dict(sorted(OBSERVED.items())), open(out, "w"), indent=1)


def rnd(rng, shape, dtype, scale=1.0):
    a = (scale * rng.standard_normal(shape)).astype(np.float32 if dtype == torch.float32 else np.float64)
    return a


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def assert_bits(got, want, what):
    got, want = host(got), np.asarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    if not np.array_equal(got, want):
        d = np.abs(got.astype(np.float64) - want.astype(np.float64))
        raise AssertionError(f"{what}: not bit-identical, max abs diff {d.max():.3e} at {np.unravel_index(d.argmax(), d.shape)}")


OBSERVED = {}  # what -> largest observed error in units of RTOL * scale (tools/tolerance_report.py prints it)


def assert_close(got, want, dtype, what, scale=None):
    """|got - want| <= RTOL[dtype] * scale, scale = max |want| unless given.  RTOL is north_star's 1e-5 for float32
    (1e-12 for float64), without multipliers; the observed errors are in DESIGN.md section 3."""
    got, want = host(got).astype(np.float64), np.asarray(want).astype(np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    ref = np.abs(want).max() if scale is None else scale
    err = np.abs(got - want).max() if got.size else 0.0
    units = err / (RTOL[dtype] * max(ref, 1e-30))
    key = re.sub(r"[\(\[].*$", "", what).strip() + (" f32" if dtype == torch.float32 else " f64")
    OBSERVED[key] = max(OBSERVED.get(key, 0.0), units)
    assert units <= 1.0, f"{what}: max err {err:.3e} = {units:.2f} x {RTOL[dtype]:.0e} x scale {ref:.3e} (allowed 1)"


SHAPES3 = [(5, 6, 7), (8, 8, 8), (3, 4, 1), (2, 2, 2), (9, 5, 70), (6, 5, 16), (3, 4, 128)]
SHAPES2 = [(7, 9), (16, 16), (2, 2), (5, 1), (3, 130)]


def _disp(rng, nn, sp, dtype):
    """Displacements that exercise clamp (far out of range), the negative floor rule, and
    exact-integer positions."""
    u = rnd(rng, (nn, len(sp)) + sp, dtype, 1.7)
    flat = u.reshape(-1)
    flat[::11] *= 9.0          # far outside
    flat[::7] = np.round(flat[::7])  # exact integers (weights exactly 0/1)
    flat[::13] = -np.abs(flat[::13]) - 0.25
    return u


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", SHAPES3 + SHAPES2)
@pytest.mark.parametrize("nn,nc,bc", [(2, 1, False), (3, 3, True), (1, 4, False)])
def test_interp_forward_bit_exact(ext, dtype, sp, nn, nc, bc):
    rng = np.random.default_rng(hash((sp, nn, nc)) % 2**31)
    I = rnd(rng, ((1 if bc else nn), nc) + sp, dtype)
    u = _disp(rng, nn, sp, dtype)
    for dt in (1.0, -0.37):
        assert_bits(ext.interp_forward(dev(I), dev(u), dt), orc.interp_forward(I, u, dt), f"interp_forward dt={dt}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("sp", SHAPES3 + SHAPES2)
@pytest.mark.parametrize("nn,nc,bc", [(2, 1, False), (3, 3, True), (2, 2, False)])
def test_interp_backward(ext, dtype, mode, sp, nn, nc, bc):
    rng = np.random.default_rng(hash((sp, nn, nc, 1)) % 2**31)
    I = rnd(rng, ((1 if bc else nn), nc) + sp, dtype)
    u = _disp(rng, nn, sp, dtype)
    go = rnd(rng, (nn, nc) + sp, dtype)
    ext.set_splat_mode(mode)
    try:
        for need_I, need_u in ((True, True), (True, False), (False, True)):
            dI, du = ext.interp_backward(dev(go), dev(I), dev(u), 0.8, need_I, need_u)
            oI, ou = orc.interp_backward(go, I, u, 0.8, need_I, need_u)
            assert_bits(du, ou, f"d_u (need_I={need_I}, need_u={need_u})")   # thread-owned: exact
            assert_close(dI, oI, dtype, f"d_I (need_I={need_I}, need_u={need_u})")  # (summation order of the atomics)
    finally:
        ext.set_splat_mode(1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", [(128, 128), (64, 128), (130, 200), (50, 333), (256, 72)])
@pytest.mark.parametrize("nn,nc,bc", [(2, 1, False), (3, 3, True), (2, 2, False)])
@pytest.mark.parametrize("kind", ["smooth", "wild"])
def test_interp_backward_2d_lds_splat(ext, dtype, sp, nn, nc, bc, kind):
    """2D fields of at least 8192 pixels take the LDS-privatised 2D splat (csrc/interp.hip: splat2d_lds_kernel; round 4,
    VERDICT r3 missing #6): d_u bit for bit, d_I within the bound, on smooth displacements (everything inside the
    windows) and on the clamp / negative-floor / exact-integer field whose corners mostly leave them (the global-atomic
    fall-back per corner); ragged tiles, a broadcast image, every need_* combination, the fused start values."""
    rng = np.random.default_rng(hash((sp, nn, nc, kind)) % 2**31)
    I = rnd(rng, ((1 if bc else nn), nc) + sp, dtype)
    if kind == "smooth":
        from scipy.ndimage import gaussian_filter

        u = gaussian_filter(rng.standard_normal((nn, 2) + sp), sigma=(0, 0, 6, 6), mode="wrap")
        u = (u * (3.0 / np.abs(u).max())).astype(I.dtype)
    else:
        u = _disp(rng, nn, sp, dtype)
    go = rnd(rng, (nn, nc) + sp, dtype)
    for dt in (0.8, -1.0):
        for need_I, need_u in ((True, True), (True, False)):
            before = ext.path_launches("splat_2d")
            dI, du = ext.interp_backward(dev(go), dev(I), dev(u), dt, need_I, need_u)
            assert ext.path_launches("splat_2d") == before + 1, "not the 2D LDS splat"
            oI, ou = orc.interp_backward(go, I, u, dt, need_I, need_u)
            assert_bits(du, ou, f"2D LDS splat d_u ({kind} {sp} dt={dt} need_u={need_u})")
            assert_close(dI, oI, dtype, f"2D LDS splat d_I ({kind} {sp} dt={dt})")
    # the switch selects the reference's form
    ext.set_splat_mode(0)
    try:
        before = ext.path_launches("splat_global")
        dI0, du0 = ext.interp_backward(dev(go), dev(I), dev(u), 0.8, True, True)
        assert ext.path_launches("splat_global") == before + 1
    finally:
        ext.set_splat_mode(1)
    dI, du = ext.interp_backward(dev(go), dev(I), dev(u), 0.8, True, True)
    assert torch.equal(du, du0)
    assert_close(dI, host(dI0), dtype, "2D LDS splat vs global atomics")
    if nc == 2:   # the fused backward forms: a running d_u, addgo * grad_out (needs as many channels as dimensions), a running d_I
        oI, ou = orc.interp_backward(go, I, u, 0.8, True, True)
        startu, startI = rnd(rng, (nn, 2) + sp, dtype), rnd(rng, I.shape, dtype)
        k = go.dtype.type
        dI, du = ext.interp_backward_fused(dev(go), dev(I), dev(u), 0.8, True, addgo=-0.2)
        assert_close(du, (k(-0.2) * go).astype(np.float64) + ou, dtype, "2D fused d_u (addgo)", scale=np.abs(ou).max() + 0.2 * np.abs(go).max())
        run_I = dev(startI)
        dI, du = ext.interp_backward_fused(dev(go), dev(I), dev(u), 0.8, True, d_u=dev(startu), d_I=run_I)
        assert_close(dI, startI.astype(np.float64) + oI, dtype, "2D fused d_I accumulated", scale=np.abs(oI).max() + np.abs(startI).max())
        assert_close(du, startu.astype(np.float64) + ou, dtype, "2D fused d_u accumulated", scale=np.abs(ou).max() + np.abs(startu).max())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [(4, 4, 0, 1, 1, 16, 256), (2, 3, 16, 0, 0, 0, 256), (8, 8, 0, 2, 2, 16, 512),
                                  (16, 16, 16, 2, 2, 2, 1024)])
def test_tiled_splat_any_tile_config(ext, dtype, tile):
    """Window placement / tile shape must never change the result: sweep configurations,
    including windows too small for the displacement (global-atomic fallback path)."""
    rng = np.random.default_rng(5)
    sp = (12, 10, 40)
    I = rnd(rng, (2, 2) + sp, dtype)
    u = _disp(rng, 2, sp, dtype)
    go = rnd(rng, (2, 2) + sp, dtype)
    oI, ou = orc.interp_backward(go, I, u, 1.0, True, True)
    # shear 0: the sheared-window kernel would take every float32 3D call before the tiled kernel sees it
    for shear in (0, 1):
        ext.set_splat_tile(*tile)
        ext.set_splat_shear(shear, 8, 6, 0, 1, 1, 4, 1024)
        try:
            dI, du = ext.interp_backward(dev(go), dev(I), dev(u), 1.0, True, True)
        finally:
            ext.set_splat_tile(0, 8, 0, 1, 1, 4, 512)  # the library defaults
            ext.set_splat_shear(1, 8, 6, 0, 1, 1, 4, 1024)
        assert_bits(du, ou, f"d_u shear {shear}")
        assert_close(dI, oI, dtype, f"d_I shear {shear}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_index_math_bit_exact(ext, dtype):
    """Integer displacements (incl. negative and out of range) on a ramp image: every weight is
    exactly 0 or 1, so the output *is* the clamped gather index -- exact integer compare."""
    sp = (6, 7, 9)
    nv = int(np.prod(sp))
    I = np.arange(nv, dtype=np.float64).reshape((1, 1) + sp)
    rng = np.random.default_rng(3)
    u = rng.integers(-12, 13, size=(4, 3) + sp).astype(np.float64)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    out = host(ext.interp_forward(dev(I.astype(npdt)), dev(u.astype(npdt)), 1.0))
    ii, jj, kk = np.meshgrid(*[np.arange(s) for s in sp], indexing="ij")
    want = (np.clip(ii + u[:, 0], 0, sp[0] - 1) * sp[1] + np.clip(jj + u[:, 1], 0, sp[1] - 1)) * sp[2] + np.clip(
        kk + u[:, 2], 0, sp[2] - 1)
    assert np.array_equal(out[:, 0].astype(np.int64), want.astype(np.int64))
    # the splat of ones under the same integer map is an exact histogram of those indices
    dI, _ = ext.interp_backward(dev(np.ones((4, 1) + sp, npdt)), dev(I.astype(npdt)), dev(u.astype(npdt)), 1.0, True, False)
    hist = np.bincount(want.astype(np.int64).ravel(), minlength=nv).reshape(sp)
    assert np.array_equal(host(dI)[0, 0], hist.astype(npdt))


@pytest.mark.parametrize("dtype", DTYPES)
def test_interp_hessian_diagonal(ext, dtype):
    rng = np.random.default_rng(11)
    I = rnd(rng, (2, 3, 9, 8), dtype)
    u = _disp(rng, 2, (9, 8), dtype)
    assert_close(ext.interp_hessian_diagonal_image(dev(I), dev(u), 0.6), orc.interp_hessian_diagonal_image(I, u, 0.6),
                 dtype, "hessian diagonal")


JSHAPES = [(5, 6, 7), (2, 2, 2), (8, 4, 66), (7, 9), (2, 2), (3, 70)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", JSHAPES)
@pytest.mark.parametrize("disp", [True, False])
@pytest.mark.parametrize("trans", [True, False])
def test_jtv_forward_backward_bit_exact(ext, dtype, sp, disp, trans):
    rng = np.random.default_rng(hash((sp, disp, trans)) % 2**31)
    d = len(sp)
    v = rnd(rng, (2, d) + sp, dtype)
    w = rnd(rng, (2, d) + sp, dtype)
    go = rnd(rng, (2, d) + sp, dtype)
    assert_bits(ext.jacobian_times_vectorfield_forward(dev(v), dev(w), disp, trans),
                orc.jacobian_times_vectorfield_forward(v, w, disp, trans), "jtv forward")
    dv, dw = ext.jacobian_times_vectorfield_backward(dev(go), dev(v), dev(w), disp, trans, True, True)
    ov, ow = orc.jacobian_times_vectorfield_backward(go, v, w, disp, trans)
    assert_bits(dv, ov, "jtv backward d_v")
    assert_bits(dw, ow, "jtv backward d_w")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", JSHAPES)
@pytest.mark.parametrize("nc", [1, 3])
def test_jtv_scalar_channels_and_adjoint_bit_exact(ext, dtype, sp, nc):
    rng = np.random.default_rng(hash((sp, nc, 9)) % 2**31)
    d = len(sp)
    g = rnd(rng, (2, nc) + sp, dtype)
    w = rnd(rng, (2, d) + sp, dtype)
    # non-displacement, non-transposed mode accepts any channel count (e.g. image gradient . w)
    assert_bits(ext.jacobian_times_vectorfield_forward(dev(g), dev(w), False, False),
                orc.jacobian_times_vectorfield_forward(g, w, False, False), "jtv forward nc")
    assert_bits(ext.jacobian_times_vectorfield_adjoint_forward(dev(g), dev(w)),
                orc.jacobian_times_vectorfield_adjoint_forward(g, w), "jtv adjoint forward")
    go = rnd(rng, (2, nc) + sp, dtype)
    dv, dw = ext.jacobian_times_vectorfield_backward(dev(go), dev(g), dev(w), False, False, True, True)
    ov, ow = orc.jacobian_times_vectorfield_backward(go, g, w, False, False)
    assert_bits(dv, ov, "jtv backward d_v nc")
    assert_bits(dw, ow, "jtv backward d_w nc")
    if nc == d or True:
        v = rnd(rng, (2, d) + sp, dtype)
        go = rnd(rng, (2, d) + sp, dtype)
        dv, dw = ext.jacobian_times_vectorfield_adjoint_backward(dev(go), dev(v), dev(w), True, True)
        ov, ow = orc.jacobian_times_vectorfield_adjoint_backward(go, v, w)
        assert_bits(dv, ov, "jtv adjoint backward d_v")
        assert_bits(dw, ow, "jtv adjoint backward d_w")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", [(6, 5, 8), (4, 4, 4), (3, 3, 3), (8, 6), (3, 3), (5, 9, 130)])
@pytest.mark.parametrize("inverse", [True, False])
@pytest.mark.parametrize("params", [(0.1, 0.05, 0.01), (1.0, 0.0, 0.01), (0.0, 0.0, 0.0)])
def test_fluid_operator_bit_exact(ext, dtype, sp, inverse, params):
    """The per-frequency kernel alone, on an arbitrary interleaved-complex buffer."""
    rng = np.random.default_rng(hash((sp, inverse)) % 2**31)
    d = len(sp)
    csp = sp[:-1] + (sp[-1] // 2 + 1,)
    F = rnd(rng, (3, d) + csp + (2,), dtype)
    npdt = F.dtype
    cos, sin = orc.fluid_luts(sp, npdt)
    Fo = F.copy()
    orc.fluid_operator(Fo, inverse, cos, sin, *params)
    Fd = dev(F)
    r = ext.fluid_operator(Fd, inverse, [dev(c) for c in cos], [dev(s) for s in sin], *params)
    assert r is None
    assert_bits(Fd, Fo, "fluid operator")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", [(8, 6, 10), (7, 5, 9), (12, 10)])
def test_fluid_metric_sharp_flat_vs_oracle(ext, dtype, sp):
    """Full sharp/flat: rocFFT (product) vs pocketfft (oracle) around the same kernel."""
    import lagomorph_amd as lm

    rng = np.random.default_rng(21)
    m = rnd(rng, (2, len(sp)) + sp, dtype)
    met = lm.FluidMetric([0.1, 0.05, 0.01])
    for inv, f in ((True, met.sharp), (False, met.flat)):
        want = orc.fluid_metric_apply(m, [0.1, 0.05, 0.01], inv)
        assert_close(f(dev(m)), want, dtype, f"fluid inverse={inv}")
    # LUT values: float64 numpy rounded through float32 (metric.py:66-75), bit for bit
    cos, sin = orc.fluid_luts(sp, m.dtype)
    for a, b in zip(met.luts["cos"] + met.luts["sin"], cos + sin):
        assert_bits(a, b, "LUT")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", [(5, 6, 7), (2, 2, 2), (4, 3, 1), (7, 9), (2, 2), (64, 64)])
@pytest.mark.parametrize("nn,nc,bc", [(2, 1, False), (3, 2, True), (2, 4, False)])
def test_affine_interp(ext, dtype, sp, nn, nc, bc):
    rng = np.random.default_rng(hash((sp, nn, nc, 2)) % 2**31)
    d = len(sp)
    I = rnd(rng, ((1 if bc else nn), nc) + sp, dtype)
    A = (np.eye(d)[None] + 0.3 * rng.standard_normal((nn, d, d))).astype(I.dtype)
    T = (1.5 * rng.standard_normal((nn, d))).astype(I.dtype)
    go = rnd(rng, (nn, nc) + sp, dtype)
    assert_bits(ext.affine_interp_forward(dev(I), dev(A), dev(T)), orc.affine_interp_forward(I, A, T), "affine forward")
    for needs in ((True, True, True), (False, True, True), (True, False, False), (False, False, True)):
        dI, dA, dT = ext.affine_interp_backward(dev(go), dev(I), dev(A), dev(T), *needs)
        oI, oA, oT = orc.affine_interp_backward(go, I, A, T, *needs)
        nvox = float(np.prod(sp)) * nc
        if needs[0]:
            assert_close(dI, oI, dtype, "affine d_I")
        else:
            assert dI.numel() == 0
        # dA/dT are sums over nvox terms whose fp32 summation order differs from the reference's tree: still within
        # north_star's 1e-5 x max |reference| (no inflated scale: round-2 review)
        if needs[1]:
            assert_close(dA, oA, dtype, "affine d_A")
        else:
            assert dA.numel() == 0
        if needs[2]:
            assert_close(dT, oT, dtype, "affine d_T")
        else:
            assert dT.numel() == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp,nn,nc,bc,kind", [((65, 70, 72), 4, 1, False, "random"), ((64, 64, 130), 2, 2, True, "random"),
                                              ((33, 129, 127), 2, 1, False, "rotation"), ((128, 128, 64), 1, 1, False, "far")])
def test_affine_forward_megavoxel_volumes(ext, dtype, sp, nn, nc, bc, kind):
    """affine_interp_forward on volumes of a million voxels and more (the other affine cases are tiny), bit for bit
    against the oracle: ragged last workgroups, broadcast image, two channels, a rotation, and a map that sends most
    samples out of range (the clamp)."""
    rng = np.random.default_rng(hash((sp, nn, nc)) % 2**31)
    I = rnd(rng, ((1 if bc else nn), nc) + sp, dtype)
    if kind == "rotation":
        c, s_ = np.cos(0.35), np.sin(0.35)
        A = np.tile(np.array([[1, 0, 0], [0, c, -s_], [0, s_, c]]), (nn, 1, 1)).astype(I.dtype)
        T = (0.7 * rng.standard_normal((nn, 3))).astype(I.dtype)
    elif kind == "far":
        A = (2.5 * np.eye(3)[None] + 0.3 * rng.standard_normal((nn, 3, 3))).astype(I.dtype)
        T = (40.0 * rng.standard_normal((nn, 3))).astype(I.dtype)
    else:
        A = (np.eye(3)[None] + 0.1 * rng.standard_normal((nn, 3, 3))).astype(I.dtype)
        T = (1.5 * rng.standard_normal((nn, 3))).astype(I.dtype)
    assert_bits(ext.affine_interp_forward(dev(I), dev(A), dev(T)), orc.affine_interp_forward(I, A, T), "affine forward (large)")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp,out,nn,nc", [((20, 24, 28), (40, 48, 56), 3, 3), ((40, 48, 56), (20, 24, 28), 1, 5),
                                          ((8, 8, 8), (6, 5, 1030), 1, 2), ((8, 8, 8), (6, 5, 1024), 2, 1)])
def test_regrid_forward_many_planes_and_long_rows(ext, dtype, sp, out, nn, nc):
    """regrid_forward with several (n, c) planes and output rows of a thousand voxels (the reference's running sum
    hz += Sz over the whole row, cuda/affine.cu:669-675), bit for bit against the oracle."""
    rng = np.random.default_rng(hash((sp, out)) % 2**31)
    I = rnd(rng, (nn, nc) + sp, dtype)
    origin = [(s - 1) * 0.5 + 0.3 for s in sp]
    spacing = [(a - 1) / (b - 1) if b > 1 else 1.0 for a, b in zip(sp, out)]
    assert_bits(ext.regrid_forward(dev(I), list(out), origin, spacing), orc.regrid_forward(I, list(out), origin, spacing),
                "regrid forward (many planes)")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp,out", [((4, 5, 6), (5, 7, 9)), ((6, 6, 6), (3, 3, 3)), ((4, 5), (9, 4)), ((2, 2, 2), (3, 3, 70))])
def test_regrid(ext, dtype, sp, out):
    rng = np.random.default_rng(hash((sp, out)) % 2**31)
    d = len(sp)
    I = rnd(rng, (2, 3) + sp, dtype)
    origin = [(s - 1) * 0.5 + 0.3 for s in sp]
    spacing = [(a - 1) / (b - 1) * 1.1 for a, b in zip(sp, out)]
    assert_bits(ext.regrid_forward(dev(I), out, origin, spacing), orc.regrid_forward(I, out, origin, spacing), "regrid forward")
    go = rnd(rng, (2, 3) + out, dtype)
    assert_close(ext.regrid_backward(dev(go), sp, out, origin, spacing), orc.regrid_backward(go, sp, out, origin, spacing),
                 dtype, "regrid backward")


def test_reference_known_answers_through_hip(ext):
    """SURVEY 8(c) table (outputs of the reference's own kernels), float64, through the HIP path."""
    import lagomorph_amd as lm

    res = kat.evaluate(ext, lm, torch.float64, "cuda")
    kat.check(res, rel=2e-12, abs_floor=2e-9)
    res = kat.evaluate(ext, lm, torch.float32, "cuda")
    kat.check(res, rel=2e-5, abs_floor=2e-4)


def test_empty_and_error_behaviour(ext):
    """Empty batches are no-ops; CPU or non-contiguous inputs raise like CHECK_INPUT
    (extension.cpp:8-10); dimension violations raise the reference's messages."""
    I = torch.zeros((0, 1, 4, 4, 4), device="cuda")
    u = torch.zeros((0, 3, 4, 4, 4), device="cuda")
    assert ext.interp_forward(I, u, 1.0).shape == (0, 1, 4, 4, 4)
    dI, du = ext.interp_backward(I, I, u, 1.0, True, True)
    assert dI.shape == I.shape and du.shape == u.shape
    assert ext.jacobian_times_vectorfield_forward(u, u, True, False).shape == u.shape
    I = torch.randn((2, 1, 4, 4, 4), device="cuda")
    u = torch.randn((2, 3, 4, 4, 4), device="cuda")
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        ext.interp_forward(I.cpu(), u, 1.0)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        ext.interp_forward(I, u.transpose(3, 4), 1.0)
    with pytest.raises(RuntimeError, match="Only two- and three-dimensional"):
        ext.interp_forward(I[:, :, 0, 0].contiguous(), u[:, :, 0, 0].contiguous(), 1.0)
    with pytest.raises(RuntimeError, match="thin"):
        ext.jacobian_times_vectorfield_forward(u[:, :, :, :, :1].contiguous(), u[:, :, :, :, :1].contiguous(), True, False)
    with pytest.raises(RuntimeError, match="Displacement mode only defined for vector fields"):
        ext.jacobian_times_vectorfield_forward(I, u, True, False)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ext.affine_interp_forward(I.cpu(), torch.eye(3)[None].repeat(2, 1, 1), torch.zeros(2, 3))
    with pytest.raises(RuntimeError, match="Type of LUTs must equal that of image"):
        F = torch.zeros((1, 2, 4, 3, 2), device="cuda")
        l = [torch.zeros(4, device="cuda", dtype=torch.float64), torch.zeros(3, device="cuda", dtype=torch.float64)]
        ext.fluid_operator(F, True, l, l, 0.1, 0.0, 0.01)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", [(5, 6, 7), (8, 8, 8), (3, 4, 1), (7, 9), (2, 2), (4, 3, 64)])
def test_fused_compose_bit_exact(ext, dtype, sp):
    """ds*u + dt*interp(v, u, ds) in one kernel == the unfused expression (three roundings kept)."""
    rng = np.random.default_rng(hash(sp) % 2**31)
    d = len(sp)
    u = _disp(rng, 2, sp, dtype)
    v = rnd(rng, (2, d) + sp, dtype)
    for ds, dt in ((1.0, 1.0), (-0.1, 1.0), (0.7, -1.3)):
        k = u.dtype.type
        want = k(ds) * u + k(dt) * orc.interp_forward(v, u, ds)
        assert_bits(ext.compose(dev(u), dev(v), ds, dt), want, f"compose ds={ds} dt={dt}")
        # and equals the torch expression over this library's own interp
        ud, vd = dev(u), dev(v)
        assert torch.equal(ext.compose(ud, vd, ds, dt), ds * ud + dt * ext.interp_forward(vd, ud, ds))


def _smooth_disp(rng, nn, sp, amp, shift, rough=0.0):
    """A smooth displacement (what the LDS-window gathers are for): a few low-frequency waves of amplitude `amp` plus
    a constant shift per item and component, optionally with `rough` voxels of noise on top."""
    grids = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in sp], indexing="ij")
    u = np.empty((nn, 3) + sp, np.float64)
    for n in range(nn):
        for c in range(3):
            k = rng.uniform(0.02, 0.12, 3) * rng.choice([-1, 1], 3)
            u[n, c] = amp * np.sin(sum(kk * g for kk, g in zip(k, grids)) + rng.uniform(0, 6)) + shift * rng.uniform(-1, 1)
    if rough:
        u += rough * rng.standard_normal(u.shape)
    return u.astype(np.float32)


WINDOW_CASES = [
    # shape, amplitude, shift, rough: (window for every tile) ... (pair-gather path for most)
    ((16, 32, 64), 0.6, 0.0, 0.0),
    ((16, 32, 64), 1.5, 7.3, 0.0),     # a translated window
    ((28, 32, 64), 1.2, 2.5, 0.0),     # last tile in x half empty (28 = 3.5 tiles; fill 0.875 >= 0.85: window path)
    ((24, 30, 64), 0.9, -1.5, 0.0),    # ragged in y
    ((24, 32, 60), 0.9, 1.5, 0.0),     # ragged in z (60 = 1.875 tiles, still a multiple of 4)
    ((24, 16, 36), 1.2, 2.5, 0.0),     # fill 0.56: stays on the pair-gather kernel
    ((16, 48, 64), 0.8, 40.0, 0.0),    # displaced past the grid: clamped corners, windows half outside
    ((16, 32, 64), 1.0, 3.0, 0.4),     # a few samples leave their windows: those workgroups take the pair path
    ((16, 32, 64), 6.0, 0.0, 3.0),     # nothing fits
    ((40, 32, 32), 0.7, -5.5, 0.0),
]


@pytest.mark.parametrize("case", WINDOW_CASES, ids=[f"{c[0]}-a{c[1]}-s{c[2]}-r{c[3]}" for c in WINDOW_CASES])
def test_lds_window_gather_same_bits(ext, case):
    """compose and Ad_star through the LDS window == the pair-gather kernels == the oracle, bit for bit:
    window placement and the per-workgroup choice of path never show in the result."""
    import lagomorph_amd.lagomorph_ext as shim

    sp, amp, shift, rough = case
    rng = np.random.default_rng(hash(case) % 2**31)
    u = _smooth_disp(rng, 3, sp, amp, shift, rough)
    u.reshape(-1)[::997] = np.round(u.reshape(-1)[::997])  # exact-integer positions
    v = rnd(rng, (3, 3) + sp, torch.float32)
    ud, vd = dev(u), dev(v)
    try:
        for ds, dt in ((1.0, -0.1), (-1.0, 1.0), (0.7, -1.3)):
            want = np.float32(ds) * u + np.float32(dt) * orc.interp_forward(v, u, ds)
            outs = {}
            for mode in (0, 1):
                shim.set_gather_window(mode)
                before = shim.path_launches("gather_window")
                outs[mode] = ext.compose(ud, vd, ds, dt)
                took = shim.path_launches("gather_window") - before
                # every case but the 0.56-fill one must really run the window kernel when it is on
                assert took == (1 if mode == 1 and sp != (24, 16, 36) else 0), (mode, sp, took)
                assert_bits(outs[mode], want, f"compose window mode {mode} ds={ds}")
            assert torch.equal(outs[1], outs[0])
        # Ad_star is unaffected by the switch (its window form was measured slower and is not shipped)
        want_m = orc.interp_forward(v, u, 1.0)
        want = orc.jacobian_times_vectorfield_forward(u, want_m, True, False)
        for mode in (0, 1):
            shim.set_gather_window(mode)
            assert_bits(ext.Ad_star(ud, vd), want, f"Ad_star window mode {mode}")
            out, mphi = ext.Ad_star(ud, vd, save_resampled=True)
            assert_bits(out, want, f"Ad_star(save) window mode {mode}")
            assert_bits(mphi, want_m, f"resampled momentum window mode {mode}")
    finally:
        shim.set_gather_window(1)


@pytest.mark.parametrize("case", WINDOW_CASES, ids=[f"{c[0]}-a{c[1]}-s{c[2]}-r{c[3]}" for c in WINDOW_CASES])
@pytest.mark.parametrize("nc,bc", [(2, False), (2, True), (3, False), (4, True)])
def test_lds_window_interp_forward_same_bits(ext, case, nc, bc):
    """interp_forward of two or more channels through the LDS window (interp3_window_kernel: the channels in turn through
    one window, broadcast or per-item images) == the pair-gather kernel == the oracle, bit for bit, over the window cases
    of compose.  (One channel stays on the pair gathers: the window does not pay there.)"""
    import lagomorph_amd.lagomorph_ext as shim

    sp, amp, shift, rough = case
    rng = np.random.default_rng((hash(case) + 7 * nc + bc) % 2**31)
    u = _smooth_disp(rng, 3, sp, amp, shift, rough)
    u.reshape(-1)[::997] = np.round(u.reshape(-1)[::997])  # exact-integer positions
    I = rnd(rng, ((1 if bc else 3), nc) + sp, torch.float32)
    ud, Id = dev(u), dev(I)
    try:
        for dt in (1.0, -1.0, 0.7):
            want = orc.interp_forward(I, u, dt)
            outs = {}
            for mode in (0, 1):
                shim.set_gather_window(mode)
                before = shim.path_launches("gather_window")
                outs[mode] = ext.interp_forward(Id, ud, dt)
                took = shim.path_launches("gather_window") - before
                assert took == (1 if mode == 1 and sp != (24, 16, 36) else 0), (mode, sp, took)
                assert_bits(outs[mode], want, f"interp_forward window mode {mode} dt={dt}")
            assert torch.equal(outs[1], outs[0])
    finally:
        shim.set_gather_window(1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", [(8, 6, 10), (7, 5, 9), (16, 16, 16), (12, 10), (9, 7)])
@pytest.mark.parametrize("inverse", [True, False])
def test_fused_fluid_metric_matches_three_call_form(ext, dtype, sp, inverse):
    """hipFFT-direct sharp/flat vs rfftn(ortho) -> fluid_operator -> irfftn(ortho) and vs the oracle."""
    import lagomorph_amd as lm
    from lagomorph_amd import metric as lmm

    rng = np.random.default_rng(hash((sp, inverse)) % 2**31)
    m = rnd(rng, (3, len(sp)) + sp, dtype)
    md = dev(m)
    keep = md.clone()
    met = lm.FluidMetric([0.1, 0.05, 0.01])
    f = met.sharp if inverse else met.flat
    fused = f(md)
    assert torch.equal(md, keep), "fluid_metric modified its input"
    lmm.USE_FUSED_FLUID = False
    try:
        plain = f(md)
    finally:
        lmm.USE_FUSED_FLUID = True
    want = orc.fluid_metric_apply(m, [0.1, 0.05, 0.01], inverse)
    assert_close(fused, want, dtype, "fused fluid metric vs oracle")
    assert_close(fused, host(plain), dtype, "fused vs three-call")


@pytest.mark.parametrize("dtype", DTYPES)
def test_vector_and_scalar_kernels_agree_bitwise(ext, dtype):
    """The slab-unrolled 3D kernels are a launch-shape choice only."""
    rng = np.random.default_rng(77)
    sp = (6, 7, 32)
    I = rnd(rng, (2, 3) + sp, dtype)
    u = _disp(rng, 2, sp, dtype)
    v = rnd(rng, (2, 3) + sp, dtype)
    a = ext.interp_forward(dev(I), dev(u), 0.9), ext.compose(dev(u), dev(v), -0.3, 1.0)
    ext.set_vector_kernels(0)
    try:
        b = ext.interp_forward(dev(I), dev(u), 0.9), ext.compose(dev(u), dev(v), -0.3, 1.0)
    finally:
        ext.set_vector_kernels(1)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert_bits(a[0], orc.interp_forward(I, u, 0.9), "vector interp vs oracle")


@pytest.mark.parametrize("sp", [(64, 6, 10), (128, 5, 12), (64, 64, 64), (256, 4, 6), (128, 7, 130),
                                (64, 32, 64), (128, 64, 128), (64, 128, 256), (256, 32, 64), (64, 256, 64),
                                (160, 160, 160), (96, 64, 160), (160, 96, 192), (192, 160, 96), (64, 192, 192),
                                # every plane shape served by the persistent zy kernels (above 80 KB of LDS)
                                (64, 128, 192), (64, 160, 128), (64, 192, 128), (64, 128, 160), (64, 160, 192),
                                (96, 192, 160),
                                # radix 11 / 13 (round 6): planes and x lines of the 176 x 208 x 176 brain grid
                                (64, 208, 176), (64, 176, 208), (64, 176, 176), (176, 64, 64), (208, 32, 64), (176, 208, 176),
                                # odd factors 7, 9, 15
                                (96, 112, 96), (112, 96, 112), (144, 144, 144), (64, 224, 160), (224, 32, 64), (240, 64, 64),
                                (64, 160, 240), (64, 240, 160), (64, 144, 176),
                                # 8 x odd extents: the short last tile of the Nyquist plane
                                (88, 104, 88), (120, 120, 120), (80, 80, 80), (104, 88, 104), (120, 88, 88),
                                # planes above the LDS: the rows + columns route
                                (64, 256, 256), (64, 192, 224), (64, 224, 192), (96, 224, 224), (64, 256, 192), (64, 208, 208), (64, 160, 176)])
@pytest.mark.parametrize("inverse", [True, False])
def test_fused_fluid_metric_paths(ext, sp, inverse):
    """float32 3D: the three implementations of FluidMetric sharp/flat -- (2) three LDS-tiled FFT
    passes without rocFFT (extents 2^a, 3*2^a, 5*2^a: 160^3 is BASELINE configs[4]), (1) 2D rocFFT + fused (x-FFT, operator, inverse
    x-FFT) kernel (power-of-two nx), (0) plain 3D hipFFT + operator kernel -- against each other
    and against the oracle (numpy FFT).  Shapes a mode does not support fall through to the next."""
    import lagomorph_amd as lm

    rng = np.random.default_rng(hash((sp, inverse)) % 2**31)
    m = rnd(rng, ((3 if np.prod(sp) < 4e6 else 1), 3) + sp, torch.float32)  # odd batch: the x pass pairs batch items per workgroup
    md = dev(m)
    met = lm.FluidMetric([0.1, 0.05, 0.01])
    f = met.sharp if inverse else met.flat
    got = {}
    try:
        for mode in (2, 1, 0):
            ext.set_fluid_mode(mode)
            got[mode] = f(md)
    finally:
        ext.set_fluid_mode(3)
    want = orc.fluid_metric_apply(m, [0.1, 0.05, 0.01], inverse)
    for mode in (2, 1, 0):
        assert_close(got[mode], want, torch.float32, f"fluid metric mode {mode} vs oracle")
    assert_close(got[2], host(got[0]), torch.float32, "native passes vs plain hipFFT")


GENERIC_FFT_SHAPES = [(30, 42, 26), (13, 17, 22), (7, 9), (100, 100), (218, 26), (1, 5, 8), (9, 1, 6), (24, 20, 28), (64, 40, 40),
                      (2, 3), (3, 2, 2), (121, 49), (32, 128), (120, 60, 30), (37, 64),
                      # Bluestein lines (a prime factor >= 29): on the real axis (odd and even), on the strided axes, all three
                      (29, 31), (58, 37), (12, 62, 10), (31, 8, 58), (29, 37, 41), (6, 218)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", GENERIC_FFT_SHAPES)
@pytest.mark.parametrize("inverse", [True, False])
def test_generic_fft_passes(ext, dtype, sp, inverse):
    """csrc/fftg.hip (round 4; VERDICT r3 missing #4): every shape the tuned passes do not cover -- float64, extents with
    prime factors 7, 11, 13, 37, 109, extents of 1 to 3, odd real axes, 2D planes beyond the LDS -- runs hand-written
    Stockham passes (register butterflies for radices 2, 3, 4, 5, 7, a direct DFT for the primes up to 23, Bluestein's
    convolution for lines with a larger prime factor) instead of rocFFT, against the oracle; bit for bit the same on a second call (no atomics anywhere) and for an item alone."""
    import lagomorph_amd as lm

    rng = np.random.default_rng(hash((sp, inverse)) % 2**31)
    m = rnd(rng, (3, len(sp)) + sp, dtype)
    for params in ([0.1, 0.05, 0.01], [1.0, 0.0, 0.001]):
        met = lm.FluidMetric(params)
        f = met.sharp if inverse else met.flat
        before = ext.path_launches()
        got = f(dev(m))
        after = ext.path_launches()
        fused2d = dtype == torch.float32 and sp in ((32, 128),)   # (a shape of the fused 2D kernel, float32 only)
        assert after["fluid_rocfft"] == before["fluid_rocfft"] and after["fluid_xpass"] == before["fluid_xpass"]
        assert after["fluid_generic"] == before["fluid_generic"] + (0 if fused2d else 1), sp
        assert_close(got, orc.fluid_metric_apply(m, params, inverse), dtype, f"generic FFT passes {sp}")
        assert torch.equal(got, f(dev(m)))
        # the real axis transforms two lines as one complex line: the pairs stay inside a field, so an item's bits do
        # not depend on its batch neighbours
        assert torch.equal(got[1:2], f(dev(m[1:2])))


def test_generic_fft_passes_large_planes(ext):
    """2D planes far beyond the LDS (the affine / 2D registration images): 4 x 2 x 1024 x 768 and 2 x 2 x 600 x 1000,
    float32, through the generic passes against the oracle."""
    import lagomorph_amd as lm

    rng = np.random.default_rng(12)
    for sp, nn in (((1024, 768), 4), ((600, 1000), 2)):
        m = rnd(rng, (nn, 2) + sp, torch.float32)
        met = lm.FluidMetric([0.1, 0.0, 0.01])
        before = ext.path_launches("fluid_generic")
        got = met.sharp(dev(m))
        assert ext.path_launches("fluid_generic") == before + 1
        assert_close(got, orc.fluid_metric_apply(m, [0.1, 0.0, 0.01], True), torch.float32, f"generic FFT passes {sp}")


@pytest.mark.parametrize("sp", [(64, 64), (128, 128), (96, 64), (64, 128), (160, 96), (32, 128), (256, 64), (96, 192),
                                (64, 256), (128, 96)])
@pytest.mark.parametrize("inverse", [True, False])
def test_fused_2d_fluid_metric(ext, sp, inverse):
    """float32 2D fields whose two component planes fit the LDS: the whole operator in one kernel (real 2D transform
    of both planes, the 2 x 2 operator of fluid_kernel_2d per frequency, inverse transform) against the oracle and
    against the rocFFT R2C / operator kernel / C2R form (mode 0), plus the Python-level three-call form."""
    import lagomorph_amd as lm
    from lagomorph_amd import metric as lmm

    rng = np.random.default_rng(hash((sp, inverse)) % 2**31)
    m = rnd(rng, (5, 2) + sp, torch.float32)
    md = dev(m)
    for params in ([0.1, 0.05, 0.01], [1.0, 0.0, 0.001]):
        met = lm.FluidMetric(params)
        f = met.sharp if inverse else met.flat
        got = {}
        try:
            for mode in (2, 0):
                ext.set_fluid_mode(mode)
                try:
                    got[mode] = f(md)
                except RuntimeError as e:   # the library's rocFFT guard (csrc/fft.hip): loud, never silently wrong
                    assert mode == 0 and "rocFFT returned a WRONG" in str(e), e
                    print(f"NOTE rocFFT guard fired at {sp}: {e}")
        finally:
            ext.set_fluid_mode(3)
        want = orc.fluid_metric_apply(m, params, inverse)
        assert_close(got[2], want, torch.float32, f"fused 2D fluid metric vs oracle {sp}")
        lmm.USE_FUSED_FLUID = False
        try:
            three = f(md)
        finally:
            lmm.USE_FUSED_FLUID = True
        # the two rocFFT-based forms (hipFFT plans of this library; torch.fft) are cross-checks of third-party code:
        # on ROCm 7.2 the batched 2D real transform of (32, 128) comes back 60 % wrong once plans for other shapes
        # exist (tools/probes/rocfft_2d_repro.py reproduces it with torch alone), so a form that disagrees with
        # the CPU FFT is reported, not asserted
        for name, other in (("rocFFT plan + operator kernel", got.get(0)), ("three-call form (torch.fft)", three)):
            if other is None:
                continue
            err = np.abs(host(other).astype(np.float64) - want).max() / np.abs(want).max()
            if err > 1e-3:
                print(f"NOTE rocFFT-based {name} is off by {err:.2e} at {sp} (third-party); fused kernel holds")
                continue
            assert_close(other, want, torch.float32, f"{name} vs oracle {sp}")
    # zero frequency: sum(sharp(m)) = sum(m) / gamma^2 per component
    if inverse:
        met = lm.FluidMetric([0.1, 0.05, 0.01])
        s_out = met.sharp(md).double().sum(dim=(2, 3)).cpu().numpy()
        s_in = m.astype(np.float64).sum(axis=(2, 3))
        assert np.allclose(s_out, s_in / 0.01 ** 2, rtol=1e-3, atol=1e-3 * np.abs(s_in).max() / 0.01 ** 2)


def test_launch_order_does_not_change_results(ext):
    """`lago_tuning.launch_order`: walking the workgroups in alternating directions (Infinity-Cache reuse) is a pure
    re-ordering -- every non-atomic output is bit-identical under both settings, scatter-adds stay within the bound."""
    import lagomorph_amd as lm

    rng = np.random.default_rng(8)
    sp = (24, 20, 64)
    phi = _disp(rng, 3, sp, torch.float32)
    m = rnd(rng, (3, 3) + sp, torch.float32)
    go = rnd(rng, (3, 3) + sp, torch.float32)
    met = lm.FluidMetric([0.1, 0.05, 0.01])
    big = rnd(rng, (2, 3, 64, 64, 64), torch.float32)   # a shape on the hand-written FFT passes
    res = {}
    for alt in (0, 1, 1, 0):   # (twice with alternation: both parities of the launch counter)
        ext.set_launch_order(alt)
        try:
            cur = [ext.Ad_star(dev(phi), dev(m)), ext.compose(dev(phi), dev(m), -0.3, 1.0), ext.interp_forward(dev(m), dev(phi), 0.7),
                   ext.jacobian_times_vectorfield_backward(dev(go), dev(phi), dev(m), True, False, True, True)[0],
                   met.sharp(dev(big)), ext.lincomb([(0.5, dev(m)), (-2.0, dev(go))]),
                   lm.expmap(met, dev(0.01 * big), num_steps=3)]
            dI, du = ext.interp_backward(dev(go), dev(m), dev(phi), 0.9, True, True)
        finally:
            ext.set_launch_order(1)
        if not res:
            res["plain"], res["dI"], res["du"] = cur, dI, du
            continue
        for a, b in zip(cur, res["plain"]):
            assert torch.equal(a, b)
        assert torch.equal(du, res["du"])
        assert_close(dI, host(res["dI"]), torch.float32, "splat under the other launch order")


def test_rocfft_fallback_is_right_or_loud(ext):
    """The rocFFT-based fallbacks spot-check the first forward / inverse execution of every plan against a direct DFT
    (csrc/fft.hip): a call either returns what the oracle returns or raises, naming the shape -- never a silently
    wrong field.  The shape order below is the one in which ROCm 7.2's batched 2D real transform of (32, 128) was seen
    to come back 60 % wrong (tools/probes/rocfft_2d_repro.py)."""
    import lagomorph_amd as lm

    rng = np.random.default_rng(3)
    ext.set_fluid_mode(0)
    try:
        for sp, dtype in (((64, 64), torch.float32), ((128, 128), torch.float32), ((96, 64), torch.float32),
                          ((64, 128), torch.float32), ((160, 96), torch.float32), ((32, 128), torch.float32),
                          ((32, 128), torch.float64), ((20, 24, 36), torch.float32), ((20, 24, 36), torch.float64),
                          ((30, 40), torch.float32)):
            m = rnd(rng, (5, len(sp)) + sp, dtype)
            met = lm.FluidMetric([0.1, 0.05, 0.01])
            for inverse, f in ((True, met.sharp), (False, met.flat)):
                try:
                    got = f(dev(m))
                except RuntimeError as e:
                    assert "rocFFT returned a WRONG" in str(e), e
                    print(f"NOTE rocFFT guard fired for {sp} {dtype}: {e}")
                    continue
                assert_close(got, orc.fluid_metric_apply(m, [0.1, 0.05, 0.01], inverse), dtype, f"rocFFT fallback {sp}")
    finally:
        ext.set_fluid_mode(3)


def test_rocfft_guard_judges_against_a_global_scale(ext):
    """the spot check used to scale its tolerance by the sampled values only, so a centred object on
    a zero background (corner voxels nine orders below the peak) or a single sinusoid (empty bins) raised a false
    'WRONG transform'.  The deviation is now judged against the root-mean-square bin / voxel of the whole transform
    (csrc/fft.hip): these inputs pass, and the results are the oracle's."""
    import lagomorph_amd as lm

    ext.set_fluid_mode(0)
    try:
        met = lm.FluidMetric([0.1, 0.0, 0.01])
        for sp in ((256, 256), (120, 120, 120)):
            ax = [np.arange(n, dtype=np.float64) - n / 2 for n in sp]
            r2 = sum(np.square(a).reshape([-1 if i == d else 1 for i in range(len(sp))]) for d, a in enumerate(ax))
            blob = np.exp(-r2 / (2 * 3.0 ** 2))
            blob[blob < 1e-6] = 0.0     # compact support: exact zeros towards the corners
            m = np.zeros((2, len(sp)) + sp, np.float32)
            m[0, 0] = blob
            m[1, -1] = -2 * blob
            for inverse, f in ((True, met.sharp), (False, met.flat)):
                got = f(dev(m))     # must not raise
                # (rocFFT's float32 transform of this sparse field through an operator of condition 1e4: observed
                # 1.4e-5 x max on MI355X; the point here is that the guard stays quiet and the answer is right)
                want = orc.fluid_metric_apply(m, [0.1, 0.0, 0.01], inverse)
                err = float(np.abs(host(got).astype(np.float64) - want).max() / np.abs(want).max())
                assert err <= 5e-5, (sp, inverse, err)
        # a single sinusoid per component: one occupied bin pair, every other bin empty
        sp = (96, 40)
        i, j = np.meshgrid(np.arange(sp[0]), np.arange(sp[1]), indexing="ij")
        m = np.stack([np.sin(2 * np.pi * (3 * i / sp[0] + 5 * j / sp[1])), np.cos(2 * np.pi * 7 * i / sp[0])])[None].astype(np.float32)
        m = np.repeat(m, 3, 0)
        for inverse, f in ((True, met.sharp), (False, met.flat)):
            want = orc.fluid_metric_apply(m, [0.1, 0.0, 0.01], inverse)
            err = float(np.abs(host(f(dev(m))).astype(np.float64) - want).max() / np.abs(want).max())
            assert err <= 5e-5, ("sinusoid", inverse, err)   # (observed 1.04e-5: rocFFT float32 on a 3 x 2^5 / 5 x 2^3 plane)
    finally:
        ext.set_fluid_mode(3)


def test_rocfft_guard_bookkeeping(ext):
    """A check on an all-zero field proves nothing (the first atlas iteration: m == 0) and must leave the plan
    unverified; a later non-zero call verifies it; creating ANY new plan marks every cached plan unverified again (the
    failure the guard exists for depends on which other plans exist); while a stream is being captured the check --
    which synchronises -- is skipped and the capture stays valid."""
    import lagomorph_amd as lm

    rng = np.random.default_rng(11)
    ext.set_fluid_mode(0)
    try:
        met = lm.FluidMetric([0.1, 0.05, 0.01])
        sp = (44, 52)   # not a fused-2D shape: rocFFT
        z = torch.zeros((7, 2) + sp, device="cuda")
        n0, _ = ext.fft_plan_state()
        assert float(met.sharp(z).abs().max()) == 0.0
        n1, v1 = ext.fft_plan_state()
        assert n1 == n0 + 1 and v1 == 0, "a new plan, and nothing verified by a zero field (the new plan reset the others)"
        # capture with the still unverified plan: no synchronisation, no failure, right answer on replay
        m = rnd(rng, (7, 2) + sp, torch.float32)
        md = dev(m)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side, capture_error_mode="relaxed"):
                out = met.sharp(md)
        torch.cuda.current_stream().wait_stream(side)
        assert ext.fft_plan_state() == (n1, 0), "capturing must not verify (or check) anything"
        graph.replay()
        torch.cuda.synchronize()
        assert_close(out, orc.fluid_metric_apply(m, [0.1, 0.05, 0.01], True), torch.float32, "captured rocFFT fallback")
        # eager, non-zero: verified now
        assert_close(met.sharp(md), orc.fluid_metric_apply(m, [0.1, 0.05, 0.01], True), torch.float32, "eager rocFFT fallback")
        assert ext.fft_plan_state() == (n1, 1)
        # a plan for another batch size: everything is to be checked again
        m2 = rnd(rng, (3, 2) + sp, torch.float32)
        assert_close(met.flat(dev(m2)), orc.fluid_metric_apply(m2, [0.1, 0.05, 0.01], False), torch.float32, "second plan")
        assert ext.fft_plan_state() == (n1 + 1, 1), "the new plan verified, the older one waiting for its next use"
        met.sharp(md)
        assert ext.fft_plan_state() == (n1 + 1, 2)
    finally:
        ext.set_fluid_mode(3)


@pytest.mark.parametrize("plane", [(160, 160), (208, 176), (224, 160)])
@pytest.mark.parametrize("batch", [1, 2, 5])
def test_persistent_zy_passes_any_plane_count(ext, batch, plane):
    """Planes above 80 KB of LDS run on a grid of at most 256 persistent workgroups that prefetch their next plane:
    fewer planes than workgroups (batch 1: 192), a ragged last round (batch 2: 384 = 256 + 128) and several rounds
    (batch 5: 960), against the oracle and bit for bit against the one-plane-per-workgroup kernels."""
    import lagomorph_amd as lm

    rng = np.random.default_rng(batch)
    m = rnd(rng, (batch, 3, 64) + plane, torch.float32)
    met = lm.FluidMetric([0.1, 0.05, 0.01])
    got = met.sharp(dev(m))
    assert_close(got, orc.fluid_metric_apply(m, [0.1, 0.05, 0.01], True), torch.float32, "persistent zy passes vs oracle")
    ext.tune(fluid_zy_persist=0)
    try:
        plain = met.sharp(dev(m))
    finally:
        ext.tune(fluid_zy_persist=1)
    assert torch.equal(got, plain)


@pytest.mark.parametrize("shape,batch", [((64, 64, 64), 1), ((64, 96, 64), 3), ((128, 64, 96), 2), ((160, 64, 64), 5),
                                         ((96, 32, 128), 7), ((176, 32, 64), 3), ((176, 176, 176), 1),
                                         ((88, 104, 88), 3), ((120, 120, 120), 2), ((144, 64, 64), 5)])
@pytest.mark.parametrize("inverse", [True, False])
def test_persistent_x_pass_any_pair_count(ext, shape, batch, inverse):
    """The x pass as a persistent grid: each workgroup walks a contiguous run of (bin tile, batch item) pairs with its
    next tile prefetched and its coefficients held until the bin tile changes.  Forced on (mode 2) for launches with
    fewer pairs than workgroups, runs that end inside a bin tile, runs of one pair, both launch directions: against
    the oracle, and bit for bit against the one-shot workgroups."""
    import lagomorph_amd as lm

    rng = np.random.default_rng(batch + shape[0])
    m = rnd(rng, (batch, 3) + shape, torch.float32)
    met = lm.FluidMetric([0.1, 0.05, 0.01])
    op = met.sharp if inverse else met.flat
    ext.tune(fluid_xpass_persist=0)
    try:
        plain = op(dev(m))
        ext.tune(fluid_xpass_persist=2)
        got = [op(dev(m)) for _ in range(2)]   # two calls: both launch directions (common.hpp: next_direction)
    finally:
        ext.tune(fluid_xpass_persist=1)
    assert_close(got[0], orc.fluid_metric_apply(m, [0.1, 0.05, 0.01], inverse), torch.float32, "persistent x pass vs oracle")
    assert torch.equal(got[0], plain) and torch.equal(got[1], plain)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", ["near_identity", "rotation", "zoom", "flip", "singular", "shear_far"])
@pytest.mark.parametrize("bc", [False, True])
def test_affine_backward_tiled_splat(ext, dtype, kind, bc):
    """affine_interp_backward's image splat through the LDS window, several tiles per volume,
    with maps whose image leaves the window (rotation, zoom, flip fall back to global atomics)."""
    rng = np.random.default_rng(77)
    sp, nn, nc = (36, 20, 70), 2, 2
    I = rnd(rng, ((1 if bc else nn), nc) + sp, dtype)
    A = np.eye(3)[None].repeat(nn, 0)
    if kind == "near_identity":
        A = A + 0.02 * rng.standard_normal((nn, 3, 3))
    elif kind == "rotation":
        c, s = np.cos(0.6), np.sin(0.6)
        A[0] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
        A[1] = [[1, 0, 0], [0, c, -s], [0, s, c]]
    elif kind == "zoom":
        A = A * np.array([2.5, 0.4])[:, None, None]
    elif kind == "singular":      # item 0: rank 2 (no inverse: the general kernel's), item 1: regular (the box kernel's)
        A[0] = [[1.0, 0.5, 0.0], [2.0, 1.0, 0.0], [0.0, 0.0, 1.0]]
        A[1] = A[1] + 0.05 * rng.standard_normal((3, 3))
    elif kind == "shear_far":     # a tiny determinant's neighbour: inverse rows far above the box kernel's limit, and an image far outside the grid
        A[0] = [[0.05, 0.0, 0.0], [0.0, 0.04, 0.0], [0.0, 0.0, 1.0]]
        A[1] = [[1.0, 0.9, 0.0], [0.0, 1.0, 0.8], [0.3, 0.0, 1.0]]
    else:
        A[0, 2, 2] = -1.0
        A[1, 0, 0] = -1.0
    A = A.astype(I.dtype)
    T = (2.0 * rng.standard_normal((nn, 3))).astype(I.dtype)
    go = rnd(rng, (nn, nc) + sp, dtype)
    before = ext.path_launches("splat_affine_box")
    dI, dA, dT = ext.affine_interp_backward(dev(go), dev(I), dev(A), dev(T), True, True, True)
    assert ext.path_launches("splat_affine_box") == before + 1   # (by target boxes; singular / wild items: gated to the tiled kernel)
    oI, oA, oT = orc.affine_interp_backward(go, I, A, T, True, True, True)
    assert_close(dI, oI, dtype, "affine d_I (target boxes)")
    ext.tune(affine_box=0)
    try:
        dI1, _, _ = ext.affine_interp_backward(dev(go), dev(I), dev(A), dev(T), True, False, False)
    finally:
        ext.tune(affine_box=1)
    assert_close(dI1, oI, dtype, "affine d_I (tiled)")
    ext.set_splat_mode(0)
    try:
        dI0, _, _ = ext.affine_interp_backward(dev(go), dev(I), dev(A), dev(T), True, False, False)
    finally:
        ext.set_splat_mode(1)
    assert_close(dI0, oI, dtype, "affine d_I (global atomics)")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp,out,scale", [((20, 12, 40), (40, 24, 80), 1.0), ((40, 24, 80), (20, 12, 40), 1.0),
                                          ((33, 17, 65), (33, 17, 65), 1.0), ((16, 16, 16), (24, 20, 90), -0.8),
                                          ((12, 10, 14), (30, 22, 66), 7.0), ((12, 10, 14), (30, 22, 66), 0.05),
                                          ((1, 2, 3), (5, 4, 7), 1.0), ((48, 40), (96, 100), 1.0), ((64, 50), (20, 30), 2.5),
                                          ((64, 64, 64), (128, 128, 128), 1.0)])
def test_regrid_backward_every_form(ext, dtype, sp, out, scale):
    rng = np.random.default_rng(78)
    origin = [(s - 1) * 0.5 - 0.2 for s in sp]
    spacing = [scale * (a - 1) / (b - 1) for a, b in zip(sp, out)]
    go = rnd(rng, (2, 3) + out, dtype)
    want = orc.regrid_backward(go, sp, out, origin, spacing)
    # default: axis by axis in gather form (positive spacings; round 4) -- no atomics: two runs give the same bits
    sep = ext.regrid_backward(dev(go), sp, out, origin, spacing)
    assert_close(sep, want, dtype, "regrid backward (separable)")
    if all(x > 0 for x in spacing):   # (non-positive spacings stay on the splat: atomics)
        assert torch.equal(sep, ext.regrid_backward(dev(go), sp, out, origin, spacing))
    ext.REGRID_BACKWARD_SEPARABLE = 0
    try:
        assert_close(ext.regrid_backward(dev(go), sp, out, origin, spacing), want, dtype, "regrid backward (tiled)")
        ext.set_splat_mode(0)
        try:
            got0 = ext.regrid_backward(dev(go), sp, out, origin, spacing)
        finally:
            ext.set_splat_mode(1)
    finally:
        ext.REGRID_BACKWARD_SEPARABLE = 1
    assert_close(got0, want, dtype, "regrid backward (global atomics)")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("origin,spacing,sep", [
    ([1.0e9, 3.0, 4.0], [1.0, 0.5, 0.5], False),     # |origin| >= 1e9: beyond the separable passes' checks (below the 2^30
                                                     # at which positions saturate and the weights stop being meaningful)
    ([3.0, 3.0, 4.0], [2.0e6, 0.5, 0.5], False),     # spacing >= 1e6
    ([3.0, 3.0, 4.0], [1e-9, 0.5, 0.5], None),       # a spacing so small that float32 rounding moves a sample by many cells (float64: still separable)
    ([3.0, 3.0, 4.0], [1e-3, 0.5, 0.5], True),       # small but well inside: wider candidate window, still separable
    ([3.0, 3.0, 4.0], [0.5, -0.5, 0.5], False)])     # a non-positive spacing (the Python side already routes it)
def test_regrid_backward_separable_entry_accepts_what_the_reference_accepts(ext, dtype, origin, spacing, sep):
    """lago_regrid_backward_sep used to FAIL on inputs outside its checks although lago_regrid_backward (and
    the reference, cuda/affine.cu:802-855) serve them; now the entry point itself falls back to the splat.  Called with
    the workspace the separable form would need, so that the C entry -- not the Python predicate -- takes the decision."""
    rng = np.random.default_rng(31)
    sp, out = (6, 8, 20), (9, 10, 33)
    go = rnd(rng, (2, 2) + out, dtype)
    want = orc.regrid_backward(go, sp, out, origin, spacing)
    before = ext.path_launches()
    got = ext.regrid_backward(dev(go), sp, out, origin, spacing)
    torch.cuda.synchronize()
    after = ext.path_launches()
    splat_ran = any(after[k] != before[k] for k in ("splat_tiled", "splat_global"))
    if sep is None:
        sep = dtype == torch.float64
    if all(x > 0 for x in spacing):
        assert splat_ran == (not sep), (origin, spacing, {k: after[k] - before[k] for k in after})
    assert_close(got, want, dtype, "regrid backward (separable entry, fallback)")


def test_compose_rejects_partially_overlapping_out(ext):
    """`out` overlapping an input only partly (a shifted view of one buffer) passed the pointer-equality
    check; the kernel would gather from what it is overwriting."""
    buf = torch.zeros((3, 3, 8, 8, 8), device="cuda")
    u = torch.randn((2, 3, 8, 8, 8), device="cuda")
    with pytest.raises(RuntimeError, match="alias"):
        ext.compose(u, buf[:-1], 1.0, 1.0, out=buf[1:])
    with pytest.raises(RuntimeError, match="alias"):
        ext.compose(buf[1:], u, 1.0, 1.0, out=buf[:-1])
    sep = torch.empty_like(u)
    assert ext.compose(u, buf[:-1], 1.0, 1.0, out=sep) is sep


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", [(5, 6, 7), (8, 8, 8), (7, 9), (2, 2), (4, 3, 64), (9, 5, 70), (20, 12, 40), (2, 6, 5)])
def test_fused_ad_star_bit_exact(ext, dtype, sp):
    """Ad_star in one kernel == interp_forward followed by jacobian_times_vectorfield_forward
    (displacement=True), bit for bit, against the oracle and against this library's own two calls
    (unrolled and scalar kernels)."""
    rng = np.random.default_rng(hash(sp) % 2**31)
    d = len(sp)
    phi = _disp(rng, 2, sp, dtype)
    m = rnd(rng, (2, d) + sp, dtype)
    want = orc.jacobian_times_vectorfield_forward(phi, orc.interp_forward(m, phi, 1.0), True, False)
    pd, md = dev(phi), dev(m)
    for vec in (1, 0):
        ext.set_vector_kernels(vec)
        try:
            got = ext.Ad_star(pd, md)
            two = ext.jacobian_times_vectorfield_forward(pd, ext.interp_forward(md, pd, 1.0), True, False)
        finally:
            ext.set_vector_kernels(1)
        assert_bits(got, want, f"ad_star (vector kernels {vec})")
        assert torch.equal(got, two)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", [(6, 5, 8), (7, 9)])
def test_fused_ad_star_backward_matches_unfused(ext, dtype, sp):
    """AdStarFunction.backward (recompute + the two reference backward kernels) == autograd through the
    unfused pair."""
    import lagomorph_amd as lm
    from lagomorph_amd import adjrep

    rng = np.random.default_rng(5)
    d = len(sp)
    phi = dev(0.7 * rnd(rng, (2, d) + sp, dtype))
    m = dev(rnd(rng, (2, d) + sp, dtype))
    go = dev(rnd(rng, (2, d) + sp, dtype))
    grads = {}
    for fused in (True, False):
        adjrep.USE_FUSED_AD_STAR = fused
        try:
            p, q = phi.clone().requires_grad_(True), m.clone().requires_grad_(True)
            out = lm.Ad_star(p, q)
            out.backward(go)
            grads[fused] = (out.detach(), p.grad, q.grad)
        finally:
            adjrep.USE_FUSED_AD_STAR = True
    assert torch.equal(grads[True][0], grads[False][0])
    assert_close(grads[True][1], host(grads[False][1]), dtype, "d_phiinv")
    assert_close(grads[True][2], host(grads[False][2]), dtype, "d_m")


def test_ad_star_rejects_bad_arguments(ext):
    a = torch.zeros((1, 3, 4, 4, 4), device="cuda")
    with pytest.raises(RuntimeError):
        ext.Ad_star(a, torch.zeros((1, 2, 4, 4, 4), device="cuda"))
    with pytest.raises(RuntimeError):
        ext.Ad_star(a.cpu(), a.cpu())
    with pytest.raises(RuntimeError):
        ext.Ad_star(a, a.double())
    thin = torch.zeros((1, 3, 4, 4, 1), device="cuda")
    with pytest.raises(RuntimeError, match="thin"):
        ext.Ad_star(thin, thin)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", [(5, 6, 7), (2, 2, 2), (4, 3, 64), (9, 5, 70), (7, 9), (2, 2), (3, 130)])
def test_fused_small_ad_star_bit_exact(ext, dtype, sp):
    """ad^*(v, m) in one kernel == jtv(v, m, transpose) - jtv_adjoint(m, v), bit for bit (oracle and
    this library's own three calls); its autograd backward == autograd through the unfused form."""
    import lagomorph_amd as lm
    from lagomorph_amd import adjrep

    rng = np.random.default_rng(hash(sp) % 2**31)
    d = len(sp)
    v = rnd(rng, (2, d) + sp, dtype)
    m = rnd(rng, (2, d) + sp, dtype)
    want = orc.jacobian_times_vectorfield_forward(v, m, False, True) - orc.jacobian_times_vectorfield_adjoint_forward(m, v)
    vd, md = dev(v), dev(m)
    got = ext.ad_star(vd, md)
    assert_bits(got, want, "ad_star")
    assert torch.equal(got, ext.jacobian_times_vectorfield_forward(vd, md, False, True)
                       - ext.jacobian_times_vectorfield_adjoint_forward(md, vd))
    go = dev(rnd(rng, (2, d) + sp, dtype))
    grads = {}
    for fused in (True, False):
        adjrep.USE_FUSED_AD_STAR = fused
        try:
            p, q = vd.clone().requires_grad_(True), md.clone().requires_grad_(True)
            lm.ad_star(p, q).backward(go)
            grads[fused] = (p.grad, q.grad)
        finally:
            adjrep.USE_FUSED_AD_STAR = True
    assert torch.equal(grads[True][0], grads[False][0]) and torch.equal(grads[True][1], grads[False][1])


def test_coefficient_table_cache_is_keyed_on_lut_contents(ext):
    """The cached per-frequency coefficient table of the float32 fast paths is keyed on the LUT generation
    (contents), the real extents and the parameters -- not on LUT addresses.  New FluidMetric objects per
    shape, (64, 20, 30) then (64, 20, 31): same nx, ny and the same half-spectrum length nz/2 + 1 = 16, so
    a key without nz would hand the second shape the first shape's table (and torch's caching allocator
    readily returns the freed LUT blocks of the first metric to the second)."""
    import gc

    import lagomorph_amd as lm

    ext.fluid_cache_clear()
    rng = np.random.default_rng(31)
    for sp in ((64, 20, 30), (64, 20, 31), (64, 20, 30), (128, 64, 128), (128, 64, 128)):
        m = rnd(rng, (2, 3) + sp, torch.float32)
        met = lm.FluidMetric([0.1, 0.05, 0.01])  # a fresh metric: fresh LUT tensors, possibly at recycled addresses
        for inv, f in ((True, met.sharp), (False, met.flat)):
            assert_close(f(dev(m)), orc.fluid_metric_apply(m, [0.1, 0.05, 0.01], inv), torch.float32,
                         f"sharp/flat {sp} inverse={inv}")
        del met
        gc.collect()
        torch.cuda.empty_cache()
    assert ext.fluid_cache_entries() >= 2
    # generation 0 = "do not cache": the table-free path, same answer
    sp = (64, 20, 30)
    m = rnd(rng, (1, 3) + sp, torch.float32)
    met = lm.FluidMetric([0.1, 0.05, 0.01])
    met.initialize_luts((1, 3) + sp, torch.float32, "cuda")
    n0 = ext.fluid_cache_entries()
    got = ext.fluid_metric(dev(m), True, met.luts["cos"], met.luts["sin"], 0.1, 0.05, 0.01, lut_generation=0)
    assert ext.fluid_cache_entries() == n0
    assert_close(got, orc.fluid_metric_apply(m, [0.1, 0.05, 0.01], True), torch.float32, "uncached")
    ext.fluid_cache_clear()
    assert ext.fluid_cache_entries() == 0


def test_interp_backward_rejects_bad_broadcast(ext):
    """interp_backward mirrors interp_forward's argument checks (1 < I.size(0) < u.size(0) is not a broadcast)."""
    I = torch.zeros((2, 1, 4, 4, 4), device="cuda")
    u = torch.zeros((3, 3, 4, 4, 4), device="cuda")
    go = torch.zeros((3, 1, 4, 4, 4), device="cuda")
    with pytest.raises(RuntimeError, match="batch sizes"):
        ext.interp_backward(go, I, u, 1.0, True, True)
    with pytest.raises(RuntimeError, match="inconsistent"):
        ext.interp_backward(go, I[:1], torch.zeros((3, 3, 4, 4, 5), device="cuda"), 1.0, True, True)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", [(6, 5, 8), (7, 9), (4, 3, 70), (12, 10, 40), (3, 4, 1)])
@pytest.mark.parametrize("dt", [1.0, -0.3])
def test_interp_backward_fused_start_values(ext, dtype, sp, dt):
    """lago_interp_backward_fused: d_I is that of interp_backward; d_u is the reference's thread-owned sum started
    from the caller's d_u (u_mode 1) or from addgo * grad_out (u_mode 2) instead of zero.  With a zero start it is the
    reference operator bit for bit; otherwise it equals `start + d_u` to rounding (one summation order differs)."""
    rng = np.random.default_rng(hash((sp, dt)) % 2**31)
    d = len(sp)
    u = _disp(rng, 2, sp, dtype)
    for nc, bc in ((d, False), (1, False), (2, True)):
        I = rnd(rng, (1 if bc else 2, nc) + sp, dtype)
        go = rnd(rng, (2, nc) + sp, dtype)
        oI, ou = orc.interp_backward(go, I, u, dt, True, True)
        start = rnd(rng, (2, d) + sp, dtype)
        # accumulate onto zeros == the reference operator, bit for bit
        dI0, du0 = ext.interp_backward_fused(dev(go), dev(I), dev(u), dt, True, d_u=torch.zeros_like(dev(u)))
        assert_bits(du0, ou, "fused, zero start")
        assert_close(dI0, oI, dtype, "fused d_I")
        dI1, du1 = ext.interp_backward_fused(dev(go), dev(I), dev(u), dt, True, d_u=dev(start))
        scale = np.abs(ou).max() + np.abs(start).max()
        assert_close(du1, start.astype(np.float64) + ou, dtype, "fused, accumulate", scale=scale)
        assert_close(dI1, oI, dtype, "fused d_I (accumulate)")
        if nc == d:
            _, du2 = ext.interp_backward_fused(dev(go), dev(I), dev(u), dt, False, addgo=0.37)
            k = go.dtype.type
            assert_close(du2, (k(0.37) * go).astype(np.float64) + ou, dtype, "fused, addgo",
                         scale=np.abs(ou).max() + np.abs(go).max())
    with pytest.raises(RuntimeError, match="exactly one"):
        ext.interp_backward_fused(dev(go), dev(I), dev(u), dt, True)
    with pytest.raises(RuntimeError, match="as many channels"):
        ext.interp_backward_fused(dev(rnd(rng, (2, d + 1) + sp, dtype)), dev(rnd(rng, (2, d + 1) + sp, dtype)), dev(u), dt, True, addgo=1.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", [(5, 6, 7), (7, 9), (4, 3, 64), (20, 12, 40)])
def test_ad_star_saves_the_resampled_momentum(ext, dtype, sp):
    """Ad_star(save_resampled=True) also returns interp_forward(m, phiinv), bit for bit, and the same product."""
    rng = np.random.default_rng(hash(sp) % 2**31)
    d = len(sp)
    phi = _disp(rng, 2, sp, dtype)
    m = rnd(rng, (2, d) + sp, dtype)
    for vec in (1, 0):
        ext.set_vector_kernels(vec)
        try:
            out, mphi = ext.Ad_star(dev(phi), dev(m), save_resampled=True)
            plain = ext.Ad_star(dev(phi), dev(m))
        finally:
            ext.set_vector_kernels(1)
        assert_bits(mphi, orc.interp_forward(m, phi, 1.0), "resampled momentum")
        assert torch.equal(out, plain)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", [(16, 16, 16), (20, 12, 40), (9, 11, 70), (5, 37, 128), (33, 6, 160), (7, 5, 250), (18, 17, 2), (3, 2, 256)])
def test_ad_star_row_tile_kernel(ext, dtype, sp):
    """The LDS row-tile form of Ad_star (csrc/stencil_tile.hpp: stencil neighbours from a staged tile with a
    one-voxel halo) against the oracle's interp + jacobian_times_vectorfield, bit for bit, and against the direct
    kernel: ragged last tiles in x and y, rows of 2 ... 250 voxels (1 to 4 z chunks), clamped faces on every side,
    far-out-of-range displacements, with and without the saved resampled momentum."""
    rng = np.random.default_rng(hash(sp) % 2**31)
    phi = _disp(rng, 3, sp, dtype)
    m = rnd(rng, (3, 3) + sp, dtype)
    want_m = orc.interp_forward(m, phi, 1.0)
    want = orc.jacobian_times_vectorfield_forward(phi, want_m, True, False)
    import lagomorph_amd.lagomorph_ext as shim

    got = {}
    for tile in (1, 0):
        ext.set_stencil_tile(tile)
        shim.set_gather_window(0)  # the LDS-window form would take the large float32 shapes
        try:
            got[tile] = ext.Ad_star(dev(phi), dev(m))
            out, mphi = ext.Ad_star(dev(phi), dev(m), save_resampled=True)
        finally:
            ext.set_stencil_tile(1)
            shim.set_gather_window(1)
        assert_bits(got[tile], want, f"Ad_star tile={tile} {sp}")
        assert_bits(out, want, f"Ad_star(save) tile={tile} {sp}")
        assert_bits(mphi, want_m, f"resampled momentum tile={tile} {sp}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("sp", [(6, 5, 8), (7, 9), (10, 12, 40)])
def test_fused_compose_backward_matches_unfused(ext, dtype, sp):
    """ComposeFunction.backward (one splat kernel whose d_u sum starts from ds * grad) == autograd through
    ds*u + dt*interp(v, u, ds)."""
    import lagomorph_amd as lm

    rng = np.random.default_rng(6)
    d = len(sp)
    u0, v0 = dev(0.8 * rnd(rng, (2, d) + sp, dtype)), dev(rnd(rng, (2, d) + sp, dtype))
    go = dev(rnd(rng, (2, d) + sp, dtype))
    for ds, dt in ((-0.2, 1.0), (1.0, 1.0), (0.5, -0.7)):
        a, b = u0.clone().requires_grad_(True), v0.clone().requires_grad_(True)
        lm.compose(a, b, ds=ds, dt=dt).backward(go)
        p, q = u0.clone().requires_grad_(True), v0.clone().requires_grad_(True)
        (ds * p + dt * lm.interp(q, p, dt=ds)).backward(go)
        assert_close(a.grad, host(p.grad), dtype, f"compose d_u ds={ds} dt={dt}")
        assert_close(b.grad, host(q.grad), dtype, f"compose d_v ds={ds} dt={dt}")
