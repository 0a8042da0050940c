"""Which kernel a call really runs.  Every fast path has a slower sibling that gives the same answer, so a parity test
cannot tell them apart; `lago_path_launches` (include/lagomorph_hip.h) can.  These cases pin the dispatch of the
BASELINE.json shapes and of the small shapes the parity tests use to exercise each path."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lm():
    import lagomorph_amd

    return lagomorph_amd


def _delta(shim, f):
    before = shim.path_launches()
    f()
    torch.cuda.synchronize()
    after = shim.path_launches()
    return {k: after[k] - before[k] for k in after if after[k] != before[k]}


def _fields(n, c, sp, dtype=torch.float32):
    g = torch.Generator(device="cuda").manual_seed(7)
    return torch.randn((n, c) + sp, device="cuda", dtype=dtype, generator=g)


@pytest.mark.parametrize("sp", [(128, 128, 128), (160, 160, 160), (32, 32, 32), (28, 32, 64)])
def test_headline_operators_take_their_fast_paths(lm, sp):
    """float32 3D at configs[1] / configs[4] extents (batch 1) and at the smallest shapes that qualify."""
    shim = lm.lagomorph_ext
    u, v = 0.5 * _fields(1, 3, sp), _fields(1, 3, sp)
    assert _delta(shim, lambda: shim.compose(u, v, 1.0, -0.1)) == {"gather_window": 1}
    # one 32^3 item is too small for the row tiles (make_row_tile wants full workgroups): it takes the direct kernel
    assert _delta(shim, lambda: shim.Ad_star(u, v)) == {"vector_gather" if sp == (32, 32, 32) else "stencil_tile": 1}
    I = _fields(1, 1, sp)
    assert _delta(shim, lambda: shim.interp_forward(I, u, 1.0)) == {"vector_gather": 1}     # one channel: pair gathers
    assert _delta(shim, lambda: shim.interp_forward(v, u, 1.0)) == {"gather_window": 1}     # several: through the LDS window
    go = _fields(1, 1, sp)
    assert _delta(shim, lambda: shim.interp_backward(go, I, u, 1.0, True, True)) == {"splat_shear": 1}
    go3 = _fields(1, 3, sp)
    assert _delta(shim, lambda: shim.interp_backward(go3, v, u, -0.1, True, True)) == {"splat_shear_mc": 1}


@pytest.mark.parametrize("sp,path", [((128, 128, 128), "fluid_lds"), ((160, 160, 160), "fluid_lds"), ((64, 96, 128), "fluid_lds"),
                                     ((64, 40, 40), "fluid_generic"), ((24, 20, 28), "fluid_generic"),
                                     ((128, 128), "fluid_2d"), ((64, 256), "fluid_2d"), ((100, 100), "fluid_generic")])
def test_fluid_metric_dispatch(lm, sp, path):
    metric = lm.FluidMetric([0.1, 0.0, 0.01])
    m = _fields(2, len(sp), sp)
    assert _delta(lm.lagomorph_ext, lambda: metric.sharp(m)) == {path: 1}
    assert _delta(lm.lagomorph_ext, lambda: metric.flat(m)) == {path: 1}
    # float64 never reaches rocFFT either (round 4: generic hand-written passes)
    assert _delta(lm.lagomorph_ext, lambda: metric.sharp(m.double())) == {"fluid_generic": 1}
    if path == "fluid_generic":   # rocFFT stays selectable: fluid_mode 2 and below
        lm.lagomorph_ext.set_fluid_mode(2)
        try:
            took = _delta(lm.lagomorph_ext, lambda: metric.sharp(m))
        finally:
            lm.lagomorph_ext.set_fluid_mode(3)
        assert list(took) in (["fluid_rocfft"], ["fluid_xpass"]), took


def test_extents_above_the_generic_passes_take_the_guarded_rocfft_path(lm):
    """csrc/fftg.hip holds a line of at most 4096 float32 / 2048 float64 points in LDS: a longer axis is the one shape
    class the default mode still hands to rocFFT (spot-checked).  Both sides of the limit, against the oracle."""
    import numpy as np
    from oracle import lago_oracle as orc

    metric = lm.FluidMetric([0.1, 0.0, 0.01])
    for sp, dtype, path in (((4096, 8), torch.float32, "fluid_generic"), ((8192, 4), torch.float32, "fluid_rocfft"),
                            ((2048, 6), torch.float64, "fluid_generic"), ((4096, 8), torch.float64, "fluid_rocfft"),
                            ((3000, 6), torch.float32, "fluid_generic"),
                            # a large prime factor runs as a Bluestein line of M >= 2 N - 1 points, which needs its own
                            # room in LDS: a prime above 2048 (float32) / 1024 (float64) does not fit and must not fall to
                            # the O(N r) direct stage silently -- it goes to the guarded rocFFT plan; just
                            # below, the Bluestein line serves
                            ((6, 2053), torch.float32, "fluid_rocfft"), ((6, 2039), torch.float32, "fluid_generic"),
                            ((4, 1031), torch.float64, "fluid_rocfft"), ((4, 1021), torch.float64, "fluid_generic")):
        m = _fields(1, 2, sp).to(dtype)
        took = _delta(lm.lagomorph_ext, lambda: metric.sharp(m))
        assert list(took) == [path], (sp, dtype, took)
        want = orc.fluid_metric_apply(m.cpu().numpy(), [0.1, 0.0, 0.01], True)
        got = metric.sharp(m).cpu().numpy()
        tol = 1e-5 if dtype == torch.float32 else 1e-11
        assert np.abs(got - want).max() <= tol * np.abs(want).max(), (sp, dtype)


def test_the_switches_select_the_slower_siblings(lm):
    shim = lm.lagomorph_ext
    sp = (32, 32, 32)
    u, v = 0.5 * _fields(1, 3, sp), _fields(1, 3, sp)
    try:
        shim.set_gather_window(0)
        assert _delta(shim, lambda: shim.compose(u, v, 1.0, -0.1)) == {"vector_gather": 1}
        shim.set_stencil_tile(0)
        assert _delta(shim, lambda: shim.Ad_star(u, v)) == {"vector_gather": 1}
        shim.set_vector_kernels(0)
        assert _delta(shim, lambda: shim.compose(u, v, 1.0, -0.1)) == {}
        assert _delta(shim, lambda: shim.interp_backward(v, v, u, -0.1, True, True)) == {"splat_tiled": 1}
        shim.set_splat_mode(0)
        assert _delta(shim, lambda: shim.interp_backward(v, v, u, -0.1, True, True)) == {"splat_global": 1}
    finally:
        shim.set_gather_window(1)
        shim.set_stencil_tile(1)
        shim.set_vector_kernels(1)
        shim.set_splat_mode(1)


def test_float64_and_2d_take_the_general_kernels(lm):
    shim = lm.lagomorph_ext
    u, v = 0.5 * _fields(1, 3, (32, 32, 32), torch.float64), _fields(1, 3, (32, 32, 32), torch.float64)
    assert _delta(shim, lambda: shim.compose(u, v, 1.0, -0.1)) == {"vector_gather": 1}
    assert _delta(shim, lambda: shim.interp_backward(v, v, u, -0.1, True, True)) == {"splat_tiled": 1}
    u2, v2 = 0.5 * _fields(1, 2, (64, 64)), _fields(1, 2, (64, 64))
    assert _delta(shim, lambda: shim.compose(u2, v2, 1.0, -0.1)) == {}
    assert _delta(shim, lambda: shim.interp_backward(v2, v2, u2, -0.1, True, True)) == {"splat_global": 1}   # 4096 pixels: too few
    u3, v3 = 0.5 * _fields(1, 2, (128, 128)), _fields(1, 2, (128, 128))
    assert _delta(shim, lambda: shim.interp_backward(v3, v3, u3, -0.1, True, True)) == {"splat_2d": 1}
    assert _delta(shim, lambda: shim.interp_backward(v3, v3, u3, -0.1, False, True)) == {"splat_global": 1}   # nothing to splat


def test_scatter_direction_is_history_free(lm):
    """VERDICT r4 item 6.  Launches alternate their block order (common.hpp: next_direction) from a process-wide
    counter; a scatter-add rounds its float atomics in arrival order, so until round 4 the block order of a splat -- and
    with it the last bits of d_I, d_A, d_T -- followed the PARITY of the number of library calls made before it, rejected
    ones included.  Scatter launches now always walk ascending and leave the counter alone.  Bit equality of two runs
    cannot show that: float atomics arrive in a different order on every run of the SAME launch (observed: 0 of 5
    repeats of one call reproduce its bits), so the mechanism itself is asserted through `lago_reversed_launches`:
    (1) no scatter-add entry point ever launches in descending order, whatever came before it (valid and rejected
    calls); (2) it does not advance the alternation either: order-independent kernels keep alternating strictly across
    any number of interleaved scatter calls; (3) with integer-valued data, where every sum is exact, the bits agree --
    a control that the calls compare like with like."""
    shim = lm.lagomorph_ext
    sp = (40, 36, 128)
    g = torch.Generator(device="cuda").manual_seed(11)
    u = 1.7 * torch.randn((2, 3) + sp, device="cuda", generator=g)
    sm = torch.nn.functional.avg_pool3d(u, 5, stride=1, padding=2)   # smooth enough for the LDS windows to hold it
    go = torch.randn((2, 1) + sp, device="cuda", generator=g)
    I = torch.randn((2, 1) + sp, device="cuda", generator=g)
    go3 = torch.randn((2, 3) + sp, device="cuda", generator=g)
    w = torch.randn((1, 3, 8, 8, 8), device="cuda", generator=g)
    A = (torch.eye(3, device="cuda").repeat(2, 1, 1) * 1.05).contiguous()
    T = torch.full((2, 3), 0.3, device="cuda")
    u2 = 0.5 * torch.randn((2, 2, 128, 128), device="cuda", generator=g)
    I2 = torch.randn((2, 1, 128, 128), device="cuda", generator=g)

    def unrelated(k):
        for _ in range(k):
            shim.jacobian_times_vectorfield_forward(w, w, True, False)   # one alternating launch each
        if k:
            with pytest.raises(RuntimeError):
                shim.interp_forward(w, torch.zeros((1, 3, 8, 8, 9), device="cuda"), 1.0)   # rejected after make_geom

    scatters = [
        lambda: shim.interp_backward(go, I, sm, 1.0, True, True),                 # sheared-window splat
        lambda: shim.interp_backward(go3, go3, sm, -0.3, True, True),             # its multi-channel form
        lambda: shim.interp_backward(go, I, 30.0 * u, 1.0, True, False),          # rough field: strays, global atomics
        lambda: shim.interp_backward(I2, I2, u2, 1.0, True, True),                # 2D LDS splat
        lambda: shim.affine_interp_backward(go, I, A, T, True, True, True),       # target boxes + the d_A / d_T reduction
        lambda: shim.regrid_backward(go3, [20, 18, 64], list(sp), [9.5, 8.5, 31.5], [19 / 39, 17 / 35, 63 / 127]),
        lambda: shim.interp_hessian_diagonal_image(I2, u2, 1.0),
    ]
    shim.REGRID_BACKWARD_SEPARABLE = 0    # (the separable form has no atomics; the splat form is the one at stake)
    try:
        for k in (0, 1, 2, 3):
            unrelated(k)
            before = shim.reversed_launches()
            for fn in scatters:
                fn()
            assert shim.reversed_launches() == before, f"a scatter-add launched in descending order after {k} unrelated calls"
        # (2) strict alternation of an order-independent kernel across interleaved scatter calls
        seen = []
        for i in range(6):
            before = shim.reversed_launches()
            shim.jacobian_times_vectorfield_forward(w, w, True, False)
            seen.append(shim.reversed_launches() - before)
            for fn in scatters[: i % 3]:
                fn()
        assert seen in ([0, 1, 0, 1, 0, 1], [1, 0, 1, 0, 1, 0]), seen
    finally:
        shim.REGRID_BACKWARD_SEPARABLE = 1
    # (3) exact arithmetic: bits cannot depend on anything
    ui = torch.round(sm)
    goi = torch.round(4 * go)
    ref_bits = shim.interp_backward(goi, I, ui, 1.0, True, False)[0]
    for k in (1, 2, 3):
        unrelated(k)
        assert torch.equal(shim.interp_backward(goi, I, ui, 1.0, True, False)[0], ref_bits)
