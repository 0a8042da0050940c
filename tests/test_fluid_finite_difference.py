"""BASELINE configs[2] names "FFT vs finite-difference solver": `FluidMetric.flat` (Fourier-domain symbol, cuda/metric.cu:
236-254) against the same operator applied in the spatial domain by periodic finite-difference stencils (tests/fd_fluid.py,
an independent derivation from the symbol), and `sharp` as its inverse.  CPU: the oracle backend, 2D and 3D, float64.  GPU:
the HIP kernels at the configs[2] size (batch 8 x 3 x 128^3) and on the other FFT paths.

The reference rounds its cos / sin tables through float32 even for float64 fields (metric.py:66-75, replicated), so the
Fourier operator differs from the exact stencils by that rounding: 1e-7 of the symbol.  With exact tables injected into
the metric the two agree to float64 rounding."""
import numpy as np
import pytest
import torch

from fd_fluid import flat_fd


def _exact_luts(lm, spatial_shape, dtype, device):
    """fluid_luts without the reference's float32 rounding."""
    cos, sin = [], []
    nd = len(spatial_shape)
    for d, N in enumerate(spatial_shape):
        Nf = N // 2 + 1 if d == nd - 1 else N
        k = np.arange(Nf)
        cos.append(torch.from_numpy(2.0 * (1.0 - np.cos(2 * np.pi * k / N))).to(dtype).to(device))
        sin.append(torch.from_numpy(np.sin(2.0 * np.pi * k / N)).to(dtype).to(device))
    return {"cos": cos, "sin": sin, "gen": next(lm.metric._LUT_GENERATION)}


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


PARAMS = [[0.1, 0.0, 0.01], [0.1, 0.05, 0.01], [1.0, 0.3, 0.2]]


@pytest.mark.parametrize("params", PARAMS)
@pytest.mark.parametrize("sp", [(12, 10), (7, 9), (8, 6, 10), (5, 7, 6)])
def test_flat_equals_finite_difference_stencils_cpu(oracle_ext, monkeypatch, sp, params):
    import lagomorph_amd as lm

    g = torch.Generator().manual_seed(sum(sp))
    v = torch.randn((2, len(sp)) + sp, generator=g, dtype=torch.float64)
    want = flat_fd(v, params)
    got = lm.FluidMetric(params).flat(v)
    assert _rel(got, want) <= 2e-7            # the float32-rounded tables of the reference
    monkeypatch.setattr(lm.metric, "fluid_luts", lambda s, dt, dev: _exact_luts(lm, s, dt, dev))
    met = lm.FluidMetric(params)
    got = met.flat(v)
    assert _rel(got, want) <= 1e-12, "exact tables: the Fourier operator IS the stencil operator"
    # and sharp inverts it
    back = met.sharp(want)
    assert _rel(back, v) <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("params", PARAMS[:2])
@pytest.mark.parametrize("sp,B,dtype,tol", [((128, 128, 128), 8, torch.float32, 1e-5),    # configs[2]: the tuned LDS passes
                                            ((64, 48, 80), 2, torch.float64, 2e-7),      # generic passes, float64
                                            ((33, 29, 31), 2, torch.float64, 2e-7),      # odd extents (Bluestein / direct stages)
                                            ((96, 80), 3, torch.float32, 1e-5),          # fused 2D kernel
                                            ((160, 160, 160), 2, torch.float32, 1e-5)])
def test_flat_equals_finite_difference_stencils_hip(sp, B, dtype, tol, params):
    import lagomorph_amd as lm

    g = torch.Generator(device="cuda").manual_seed(sum(sp))
    v = torch.randn((B, len(sp)) + sp, device="cuda", generator=g, dtype=dtype)
    met = lm.FluidMetric(params)
    with torch.no_grad():
        got = met.flat(v)
        want = flat_fd(v.double(), params)
        e = _rel(got, want)
        back = _rel(met.sharp(got), v)
    print(f"flat vs finite differences {sp} x{B} {dtype} {params}: {e:.2e}; sharp(flat(v)) vs v: {back:.2e}")
    assert e <= tol
    assert back <= (1e-4 if dtype == torch.float32 else 1e-9)   # (the inverse amplifies by the operator's condition number)


@pytest.mark.gpu
def test_flat_float64_exact_tables_hip(monkeypatch):
    import lagomorph_amd as lm

    monkeypatch.setattr(lm.metric, "fluid_luts", lambda s, dt, dev: _exact_luts(lm, s, dt, dev))
    sp = (40, 36, 44)
    g = torch.Generator(device="cuda").manual_seed(4)
    v = torch.randn((2, 3) + sp, device="cuda", generator=g, dtype=torch.float64)
    with torch.no_grad():
        assert _rel(lm.FluidMetric([0.1, 0.05, 0.01]).flat(v), flat_fd(v, [0.1, 0.05, 0.01])) <= 1e-12
