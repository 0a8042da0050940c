"""GPU: volumes at the library's size limit (one channel plane of one batch item below 2^29 voxels: gathers address a
plane with 32-bit byte offsets) and (y, z) planes above the 24-bit-multiply range of the fast
index arithmetic -- the "maximum sizes" edge of the hot path.  The CPU oracle would need minutes per case at half a
billion voxels, so the checks are exact PROPERTIES instead: with an integer-valued displacement every sample lands on a
grid point, interp_forward is a clamped index shift (bit for bit), its adjoint is the matching index_add of an
integer-valued field (every sum exact, whatever the order of the atomics), and d_u is the forward difference of the
image at the sample times grad_out (one rounding, the kernel's own).  Sizes that do not fit are rejected, not wrapped."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ext():
    import lagomorph_amd as lm

    if torch.cuda.get_device_properties(0).total_memory < 64 * 2**30:
        pytest.skip("needs 64 GB of device memory")
    return lm.lagomorph_ext


def _shifted(I, shift):
    """I[clamp(i + a), clamp(j + b), clamp(k + c)] for (1, 1, nx, ny, nz)."""
    out = I
    for d, s in zip((2, 3, 4), shift):
        n = I.size(d)
        out = out.index_select(d, (torch.arange(n, device=I.device) + s).clamp_(0, n - 1))
    return out


# (nx, ny, nz): 530.8 M voxels, just below 2^29 = 536.9 M; and a (y, z) plane of 21.2 MB >= 2^24 B, where the
# 24-bit-multiply index arithmetic (common.hpp: Lerp3::setup) and the sheared-window splat (splat.hip: make_shear) step aside
@pytest.mark.parametrize("sp,shift", [((768, 768, 900), (3, -2, 5)), ((96, 2304, 2304), (-1, 4, -7))])
def test_interp_forward_and_adjoint_at_the_size_limit(ext, sp, shift):
    g = torch.Generator(device="cuda").manual_seed(5)
    I = torch.randn((1, 1) + sp, device="cuda", generator=g)
    u = torch.empty((1, 3) + sp, device="cuda")
    for c in range(3):
        u[:, c] = float(shift[c])
    out = ext.interp_forward(I, u, 1.0)
    want = _shifted(I, shift)
    assert torch.equal(out, want), "interp_forward at the size limit is not the clamped index shift"
    del out
    go = torch.round(4.0 * torch.randn((1, 1) + sp, device="cuda", generator=g))
    d_I, d_u = ext.interp_backward(go, I, u, 1.0, True, True)
    # adjoint of the clamped shift: index_add along each axis (integer-valued: exact in any order)
    acc = go
    for d, s in zip((2, 3, 4), shift):
        n = sp[d - 2]
        idx = (torch.arange(n, device="cuda") + s).clamp_(0, n - 1)
        acc = torch.zeros_like(acc).index_add_(d, idx, acc)
    assert torch.equal(d_I, acc), "splat at the size limit is not the adjoint of the clamped shift"
    del acc, d_I
    # d_u: forward difference of I at the (integer) sample, clamped like the corners, times grad_out (include/interp.h:315-326
    # with zero fractions)
    for c in range(3):
        up = list(shift)
        up[c] += 1
        gc = _shifted(I, up) - want
        if shift[c] + 1 > 0:   # samples whose +1 corner is clamped onto the same cell: zero difference -- already so
            pass
        assert torch.equal(d_u[:, c:c + 1], gc * go), f"d_u[{c}] at the size limit"
        del gc


@pytest.mark.parametrize("sp,shift", [((768, 768, 900), (2, -3, 4)), ((96, 2304, 2304), (1, -5, 6))])
def test_compose_and_ad_star_at_the_size_limit(ext, sp, shift):
    """Three-channel fields of 6.4 GB each: with a CONSTANT integer displacement u, compose(u, v, 1, 1) = u + v(x + u) and
    Ad_star(u, m) = m(x + u) (the Jacobian of a constant field vanishes exactly: (c - c) / 2 = 0, 0 * m + m = m) -- both
    bit for bit; the fused kernels' 160 / 128-row fast paths do not apply to these rows, the general ones run."""
    g = torch.Generator(device="cuda").manual_seed(6)
    v = torch.randn((1, 3) + sp, device="cuda", generator=g)
    u = torch.empty((1, 3) + sp, device="cuda")
    for c in range(3):
        u[:, c] = float(shift[c])
    want = torch.cat([_shifted(v[:, c:c + 1], shift) for c in range(3)], dim=1)
    got = ext.Ad_star(u, v)
    assert torch.equal(got, want), "Ad_star at the size limit"
    del got
    got = ext.compose(u, v, 1.0, 1.0)
    want += u
    assert torch.equal(got, want), "compose at the size limit"


def test_planes_of_2_29_voxels_are_rejected(ext):
    I = torch.empty((1, 1, 1024, 1024, 512), device="cuda")   # exactly 2^29 voxels
    u = torch.empty((1, 3, 1024, 1024, 512), device="cuda")
    with pytest.raises(RuntimeError, match="bad extent"):
        ext.interp_forward(I, u, 1.0)
    with pytest.raises(RuntimeError, match="bad extent"):
        ext.jacobian_times_vectorfield_forward(u, u, True, False)


def test_fluid_metric_round_trip_at_512_cubed():
    """FluidMetric on a 512^3 field (134 M voxels, 1.6 GB per three-component field: the generic FFT passes, 512-point
    lines): flat(sharp(m)) returns m (testing/test_metric.py: test_fluid_inverse, at the volume's scale)."""
    import lagomorph_amd as lm

    if torch.cuda.get_device_properties(0).total_memory < 64 * 2**30:
        pytest.skip("needs 64 GB of device memory")
    g = torch.Generator(device="cuda").manual_seed(9)
    m = torch.randn((1, 3, 512, 512, 512), device="cuda", generator=g)
    met = lm.FluidMetric([0.1, 0.05, 0.01])
    back = met.flat(met.sharp(m))
    err = float((back - m).abs().max() / m.abs().max())
    assert err <= 1e-4, err   # the operator's condition number at gamma = 0.01 is 1e4 x float32 rounding (the reference's own test: atol 1e-2... rtol 1e-3)
