"""The reference's own test-suite (/root/reference/testing/test_*.py), restated against this
build's host mirror.  Every test runs twice:

  * backend "oracle" (CPU, not gpu): the CPU oracle stands in for lagomorph_ext -- this pins the
    oracle (and the autograd wrappers / compositions) to the properties the reference tests pin:
    float64 gradcheck, adjoint identities, 2D == 3D, identity transforms, inverse round trips;
  * backend "hip" (marked gpu): the same assertions through the HIP kernels on cuda.

Sizes, dtypes, parameter grids and tolerances are the reference's (res 2-3, float64; expmap at
a reduced resolution on the CPU backend so the suite stays fast).
"""
import pytest
import torch

TF = [True, False]
BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def lm(request):
    import lagomorph_amd

    if request.param == "oracle":
        request.getfixturevalue("oracle_ext")
        lagomorph_amd._test_device = "cpu"
    else:
        lagomorph_amd.set_debug_mode(True)
        lagomorph_amd._test_device = "cuda"
    torch.manual_seed(1)
    return lagomorph_amd


def T(lm, shape, requires_grad=False):
    return torch.randn(shape, dtype=torch.float64, device=lm._test_device, requires_grad=requires_grad)


def gradcheck(fn, args, eps=1e-6):
    assert torch.autograd.gradcheck(fn, args, eps=eps)


# ---- testing/test_interp.py -------------------------------------------------------------


@pytest.mark.parametrize("nc", [1, 2, 4])
@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("testI,testu", [(True, True), (True, False), (False, True)])
@pytest.mark.parametrize("broadcastI", TF)
def test_interp_gradcheck(lm, bs, nc, dim, testI, testu, broadcastI):
    res = 2
    imsh = tuple([1 if broadcastI else bs, nc] + [res] * dim)
    I = T(lm, imsh, testI)
    u = T(lm, tuple([bs, dim] + [res] * dim), testu)
    gradcheck(lm.interp, (I, u))


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("nc", [1, 2, 4])
@pytest.mark.parametrize("broadcastI", TF)
def test_interp_2d_match_3d(lm, bs, nc, broadcastI):
    res = 2
    I = T(lm, tuple([1 if broadcastI else bs, nc] + [res] * 2))
    u = T(lm, tuple([bs, 2] + [res] * 2))
    u3 = torch.zeros(tuple([bs, 3] + [res] * 2 + [1]), dtype=u.dtype, device=u.device)
    u3[:, :2, ...] = u.unsqueeze(4)
    assert torch.allclose(lm.interp(I, u).unsqueeze(4), lm.interp(I.unsqueeze(4), u3))


# ---- testing/test_diff.py ---------------------------------------------------------------


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("disp", TF)
@pytest.mark.parametrize("trans", TF)
@pytest.mark.parametrize("testphi,testm", [(True, True), (True, False), (False, True)])
def test_jacobian_times_vectorfield_gradcheck(lm, bs, dim, disp, trans, testphi, testm):
    sh = tuple([bs, dim] + [2] * dim)
    phiinv, m = T(lm, sh, testphi), T(lm, sh, testm)
    gradcheck(lambda v, w: lm.jacobian_times_vectorfield(v, w, displacement=disp, transpose=trans), (phiinv, m))


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("disp", TF)
def test_jacobian_times_vectorfield_transpose(lm, bs, dim, disp):
    sh = tuple([bs, dim] + [2] * dim)
    g, u, v = T(lm, sh), T(lm, sh), T(lm, sh)
    Dguv = (lm.jacobian_times_vectorfield(g, u, displacement=disp, transpose=False) * v).sum()
    uDgTv = (u * lm.jacobian_times_vectorfield(g, v, displacement=disp, transpose=True)).sum()
    assert torch.allclose(Dguv, uDgTv)


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("dim", [2, 3])
def test_jacobian_times_vectorfield_adjoint(lm, bs, dim):
    sh = tuple([bs, dim] + [2] * dim)
    u, v, m = T(lm, sh), T(lm, sh), T(lm, sh)
    Duvm = (lm.jacobian_times_vectorfield(u, v, displacement=False, transpose=False) * m).sum()
    uadjvm = (u * lm.jacobian_times_vectorfield_adjoint(m, v)).sum()
    assert torch.allclose(Duvm, uadjvm)


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("dim", [2, 3])
def test_jacobian_times_vectorfield_adjoint_gradcheck(lm, bs, dim):
    sh = tuple([bs, dim] + [2] * dim)
    gradcheck(lm.jacobian_times_vectorfield_adjoint, (T(lm, sh, True), T(lm, sh, True)))


def _lift(x2):
    x3 = torch.zeros(tuple(x2.shape[:1]) + (3,) + tuple(x2.shape[2:]) + (2,), dtype=x2.dtype, device=x2.device)
    x3[:, :2, :, :, 0] = x2
    x3[:, :2, :, :, 1] = x2
    return x3


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("disp", TF)
@pytest.mark.parametrize("trans", TF)
def test_jacobian_times_vectorfield_2d_match_3d(lm, bs, disp, trans):
    v2, m2 = T(lm, (bs, 2, 2, 2)), T(lm, (bs, 2, 2, 2))
    d2 = lm.jacobian_times_vectorfield(v2, m2, displacement=disp, transpose=trans)
    d3 = lm.jacobian_times_vectorfield(_lift(v2), _lift(m2), displacement=disp, transpose=trans)
    assert torch.allclose(d3[:, :2, :, :, 0], d2)


@pytest.mark.parametrize("bs", [1, 2])
def test_jacobian_times_vectorfield_adjoint_2d_match_3d(lm, bs):
    v2, m2 = T(lm, (bs, 2, 2, 2)), T(lm, (bs, 2, 2, 2))
    d2 = lm.jacobian_times_vectorfield_adjoint(v2, m2)
    d3 = lm.jacobian_times_vectorfield_adjoint(_lift(v2), _lift(m2))
    assert torch.allclose(d3[:, :2, :, :, 0], d2)


# ---- testing/test_metric.py -------------------------------------------------------------


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("dim", [2, 3])
def test_fluid_sharp_gradcheck(lm, bs, dim):
    m = T(lm, tuple([bs, dim] + [3] * dim), True)
    gradcheck(lm.FluidMetric([0.1, 0.01, 0.001]).sharp, (m,), eps=1e-4)


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("dim", [2, 3])
def test_fluid_flat_gradcheck(lm, bs, dim):
    v = T(lm, tuple([bs, dim] + [3] * dim), True)
    gradcheck(lm.FluidMetric([0.1, 0.01, 0.001]).flat, (v,))


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("dim", [2, 3])
def test_fluid_inverse(lm, bs, dim):
    m = T(lm, tuple([bs, dim] + [3] * dim))
    metric = lm.FluidMetric([0.1, 0.01, 0.001])
    assert torch.allclose(metric.flat(metric.sharp(m)), m, atol=1e-3)


# ---- testing/test_adjrep.py, test_lddmm.py ----------------------------------------------


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("dim", [2, 3])
def test_Ad_star_gradcheck(lm, bs, dim):
    sh = tuple([bs, dim] + [2] * dim)
    gradcheck(lm.Ad_star, (T(lm, sh, True), T(lm, sh)))


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("step", [1, 5])
def test_expmap_zero(lm, bs, dim, step):
    # the reference uses res = 128 (test_lddmm.py:18); the CPU oracle backend runs 3D at 24^3
    res = 128 if (lm._test_device == "cuda" or dim == 2) else 24
    if lm._test_device == "cpu" and dim == 2:
        res = 64
    m = torch.zeros(tuple([bs, dim] + [res] * dim), dtype=torch.float64, device=lm._test_device)
    h = lm.expmap(lm.FluidMetric([1.0, 0.1, 0.01]), m, num_steps=step)
    assert torch.allclose(m, h)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("masked", TF)
def test_expmap_first_step_closed_form(lm, dim, dtype, masked):
    """expmap without a phiinv evaluates the Euler step from the identity as -dt sharp(m0) (lddmm._first_step).  That
    must be what the reference's loop returns when it is run step by step from an explicit zero displacement
    (lddmm.py:39-44, :99-104) through the Ad_star and compose kernels: the same displacement bit for bit, and the same
    gradient with respect to the momentum."""
    from lagomorph_amd import lddmm

    sp = (9, 10, 11)[:dim] if dim == 3 else (17, 12)
    met = lm.FluidMetric([0.1, 0.01, 0.01])
    m0 = torch.randn((2, dim) + sp, dtype=torch.float64, device=lm._test_device)
    m0 = (m0 * (2.0 / met.sharp(m0).abs().max())).to(dtype)
    mask = (torch.rand((2, 1) + sp, device=lm._test_device) > 0.3).to(dtype) if masked else None
    go = torch.randn((2, dim) + sp, dtype=torch.float64, device=lm._test_device).to(dtype)
    steps = 3
    a = m0.clone().requires_grad_(True)
    h = lm.expmap(met, a, num_steps=steps, mommask=mask)
    h.backward(go)
    b = m0.clone().requires_grad_(True)
    phi = torch.zeros_like(m0)
    for _ in range(steps):
        phi = lddmm.EPDiff_step(met, b, 1.0 / steps, phi, mommask=mask)
    phi.backward(go)
    assert torch.equal(h, phi)
    scale = float(b.grad.abs().max())
    assert float((a.grad - b.grad).abs().max()) <= (1e-13 if dtype == torch.float64 else 1e-6) * scale
    # ... and sharp(m0) handed in by the caller (lddmm_step shares it with its regulariser) changes nothing but the
    # order of two sums
    if not masked:
        c = m0.clone().requires_grad_(True)
        v0 = met.sharp(c)
        h2 = lm.expmap(met, c, num_steps=steps, v0=v0)
        assert torch.equal(h2, h)
        (h2 * go).sum().add((v0 * c).sum()).backward()
        d = m0.clone().requires_grad_(True)
        (lm.expmap(met, d, num_steps=steps) * go).sum().add((met.sharp(d) * d).sum()).backward()
        scale = float(d.grad.abs().max())
        assert float((c.grad - d.grad).abs().max()) <= (1e-12 if dtype == torch.float64 else 1e-5) * scale


# ---- testing/test_affine.py -------------------------------------------------------------


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("c", [1, 2, 4])
def test_affine_interp_identity(lm, bs, dim, c):
    I = T(lm, tuple([bs, c] + [2] * dim))
    A = torch.eye(dim, dtype=I.dtype, device=I.device).repeat(bs, 1, 1)
    Tt = torch.zeros((bs, dim), dtype=I.dtype, device=I.device)
    assert torch.allclose(lm.affine_interp(I, A, Tt), I)


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("c", [1, 2, 4])
@pytest.mark.parametrize("testI,testA,testT", [(True, True, True), (True, False, False), (False, True, False),
                                               (False, False, True), (False, True, True)])
def test_affine_interp_gradcheck(lm, bs, dim, c, testI, testA, testT):
    I = T(lm, tuple([bs, c] + [2] * dim), testI)
    A = T(lm, (bs, dim, dim), testA)
    Tt = T(lm, (bs, dim), testT)
    gradcheck(lm.affine_interp, (I, A, Tt))


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("c", [1, 2, 4])
def test_affine_2d_match_3d(lm, bs, c):
    with torch.no_grad():
        I2 = T(lm, (bs, c, 2, 2))
        A2, T2 = T(lm, (bs, 2, 2)), T(lm, (bs, 2))
        A3 = torch.zeros((bs, 3, 3), dtype=A2.dtype, device=A2.device)
        A3[:, :2, :2] = A2
        A3[:, 2, 2] = 1
        T3 = torch.cat((T2, torch.zeros((bs, 1), dtype=T2.dtype, device=T2.device)), dim=1)
        J2 = lm.affine_interp(I2, A2, T2).view(bs, c, 2, 2, 1)
        J3 = lm.affine_interp(I2.view(bs, c, 2, 2, 1), A3, T3)
        assert torch.allclose(J2, J3)


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("dim", [2, 3])
def test_affine_inverse(lm, bs, dim):
    A, Tt = torch.randn((bs, dim, dim), dtype=torch.float64), torch.randn((bs, dim), dtype=torch.float64)
    x = torch.randn((bs, dim, 1), dtype=torch.float64)
    Ainv, Tinv = lm.affine_inverse(A, Tt)
    y = torch.matmul(A, x) + Tt.unsqueeze(2)
    assert torch.allclose(x, torch.matmul(Ainv, y) + Tinv.unsqueeze(2))


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("disp", TF)
def test_regrid_identity(lm, bs, dim, disp):
    I = T(lm, tuple([bs, dim] + [2] * dim), True)
    assert torch.allclose(I, lm.regrid(I, shape=I.shape[2:], displacement=disp))


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("c", [1, 2, 4])
def test_regrid_gradcheck(lm, bs, dim, c):
    I = T(lm, tuple([bs, c] + [2] * dim), True)
    gradcheck(lambda J: lm.regrid(J, shape=[3] * dim, displacement=False), (I,))


@pytest.mark.parametrize("bs", [1, 2])
@pytest.mark.parametrize("dim", [2, 3])
def test_regrid_displacement_gradcheck(lm, bs, dim):
    I = T(lm, tuple([bs, dim] + [2] * dim), True)
    gradcheck(lambda J: lm.regrid(J, shape=[3] * dim, displacement=True), (I,))
