"""Second, independent statement of the finite-difference and fluid-metric rows
of the oracle, kept in the repo so that a regression in
``oracle/lago_oracle_impl.h`` for those rows is caught without the reference.

The oracle follows the reference's kernels line by line (cuda/diff.cu,
cuda/metric.cu): per-voxel loops, a three-case border formula for every
adjoint, a hand-unrolled Cholesky per frequency.  Here the same operators are
written the way a textbook would:

* the clamped central difference along one axis is ONE sparse matrix ``D_d``
  (built from its definition 0.5*(f[clamp(i+1)] - f[clamp(i-1)])); every
  adjoint is the matrix transpose ``D_d.T`` -- no border cases at all;
* the fluid operator is ``L(k) = l(k) @ l(k)`` with
  ``l = lambda*I - beta*diag(w) + beta*(s s^T - diag(s*s))`` applied with
  ``numpy.linalg`` (``solve`` for the inverse), over the FULL complex spectrum
  (``fftn``), not the half spectrum the reference and the oracle use.

float64 agrees to rounding (1e-12); float32 to a few ulp of the result's scale.
CPU only; the oracle is the thing under test here, not the checker.
"""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle.lago_oracle as O

SHAPES = [(7, 5), (4, 9), (1, 6), (6, 1), (5, 4, 6), (3, 7, 2), (1, 5, 4), (4, 4, 1)]
DTYPES = [np.float64, np.float32]


def _tol(dtype):
    return 2e-13 if dtype == np.float64 else 3e-6


def _d1(n):
    """Clamped central difference on n points as a sparse matrix."""
    i = np.arange(n)
    ip, im = np.minimum(i + 1, n - 1), np.maximum(i - 1, 0)
    return (sp.coo_matrix((np.full(n, 0.5), (i, ip)), shape=(n, n))
            - sp.coo_matrix((np.full(n, 0.5), (i, im)), shape=(n, n))).tocsr()


def _diffs(shape):
    """[D_0 .. D_{dim-1}] acting on C-order flattened fields."""
    mats = []
    for d in range(len(shape)):
        m = sp.identity(1, format="csr")
        for e, n in enumerate(shape):
            m = sp.kron(m, _d1(n) if e == d else sp.identity(n, format="csr"), format="csr")
        mats.append(m)
    return mats


def _jac(D, v, displacement):
    """J[n, c, d] = D_d v_c (+ delta_cd) as flat fields, float64."""
    nn, nc = v.shape[:2]
    dim = len(D)
    J = np.empty((nn, nc, dim, v[0, 0].size))
    for n in range(nn):
        for c in range(nc):
            f = v[n, c].reshape(-1).astype(np.float64)
            for d in range(dim):
                J[n, c, d] = D[d] @ f
                if displacement and c == d:
                    J[n, c, d] += 1.0
    return J


def _flat(a):
    return a.reshape(a.shape[0], a.shape[1], -1).astype(np.float64)


def _DT(D, p):
    """sum over the trailing axis pairing: p[n, c, d, :] -> sum_d D_d^T p[n, c, d]."""
    out = np.zeros(p.shape[:2] + p.shape[3:])
    for n in range(p.shape[0]):
        for c in range(p.shape[1]):
            for d in range(len(D)):
                out[n, c] += D[d].T @ p[n, c, d]
    return out


def _close(got, want, dtype, scale=None):
    want = np.asarray(want).reshape(got.shape)
    s = max(1.0, float(np.abs(want).max())) if scale is None else scale
    err = float(np.abs(got.astype(np.float64) - want).max()) / s
    assert err <= _tol(dtype), err


def _fields(rng, nn, nc, shape, dtype):
    return rng.standard_normal((nn, nc) + shape).astype(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("displacement", [False, True])
@pytest.mark.parametrize("transpose", [False, True])
def test_jtv_forward_and_backward_against_sparse_operators(shape, dtype, displacement, transpose):
    rng = np.random.default_rng(hash((shape, displacement, transpose)) & 0xffff)
    dim, nn = len(shape), 2
    D = _diffs(shape)
    v = _fields(rng, nn, dim, shape, dtype)
    w = _fields(rng, nn, dim, shape, dtype)
    go = _fields(rng, nn, dim, shape, dtype)
    J = _jac(D, v, displacement)
    wf, gf = _flat(w), _flat(go)

    out = O.jacobian_times_vectorfield_forward(v, w, displacement, transpose)
    want = np.einsum("ncdx,ncx->ndx", J, wf) if transpose else np.einsum("ncdx,ndx->ncx", J, wf)
    _close(out, want, dtype)

    d_v, d_w = O.jacobian_times_vectorfield_backward(go, v, w, displacement, transpose)
    if transpose:
        want_w = np.einsum("ncdx,ndx->ncx", J, gf)
        want_v = _DT(D, np.einsum("ncx,ndx->ncdx", wf, gf))
    else:
        want_w = np.einsum("ncdx,ncx->ndx", J, gf)
        want_v = _DT(D, np.einsum("ndx,ncx->ncdx", wf, gf))
    _close(d_w, want_w, dtype)
    _close(d_v, want_v, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(7, 5), (5, 4, 6)])
@pytest.mark.parametrize("nc", [1, 4])
def test_jtv_forward_any_channel_count(shape, dtype, nc):
    """displacement=False, transpose=False takes any channel count (an image
    gradient contracted with a field: diff.cu:129-185)."""
    rng = np.random.default_rng(nc)
    dim = len(shape)
    D = _diffs(shape)
    v = _fields(rng, 2, nc, shape, dtype)
    w = _fields(rng, 2, dim, shape, dtype)
    out = O.jacobian_times_vectorfield_forward(v, w, False, False)
    _close(out, np.einsum("ncdx,ndx->ncx", _jac(D, v, False), _flat(w)), dtype)
    go = _fields(rng, 2, nc, shape, dtype)
    d_v, d_w = O.jacobian_times_vectorfield_backward(go, v, w, False, False)
    _close(d_w, np.einsum("ncdx,ncx->ndx", _jac(D, v, False), _flat(go)), dtype)
    _close(d_v, _DT(D, np.einsum("ndx,ncx->ncdx", _flat(w), _flat(go))), dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", SHAPES)
def test_jtv_adjoint_forward_and_backward_against_sparse_operators(shape, dtype):
    rng = np.random.default_rng(len(shape) * 100 + shape[0])
    dim, nn = len(shape), 2
    D = _diffs(shape)
    z = _fields(rng, nn, dim, shape, dtype)
    w = _fields(rng, nn, dim, shape, dtype)
    go = _fields(rng, nn, dim, shape, dtype)
    zf, wf = _flat(z), _flat(w)

    out = O.jacobian_times_vectorfield_adjoint_forward(z, w)
    _close(out, _DT(D, np.einsum("ndx,ncx->ncdx", wf, zf)), dtype)

    # out_c = sum_d D_d^T (w_d z_c)  =>  d_z_c = sum_d w_d (D_d go_c),  d_w_d = sum_c z_c (D_d go_c)
    G = _jac(D, go, False)
    d_z, d_w = O.jacobian_times_vectorfield_adjoint_backward(go, z, w)
    _close(d_z, np.einsum("ncdx,ndx->ncx", G, wf), dtype)
    _close(d_w, np.einsum("ncdx,ncx->ndx", G, zf), dtype)


@pytest.mark.parametrize("shape", [(7, 5), (5, 4, 6)])
def test_adjoint_really_is_the_adjoint(shape):
    """<J_v w, z> == <v, adjoint(z, w)> for the plain (non-displacement) form --
    ties the two oracle rows to each other with no helper of this file."""
    rng = np.random.default_rng(3)
    dim = len(shape)
    v, w, z = (_fields(rng, 2, dim, shape, np.float64) for _ in range(3))
    lhs = float((O.jacobian_times_vectorfield_forward(v, w, False, False) * z).sum())
    rhs = float((v * O.jacobian_times_vectorfield_adjoint_forward(z, w)).sum())
    assert abs(lhs - rhs) <= 1e-12 * max(1.0, abs(lhs))


# --------------------------------------------------------------------------- fluid metric


def _full_luts(shape):
    """metric.py:53-75 over the FULL spectrum (every axis N entries), float32
    rounded like torch.Tensor(...) does, returned as float64."""
    cos, sin = [], []
    for N in shape:
        k = np.arange(N)
        cos.append(np.float32(2.0 * (1.0 - np.cos(2 * np.pi * k / N))).astype(np.float64))
        sin.append(np.float32(np.sin(2.0 * np.pi * k / N)).astype(np.float64))
    return cos, sin


def _L_of_k(shape, params):
    alpha, beta, gamma = params
    dim = len(shape)
    cos, sin = _full_luts(shape)
    W = np.stack(np.meshgrid(*cos, indexing="ij"), axis=-1)  # (..., dim)
    S = np.stack(np.meshgrid(*sin, indexing="ij"), axis=-1)
    lam = gamma + alpha * W.sum(-1)
    eye = np.eye(dim)
    l = beta * S[..., :, None] * S[..., None, :] * (1.0 - eye)
    l = l + (lam[..., None] - beta * W)[..., :, None] * eye
    return l @ l


def _apply_full_spectrum(m, params, inverse):
    dim = m.ndim - 2
    axes = tuple(range(2, 2 + dim))
    F = np.fft.fftn(m.astype(np.float64), axes=axes, norm="ortho")
    F = np.moveaxis(F, 1, -1)[..., None]  # (n, *shape, dim, 1)
    L = _L_of_k(m.shape[2:], params)[None]
    R = np.linalg.solve(np.broadcast_to(L, F.shape[:-2] + L.shape[-2:]), F) if inverse else L @ F
    R = np.moveaxis(R[..., 0], -1, 1)
    back = np.fft.ifftn(R, axes=axes, norm="ortho")
    assert float(np.abs(back.imag).max()) <= 1e-9 * max(1.0, float(np.abs(back.real).max()))
    return back.real


FLUID_SHAPES = [(8, 6), (5, 7), (16, 4), (6, 4, 8), (5, 3, 7), (4, 4, 5)]


@pytest.mark.parametrize("shape", FLUID_SHAPES)
@pytest.mark.parametrize("inverse", [False, True])
def test_fluid_metric_f64_against_full_spectrum_linear_algebra(shape, inverse):
    """The reference's default parameters (metric.py:37-44): condition number
    of L is ~1e6, so the inverse is compared at 1e-9 of its scale."""
    rng = np.random.default_rng(sum(shape))
    m = rng.standard_normal((2, len(shape)) + shape)
    params = (0.1, 0.01, 0.001)
    got = O.fluid_metric_apply(m, params, inverse)
    want = _apply_full_spectrum(m, params, inverse)
    err = float(np.abs(got - want).max()) / float(np.abs(want).max())
    assert err <= (1e-9 if inverse else 1e-13), err


@pytest.mark.parametrize("shape", FLUID_SHAPES)
@pytest.mark.parametrize("inverse", [False, True])
def test_fluid_metric_f32_against_full_spectrum_linear_algebra(shape, inverse):
    """float32 with a well-conditioned L (cond ~ 4) so that the comparison
    measures the arithmetic, not the conditioning."""
    rng = np.random.default_rng(sum(shape) + 1)
    m = rng.standard_normal((2, len(shape)) + shape).astype(np.float32)
    params = (0.05, 0.02, 1.0)
    got = O.fluid_metric_apply(m, params, inverse)
    assert got.dtype == np.float32
    want = _apply_full_spectrum(m, params, inverse)
    err = float(np.abs(got - want).max()) / float(np.abs(want).max())
    assert err <= 5e-6, err


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(8, 6), (6, 4, 8)])
@pytest.mark.parametrize("inverse", [False, True])
def test_fluid_operator_per_frequency(shape, dtype, inverse):
    """The operator row on its own (metric.cu:162-355): arbitrary complex
    input on the half spectrum, every bin against numpy.linalg."""
    rng = np.random.default_rng(11)
    dim = len(shape)
    half = shape[:-1] + (shape[-1] // 2 + 1,)
    params = (0.05, 0.02, 1.0)
    Fm = rng.standard_normal((2, dim) + half + (2,)).astype(dtype)
    cos, sin = O.fluid_luts(shape, dtype)
    ref = Fm.astype(np.float64)
    work = Fm.copy()
    O.fluid_operator(work, inverse, cos, sin, *params)
    sl = tuple(slice(0, h) for h in half)
    L = _L_of_k(shape, params)[sl][None]
    b = np.moveaxis(ref[..., 0] + 1j * ref[..., 1], 1, -1)[..., None]
    R = np.linalg.solve(np.broadcast_to(L, b.shape[:-2] + L.shape[-2:]), b) if inverse else L @ b
    R = np.moveaxis(R[..., 0], -1, 1)
    want = np.stack([R.real, R.imag], axis=-1)
    err = float(np.abs(work - want).max()) / float(np.abs(want).max())
    assert err <= (1e-13 if dtype == np.float64 else 3e-6), err


@pytest.mark.parametrize("shape", [(8, 6), (6, 4, 8)])
def test_sharp_undoes_flat(shape):
    rng = np.random.default_rng(5)
    m = rng.standard_normal((2, len(shape)) + shape)
    params = (0.1, 0.01, 0.001)
    back = O.fluid_metric_apply(O.fluid_metric_apply(m, params, False), params, True)
    assert float(np.abs(back - m).max()) <= 1e-9
