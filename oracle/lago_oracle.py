"""TEST INFRASTRUCTURE ONLY -- ctypes front end of the CPU oracle.

Loads ``oracle/_build/liblago_oracle.so`` (built by ``make -C oracle`` /
``__graft_entry__.build()``) and exposes the reference's 12 compute entry
points (``/root/reference/lagomorph/extension/extension.cpp:175-189``) over
numpy arrays, plus :class:`OracleExt`, the same surface over CPU torch tensors
so that tests can stand it in for ``lagomorph_ext`` when they exercise the host
mirror (``lagomorph_amd``) without a GPU.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import this module.  The product package never does.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LAGO_ORACLE_BUILD: another build of the same sources (oracle/Makefile `asan`: AddressSanitizer + UBSan, `make -C oracle
# check-asan` runs the CPU suite on it)
_BUILD = os.environ.get("LAGO_ORACLE_BUILD") or os.path.join(_HERE, "_build")
_LIB_PATHS = {
    False: os.path.join(_BUILD, "liblago_oracle.so"),        # documented FMA contraction (matches HIP)
    True: os.path.join(_BUILD, "liblago_oracle_strict.so"),  # unfused a*b+c (matches oracle/_ref)
}
_libs = {}
_strict = False

c_long = ctypes.c_long
c_int = ctypes.c_int
c_double = ctypes.c_double
c_void_p = ctypes.c_void_p


def set_strict(flag):
    """Select the unfused build (True) or the FMA-contracted build (False, default)."""
    global _strict
    _strict = bool(flag)


def set_threads(n):
    """OpenMP threads for the forward kernels expmap uses (interp, jacobian-times-vectorfield, fluid
    operator); every voxel is independent there, so results do not depend on it.  Default 1."""
    lib().oracle_set_threads(int(n))


def lib():
    if _strict not in _libs:
        path = _LIB_PATHS[_strict]
        if not os.path.exists(path):
            raise RuntimeError(f"oracle library missing: {path}; run `make -C oracle` or __graft_entry__.build()")
        L = ctypes.CDLL(path)
        L.oracle_version.restype = ctypes.c_char_p
        _libs[_strict] = L
    return _libs[_strict]


def _suf(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f32"
    if dtype == np.float64:
        return "_f64"
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _p(a):
    return c_void_p(a.ctypes.data) if a is not None else c_void_p(None)


def _c(a, dtype=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def _call(name, dtype, *args):
    fn = getattr(lib(), name + _suf(dtype))
    fn.restype = c_int
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f"oracle {name}{_suf(dtype)} returned {rc}")


def _sp(shape):
    """spatial dims -> (dim, nx, ny, nz)"""
    d = len(shape)
    if d == 2:
        return 2, shape[0], shape[1], 1
    if d == 3:
        return 3, shape[0], shape[1], shape[2]
    raise RuntimeError("Only two- and three-dimensional inputs are supported")


# --------------------------------------------------------------------------- numpy surface


def interp_forward(I, u, dt=1.0):
    I, u = _c(I), _c(u, I.dtype)
    dim, nx, ny, nz = _sp(I.shape[2:])
    nn = max(I.shape[0], u.shape[0])
    bc = int(I.shape[0] < nn)
    out = np.empty((nn, I.shape[1]) + I.shape[2:], dtype=I.dtype)
    _call("oracle_interp_forward", I.dtype, _p(out), _p(I), _p(u), c_double(dt), c_int(dim), c_long(nn),
          c_long(I.shape[1]), c_long(nx), c_long(ny), c_long(nz), c_int(bc))
    return out


def interp_backward(go, I, u, dt, need_I, need_u):
    I = _c(I)
    go, u = _c(go, I.dtype), _c(u, I.dtype)
    dim, nx, ny, nz = _sp(I.shape[2:])
    nn = max(I.shape[0], u.shape[0])
    bc = int(I.shape[0] < nn)
    d_I = np.empty_like(I)
    d_u = np.empty_like(u)
    _call("oracle_interp_backward", I.dtype, _p(d_I), _p(d_u), _p(go), _p(I), _p(u), c_double(dt), c_int(dim),
          c_long(nn), c_long(I.shape[1]), c_long(nx), c_long(ny), c_long(nz), c_int(bc), c_int(bool(need_I)),
          c_int(bool(need_u)))
    return d_I, d_u


def interp_hessian_diagonal_image(I, u, dt):
    I = _c(I)
    u = _c(u, I.dtype)
    nn = max(I.shape[0], u.shape[0])
    out = np.empty_like(I)
    _call("oracle_interp_hessian_diagonal_image", I.dtype, _p(out), _p(u), c_double(dt), c_long(I.shape[0]),
          c_long(nn), c_long(I.shape[1]), c_long(I.shape[2]), c_long(I.shape[3]))
    return out


def jacobian_times_vectorfield_forward(g, v, displacement, transpose):
    g = _c(g)
    v = _c(v, g.dtype)
    dim, nx, ny, nz = _sp(g.shape[2:])
    out = np.empty_like(g)
    _call("oracle_jtv_forward", g.dtype, _p(out), _p(g), _p(v), c_int(bool(displacement)), c_int(bool(transpose)),
          c_int(dim), c_long(g.shape[0]), c_long(g.shape[1]), c_long(nx), c_long(ny), c_long(nz))
    return out


def jacobian_times_vectorfield_backward(go, v, w, displacement, transpose, need_v=True, need_w=True):
    v = _c(v)
    go, w = _c(go, v.dtype), _c(w, v.dtype)
    dim, nx, ny, nz = _sp(v.shape[2:])
    d_v, d_w = np.empty_like(v), np.empty_like(w)
    _call("oracle_jtv_backward", v.dtype, _p(d_v), _p(d_w), _p(go), _p(v), _p(w), c_int(bool(displacement)),
          c_int(bool(transpose)), c_int(dim), c_long(v.shape[0]), c_long(v.shape[1]), c_long(nx), c_long(ny),
          c_long(nz))
    return d_v, d_w


def jacobian_times_vectorfield_adjoint_forward(g, v):
    g = _c(g)
    v = _c(v, g.dtype)
    dim, nx, ny, nz = _sp(g.shape[2:])
    out = np.empty_like(g)
    _call("oracle_jtv_adjoint_forward", g.dtype, _p(out), _p(g), _p(v), c_int(dim), c_long(g.shape[0]),
          c_long(g.shape[1]), c_long(nx), c_long(ny), c_long(nz))
    return out


def jacobian_times_vectorfield_adjoint_backward(go, v, w, need_v=True, need_w=True):
    v = _c(v)
    go, w = _c(go, v.dtype), _c(w, v.dtype)
    dim, nx, ny, nz = _sp(v.shape[2:])
    d_v, d_w = np.empty_like(v), np.empty_like(w)
    _call("oracle_jtv_adjoint_backward", v.dtype, _p(d_v), _p(d_w), _p(go), _p(v), _p(w), c_int(dim),
          c_long(v.shape[0]), c_long(nx), c_long(ny), c_long(nz))
    return d_v, d_w


def fluid_operator(Fmv, inverse, cosluts, sinluts, alpha, beta, gamma):
    """In place on Fmv of shape (N, d, nx, ny[, nzc], 2), like the reference."""
    assert Fmv.flags["C_CONTIGUOUS"]
    dim = Fmv.ndim - 3
    luts = []
    for d in range(3):
        if d < dim:
            luts += [_c(cosluts[d], Fmv.dtype), _c(sinluts[d], Fmv.dtype)]
        else:
            luts += [None, None]
    sh = list(Fmv.shape[2:2 + dim]) + [1] * (3 - dim)
    _call("oracle_fluid_operator", Fmv.dtype, _p(Fmv), c_int(bool(inverse)), *[_p(a) for a in luts],
          c_double(alpha), c_double(beta), c_double(gamma), c_int(dim), c_long(Fmv.shape[0]), c_long(sh[0]),
          c_long(sh[1]), c_long(sh[2]))
    return None


def interp_points(img, pts):
    """bilinear / trilinear value and gradient of one image (sx, sy[, sz]) at points (npts, dim)."""
    img = _c(img)
    pts = _c(pts, img.dtype)
    dim = img.ndim
    sh = list(img.shape) + [1] * (3 - dim)
    lerp = np.empty((pts.shape[0],), dtype=img.dtype)
    grad = np.empty((pts.shape[0], dim), dtype=img.dtype)
    _call("oracle_interp_points", img.dtype, _p(lerp), _p(grad), _p(img), _p(pts), c_long(pts.shape[0]), c_int(dim),
          c_long(sh[0]), c_long(sh[1]), c_long(sh[2]))
    return lerp, grad


def extrap_points(arr, idx):
    """The clamped accessor and the clamped central differences (lg_clamp / lg_grad_point) of one array (nx, ny[, nz]) at
    integer indices (npts, dim), in range or not.  Test hook (tests/test_oracle_ref.py)."""
    arr = _c(arr)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    dim = arr.ndim
    sh = list(arr.shape) + [1] * (3 - dim)
    val = np.empty((idx.shape[0],), dtype=arr.dtype)
    grad = np.empty((idx.shape[0], dim), dtype=arr.dtype)
    _call("oracle_extrap_points", arr.dtype, _p(val), _p(grad), _p(arr), _p(idx), c_long(idx.shape[0]), c_int(dim),
          c_long(sh[0]), c_long(sh[1]), c_long(sh[2]))
    return val, grad


def clamp_pairs(fl, ce, size):
    """clampBackground on (floor, ceil) index pairs (lg_clamp_pair).  Test hook."""
    fl = np.ascontiguousarray(fl, dtype=np.int64).copy()
    ce = np.ascontiguousarray(ce, dtype=np.int64).copy()
    _call("oracle_clamp_pairs", np.dtype(np.float64), _p(fl), _p(ce), c_long(int(size)), c_long(fl.shape[0]))
    return fl, ce


def affine_interp_forward(I, A, T, cpuref=False):
    I = _c(I)
    A, T = _c(A, I.dtype), _c(T, I.dtype)
    dim, nx, ny, nz = _sp(I.shape[2:])
    nn = A.shape[0]
    bc = int(I.shape[0] == 1 and nn > 1)
    out = np.empty((nn, I.shape[1]) + I.shape[2:], dtype=I.dtype)
    name = "oracle_affine_interp_forward_cpuref" if cpuref else "oracle_affine_interp_forward"
    _call(name, I.dtype, _p(out), _p(I), _p(A), _p(T), c_int(dim), c_long(nn), c_long(I.shape[1]), c_long(nx),
          c_long(ny), c_long(nz), c_int(bc))
    return out


def affine_interp_backward(go, I, A, T, need_I, need_A, need_T):
    I = _c(I)
    go, A, T = _c(go, I.dtype), _c(A, I.dtype), _c(T, I.dtype)
    dim, nx, ny, nz = _sp(I.shape[2:])
    nn = go.shape[0]
    bc = int(I.shape[0] == 1 and nn > 1)
    d_I = np.empty_like(I) if need_I else np.zeros((0,), I.dtype)
    d_A = np.empty_like(A) if need_A else np.zeros((0,), I.dtype)
    d_T = np.empty_like(T) if need_T else np.zeros((0,), I.dtype)
    _call("oracle_affine_interp_backward", I.dtype, _p(d_I if need_I else None), _p(d_A if need_A else None),
          _p(d_T if need_T else None), _p(go), _p(I), _p(A), _p(T), c_int(dim), c_long(nn), c_long(go.shape[1]),
          c_long(nx), c_long(ny), c_long(nz), c_int(bc), c_int(bool(need_I)), c_int(bool(need_A)),
          c_int(bool(need_T)))
    return d_I, d_A, d_T


def _dv(x, dim):
    a = (ctypes.c_double * 3)(*([float(v) for v in x] + [0.0] * (3 - dim)))
    return a


def regrid_forward(I, shape, origin, spacing):
    I = _c(I)
    dim, nx, ny, nz = _sp(I.shape[2:])
    shape = [int(s) for s in shape]
    N = shape + [1] * (3 - dim)
    out = np.empty(I.shape[:2] + tuple(shape), dtype=I.dtype)
    _call("oracle_regrid_forward", I.dtype, _p(out), _p(I), c_int(dim), c_long(I.shape[0]), c_long(I.shape[1]),
          c_long(nx), c_long(ny), c_long(nz), c_long(N[0]), c_long(N[1]), c_long(N[2]), _dv(origin, dim),
          _dv(spacing, dim))
    return out


def regrid_backward(go, inshape, shape, origin, spacing):
    go = _c(go)
    inshape = [int(s) for s in inshape]
    dim, nx, ny, nz = _sp(tuple(inshape))
    shape = [int(s) for s in shape]
    N = shape + [1] * (3 - dim)
    d_I = np.empty(go.shape[:2] + tuple(inshape), dtype=go.dtype)
    _call("oracle_regrid_backward", go.dtype, _p(d_I), _p(go), c_int(dim), c_long(go.shape[0]),
          c_long(go.shape[1]), c_long(nx), c_long(ny), c_long(nz), c_long(N[0]), c_long(N[1]), c_long(N[2]),
          _dv(origin, dim), _dv(spacing, dim))
    return d_I


# --------------------------------------------------------------------------- FluidMetric (metric.py:37-97)


def fluid_luts(shape, dtype):
    """cos/sin LUTs exactly as metric.py:53-75 builds them: float64 numpy,
    rounded through float32 (torch.Tensor(...)), then cast to `dtype`.  `shape`
    is the spatial shape; the last axis uses N//2+1 entries."""
    cos, sin = [], []
    for d, N in enumerate(shape):
        Nf = N // 2 + 1 if d == len(shape) - 1 else N
        k = np.arange(Nf)
        cos.append((2.0 * (1.0 - np.cos(2 * np.pi * k / N))).astype(np.float32).astype(dtype))
        sin.append(np.sin(2.0 * np.pi * k / N).astype(np.float32).astype(dtype))
    return cos, sin


def fluid_metric_apply(m, params, inverse):
    """FluidMetricOperator.forward (metric.py:11-19): ortho rFFT over the
    spatial axes, per-frequency operator, inverse rFFT."""
    m = _c(m)
    dim = m.ndim - 2
    axes = tuple(range(2, 2 + dim))
    F = np.fft.rfftn(m, axes=axes, norm="ortho").astype(np.complex64 if m.dtype == np.float32 else np.complex128)
    Fm = np.ascontiguousarray(F).view(m.dtype).reshape(F.shape + (2,))
    cos, sin = fluid_luts(m.shape[2:], m.dtype)
    fluid_operator(Fm, inverse, cos, sin, *params)
    F2 = Fm.reshape(F.shape[:-1] + (F.shape[-1] * 2,)).view(F.dtype)
    return np.fft.irfftn(F2, s=m.shape[2:], axes=axes, norm="ortho").astype(m.dtype)


# --------------------------------------------------------------------------- torch-tensor surface


class OracleExt:
    """The ``lagomorph_ext`` surface (extension.cpp:175-189) over CPU torch
    tensors, backed by the oracle.  Tests monkeypatch it in place of the HIP
    shim to run the host mirror's compositions on a box without a GPU."""

    def __init__(self):
        import torch

        self._t = torch

    def _n(self, t):
        return t.detach().cpu().contiguous().numpy()

    def _w(self, a, like):
        return self._t.from_numpy(np.ascontiguousarray(a)).to(like.device)

    def set_debug_mode(self, mode):
        return None

    def interp_forward(self, I, u, dt=1.0):
        return self._w(interp_forward(self._n(I), self._n(u), dt), I)

    def compose(self, u, v, ds=1.0, dt=1.0):
        # deform.py:53-55 with torch's rounding: scalars rounded to the tensor dtype, three roundings
        un, vn = self._n(u), self._n(v)
        k = un.dtype.type
        return self._w(k(ds) * un + k(dt) * interp_forward(vn, un, ds), u)

    def interp_backward(self, go, I, u, dt, need_I, need_u):
        a, b = interp_backward(self._n(go), self._n(I), self._n(u), dt, need_I, need_u)
        return [self._w(a, I), self._w(b, I)]

    def interp_hessian_diagonal_image(self, I, u, dt):
        return self._w(interp_hessian_diagonal_image(self._n(I), self._n(u), dt), I)

    def jacobian_times_vectorfield_forward(self, g, v, displacement, transpose):
        return self._w(jacobian_times_vectorfield_forward(self._n(g), self._n(v), displacement, transpose), g)

    def jacobian_times_vectorfield_backward(self, go, v, w, displacement, transpose, need_v, need_w):
        a, b = jacobian_times_vectorfield_backward(self._n(go), self._n(v), self._n(w), displacement, transpose)
        return [self._w(a, v), self._w(b, v)]

    def jacobian_times_vectorfield_adjoint_forward(self, g, v):
        return self._w(jacobian_times_vectorfield_adjoint_forward(self._n(g), self._n(v)), g)

    def jacobian_times_vectorfield_adjoint_backward(self, go, v, w, need_v, need_w):
        a, b = jacobian_times_vectorfield_adjoint_backward(self._n(go), self._n(v), self._n(w))
        return [self._w(a, v), self._w(b, v)]

    def fluid_operator(self, Fmv, inverse, cosluts, sinluts, alpha, beta, gamma):
        a = self._n(Fmv).copy()
        fluid_operator(a, inverse, [self._n(c) for c in cosluts], [self._n(s) for s in sinluts], alpha, beta, gamma)
        Fmv.copy_(self._t.from_numpy(a))
        return None

    def affine_interp_forward(self, I, A, T):
        return self._w(affine_interp_forward(self._n(I), self._n(A), self._n(T)), I)

    def affine_interp_backward(self, go, I, A, T, need_I, need_A, need_T):
        r = affine_interp_backward(self._n(go), self._n(I), self._n(A), self._n(T), need_I, need_A, need_T)
        return [self._w(x, I) for x in r]

    def regrid_forward(self, I, shape, origin, spacing):
        return self._w(regrid_forward(self._n(I), shape, origin, spacing), I)

    def regrid_backward(self, go, inshape, shape, origin, spacing):
        return self._w(regrid_backward(self._n(go), inshape, shape, origin, spacing), go)
