"""CPU: the strict (unfused) oracle build is bit-identical to oracle/_ref, the reference's own
cpu/affine.cpp compiled from /root/reference where it lies (BASELINE configs[0] included), and to the
reference's own biLerp / triLerp / biLerp_grad / triLerp_grad (include/interp.h, compiled into the same
module) at random points -- negative, far out of range, exactly integer, exactly on the border, 2D and
3D, float32 and float64.  This pins the floor rule, clamp, lerp and gradient expression order of the
oracle's interpolation core (rows a1, a2 d_u, a9, a10 of SURVEY section 8) value by value to real reference
code.  Skipped when oracle/_ref has not been built (it needs /root/reference)."""
import numpy as np
import pytest
import torch

from oracle import build_ref
from oracle import lago_oracle as orc

try:  # (re)build from /root/reference when it is there (a fresh checkout has no binaries); no-op otherwise
    build_ref.build()
except Exception as e:  # pragma: no cover - toolchain trouble must not break collection
    print(f"[tests] oracle/_ref not built: {e}")
ref = build_ref.load_ref() if __import__("os").path.exists(build_ref.built_path()) else None
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(autouse=True)
def strict_oracle():
    orc.set_strict(True)
    yield
    orc.set_strict(False)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("N,C,bc", [(2, 1, False), (3, 2, True), (1, 4, False)])
def test_cpuref_restatement_bit_exact(dtype, dim, N, C, bc):
    rng = np.random.default_rng(dim * 100 + N * 10 + C)
    sh = (7, 9) if dim == 2 else (5, 6, 7)
    I = rng.standard_normal(((1 if bc else N), C) + sh).astype(dtype)
    A = (np.eye(dim)[None] + 0.3 * rng.standard_normal((N, dim, dim))).astype(dtype)
    T = (2.0 * rng.standard_normal((N, dim))).astype(dtype)
    want = ref.affine_interp_cpu_forward(torch.from_numpy(I), torch.from_numpy(A), torch.from_numpy(T)).numpy()
    got = orc.affine_interp_forward(I, A, T, cpuref=True)
    assert np.array_equal(got, want)


def test_config0_2d_affine_batch2_64x64():
    """BASELINE.json configs[0]: 2D affine_interp forward, batch 2, 1x64x64 random images."""
    rng = np.random.default_rng(1)
    I = rng.standard_normal((2, 1, 64, 64)).astype(np.float32)
    A = (np.eye(2)[None] + 0.1 * rng.standard_normal((2, 2, 2))).astype(np.float32)
    T = rng.standard_normal((2, 2)).astype(np.float32)
    want = ref.affine_interp_cpu_forward(torch.from_numpy(I), torch.from_numpy(A), torch.from_numpy(T)).numpy()
    assert np.array_equal(orc.affine_interp_forward(I, A, T, cpuref=True), want)
    # the CUDA-path restatement (per-voxel positions) agrees with the CPU path only to rounding,
    # exactly like the reference's own test_affine_interp_gpucpu_match (allclose)
    assert np.allclose(orc.affine_interp_forward(I, A, T), want, atol=2e-3)


def test_identity_transform_is_exact():
    """testing/test_affine.py:30-40 on the real reference build and on the oracle."""
    rng = np.random.default_rng(2)
    I = rng.standard_normal((2, 3, 6, 5, 4))
    A = np.tile(np.eye(3), (2, 1, 1))
    T = np.zeros((2, 3))
    want = ref.affine_interp_cpu_forward(torch.from_numpy(I), torch.from_numpy(A), torch.from_numpy(T)).numpy()
    assert np.array_equal(want, I) and np.array_equal(orc.affine_interp_forward(I, A, T), I)


def _points(rng, sh, n, dtype):
    d = len(sh)
    p = rng.uniform(-3.0, np.array(sh) + 2.0, size=(n, d))
    p[::7] = np.round(p[::7])                       # exactly integer: weights exactly 0 / 1, gradient one-sided
    p[::11] = rng.uniform(-40.0, 40.0, size=p[::11].shape)  # far outside: clamp on both corners
    p[::13, 0] = sh[0] - 1                          # exactly on the upper border
    p[::17] = -np.abs(p[::17]) - 0.25               # negative non-integers: the floor rule (int)x - 1
    p[5] = 0.0
    p[6] = -0.0
    return p.astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("sh", [(7, 9), (2, 2), (5, 1), (5, 6, 7), (2, 2, 2), (9, 5, 1), (3, 4, 33)])
def test_interp_cores_bit_exact_against_reference_header(dtype, sh):
    """include/interp.h:9-122 (biLerp, triLerp) and :128-327 (biLerp_grad, triLerp_grad), CLAMP background."""
    rng = np.random.default_rng(len(sh) * 1000 + sum(sh))
    img = rng.standard_normal(sh).astype(dtype)
    pts = _points(rng, sh, 4000, dtype)
    lerp_ref, grad_ref = (t.numpy() for t in ref.interp_points(torch.from_numpy(img), torch.from_numpy(pts)))
    lerp, grad = orc.interp_points(img, pts)
    assert np.array_equal(lerp, lerp_ref), np.abs(lerp - lerp_ref).max()
    assert np.array_equal(grad, grad_ref[:, 1:]), np.abs(grad - grad_ref[:, 1:]).max()
    # the value the *_grad functions return alongside (Ix) is the same interpolant up to association order
    assert np.allclose(grad_ref[:, 0], lerp_ref, rtol=0, atol=(1e-5 if dtype == np.float32 else 1e-13) * np.abs(img).max())


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_interp_forward_through_reference_cores(dtype):
    """interp_forward of the oracle at dt = 1 equals the reference's triLerp evaluated at i + u (positions formed
    as the kernel forms them, cuda/interp.cu:68-70: double sum narrowed to Real)."""
    rng = np.random.default_rng(5)
    sh = (6, 7, 8)
    I = rng.standard_normal((1, 1) + sh).astype(dtype)
    u = (3.0 * rng.standard_normal((1, 3) + sh)).astype(dtype)
    ii = np.stack(np.meshgrid(*[np.arange(s) for s in sh], indexing="ij")).astype(dtype)
    pos = (ii.astype(np.float64) + 1.0 * u[0].astype(np.float64)).astype(dtype).reshape(3, -1).T.copy()
    want = ref.interp_points(torch.from_numpy(I[0, 0].copy()), torch.from_numpy(pos))[0].numpy().reshape(sh)
    assert np.array_equal(orc.interp_forward(I, u, 1.0)[0, 0], want)


# ---- include/extrap.h: the index rules -------------------------------------------------------------

def _indices(rng, sh, n):
    d = len(sh)
    idx = np.stack([rng.integers(-4, s + 4, size=n) for s in sh], axis=1)
    idx[::5] = np.stack([rng.integers(0, s, size=len(idx[::5])) for s in sh], axis=1)   # inside the grid
    idx[::7, 0] = -1                                   # just below / above every border
    idx[1::7, 0] = sh[0]
    idx[2::7, -1] = sh[-1] - 1
    idx[3::7, -1] = 0
    idx[4::11] = rng.integers(-10 ** 6, 10 ** 6, size=idx[4::11].shape)   # far out of range
    return idx.astype(np.int64)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("sh", [(7, 9), (2, 2), (5, 1), (1, 6), (2, 9), (5, 6, 7), (2, 2, 2), (9, 5, 1), (1, 1, 4), (3, 2, 33)])
def test_clamped_accessor_and_central_differences_against_reference_header(dtype, sh):
    """include/extrap.h:110-192 get_value_safe<CLAMP> -- the accessor every diff_x / diff_y / diff_z of include/diff.h:7-52
    calls -- and the central differences around it, on in-range, border, out-of-range, extent-1 and extent-2 indices: the
    oracle's lg_clamp and lg_grad_point (rows a4-a7: every jtv routine differentiates through them) bit for bit against
    real reference code."""
    rng = np.random.default_rng(7 * len(sh) + sum(sh))
    arr = rng.standard_normal(sh).astype(dtype)
    idx = _indices(rng, sh, 3000)
    val_ref, grad_ref = (t.numpy() for t in ref.extrap_points(torch.from_numpy(arr), torch.from_numpy(idx)))
    val, grad = orc.extrap_points(arr, idx)
    assert np.array_equal(val, val_ref)
    assert np.array_equal(grad, grad_ref)
    # and the accessor is what its name says: the value at the clamped index
    cl = tuple(np.clip(idx[:, a], 0, sh[a] - 1) for a in range(len(sh)))
    assert np.array_equal(val_ref, arr[cl])
    # along an axis of extent 1 the clamped difference is identically zero (the adjoint kernels' thin-axis case)
    for a, s in enumerate(sh):
        if s == 1:
            assert not grad_ref[:, a].any()


@pytest.mark.parametrize("sizes", [(7, 9), (2, 2), (5, 1), (5, 6, 7), (2, 2, 2), (9, 1, 4), (3, 4, 33)])
def test_clamp_background_and_map_point_against_reference_header(sizes):
    """include/extrap.h:46-77 clampBackground, :194-253 map_point<CLAMP>, :24-38 isInside on (floor, ceil = floor + 1)
    pairs and on arbitrary pairs: the oracle's lg_clamp_pair (interpolation footprints of rows a1, a9, a10; the clamped
    target cells of the splats, rows a2, a9, a10) against real reference code, index by index."""
    rng = np.random.default_rng(sum(sizes))
    d = len(sizes)
    fl = np.stack([rng.integers(-5, s + 5, size=4000) for s in sizes], axis=1)
    ce = fl + 1
    ce[::9] = np.stack([rng.integers(-5, s + 5, size=len(ce[::9])) for s in sizes], axis=1)   # (not a footprint: any pair)
    fl[1::13] = rng.integers(-10 ** 6, 10 ** 6, size=fl[1::13].shape)
    ce[1::13] = fl[1::13] + 1
    fc = np.concatenate([fl, ce], axis=1).astype(np.int64)
    out = ref.map_points(torch.from_numpy(fc), list(sizes)).numpy()
    assert out[:, 2 * d].all(), "map_point<CLAMP> always reports the mapped point as usable (extrap.h:212-217)"
    for a, s in enumerate(sizes):
        f, c = orc.clamp_pairs(fl[:, a], ce[:, a], s)
        assert np.array_equal(f, out[:, a]) and np.array_equal(c, out[:, d + a]), (a, s)
        # for a footprint pair the rule is the plain clamp of both members (what the HIP kernels compute: common.hpp clamp1)
        foot = ce[:, a] == fl[:, a] + 1
        assert np.array_equal(out[foot, a], np.clip(fl[foot, a], 0, s - 1))
        assert np.array_equal(out[foot, d + a], np.clip(ce[foot, a], 0, s - 1))
    inside = np.ones(len(fc), dtype=bool)
    for a, s in enumerate(sizes):
        inside &= (fl[:, a] >= 0) & (ce[:, a] < s)
    assert np.array_equal(out[:, 2 * d + 1].astype(bool), inside)
