"""CPU: the strict (unfused) oracle build is bit-identical to oracle/_ref, the reference's own
cpu/affine.cpp compiled from /root/reference where it lies (BASELINE configs[0] included).  This
pins the floor rule, clamp and lerp expression order of the oracle's interpolation core to real
reference code.  Skipped when oracle/_ref has not been built (it needs /root/reference)."""
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
