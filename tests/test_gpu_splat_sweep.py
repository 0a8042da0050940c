"""GPU: the shipped float32 displacement splat (`splat_shear_kernel`, csrc/splat.hip) and its general tiled
sibling under EVERY tuning setting, against the CPU oracle (VERDICT r2 "parity gap", ADVICE r2 coverage gap):

  * `set_splat_shear` sweep -- tiles, margins 0/1/2, 256/512/1024 threads, multi-channel form on/off -- on a smooth
    and a rough displacement at (40, 36, 96), one and three channels, unit and non-unit step: d_u bit-exact, d_I within
    north_star's 1e-5 x max;
  * the float32 general tiled kernel forced (`set_splat_shear(on=0)`), incl. its multi-channel form;
  * `interp_backward_fused` in the combination the expmap reverse sweep uses (need_I with addgo; d_I accumulation)
    at shapes that reach the sheared kernel, shear on/off, splat modes 0/1, broadcast or not.
"""
import numpy as np
import pytest
import torch

from oracle import lago_oracle as orc
from test_gpu_parity import assert_bits, assert_close, dev, host, rnd, _disp

pytestmark = pytest.mark.gpu

DEFAULT_SHEAR = (1, 8, 6, 0, 1, 1, 4, 1024)


@pytest.fixture(scope="module")
def ext():
    import lagomorph_amd

    lagomorph_amd.set_debug_mode(True)
    e = lagomorph_amd.lagomorph_ext
    yield e
    e.set_splat_shear(*DEFAULT_SHEAR)
    e.set_splat_shear_mc(2)
    e.set_splat_mc(1)
    e.set_splat_mode(1)
    e.set_splat_tile(0, 8, 0, 1, 1, 4, 512)


def smooth_np(rng, shape, sigma, amp):
    """Smooth random field (separable Gaussian blur in numpy), max |.| = amp."""
    from scipy.ndimage import gaussian_filter

    a = rng.standard_normal(shape)
    a = gaussian_filter(a, sigma=(0, 0) + (sigma,) * (len(shape) - 2), mode="wrap")
    return (a * (amp / np.abs(a).max())).astype(np.float32)


SP = (40, 36, 96)
_cache = {}


def fields(kind, nc):
    key = (kind, nc)
    if key not in _cache:
        rng = np.random.default_rng(17 + nc + (0 if kind == "smooth" else 100))
        u = smooth_np(rng, (2, 3) + SP, 6.0, 4.0) if kind == "smooth" else (2.0 * rng.standard_normal((2, 3) + SP)).astype(np.float32)
        if kind == "rough":
            u.reshape(-1)[::97] *= 30.0   # some far outside the grid
        I = smooth_np(rng, (2, nc) + SP, 2.0, 1.0)
        go = rng.standard_normal((2, nc) + SP).astype(np.float32)
        want = {dt: orc.interp_backward(go, I, u, dt, True, True) for dt in (1.0, -0.2)}
        _cache[key] = (u, I, go, want)
    return _cache[key]


SHEAR_CFGS = [
    # on tx ty tz mx my mz threads
    (1, 8, 6, 0, 1, 1, 4, 1024),     # the product's default
    (1, 8, 6, 0, 0, 0, 0, 1024),     # no margin at all: many footprints leave the window
    (1, 4, 8, 0, 2, 2, 8, 1024),
    (1, 4, 4, 32, 1, 1, 4, 512),     # z rows split into tiles (ragged last tile: 96 = 3 x 32)
    (1, 3, 5, 48, 2, 1, 2, 256),
    (1, 16, 16, 0, 1, 1, 4, 1024),   # upper bounds far above what fits: shrunk by the library
    (1, 2, 2, 16, 0, 1, 0, 256),     # tiny tiles (below 256 voxels: left to the tiled kernel)
]


@pytest.mark.parametrize("kind", ["smooth", "rough"])
@pytest.mark.parametrize("nc", [1, 3])
@pytest.mark.parametrize("cfg", SHEAR_CFGS)
def test_shear_splat_any_configuration(ext, kind, nc, cfg):
    u, I, go, want = fields(kind, nc)
    du_, dI_, dgo = dev(u), dev(I), dev(go)
    for mc in ((2, 1, 0) if nc > 1 else (2,)):
        ext.set_splat_shear(*cfg)
        ext.set_splat_shear_mc(mc)
        try:
            for dt in (1.0, -0.2):
                oI, ou = want[dt]
                dI, du = ext.interp_backward(dgo, dI_, du_, dt, True, True)
                assert_bits(du, ou, f"shear sweep d_u {kind} C={nc} dt={dt} mc={mc} cfg={cfg}")
                assert_close(dI, oI, torch.float32, f"shear sweep d_I ({kind} C={nc} dt={dt} mc={mc} cfg={cfg})")
                dI2, _ = ext.interp_backward(dgo, dI_, du_, dt, True, False)
                assert_close(dI2, oI, torch.float32, f"shear sweep d_I only ({kind} C={nc} dt={dt} cfg={cfg})")
        finally:
            ext.set_splat_shear(*DEFAULT_SHEAR)
            ext.set_splat_shear_mc(2)


@pytest.mark.parametrize("kind", ["smooth", "rough"])
@pytest.mark.parametrize("nc", [1, 3])
@pytest.mark.parametrize("tile", [(0, 8, 0, 1, 1, 4, 512), (4, 4, 0, 1, 1, 16, 256), (2, 3, 16, 0, 0, 0, 256),
                                  (8, 8, 0, 2, 2, 16, 512), (16, 16, 16, 2, 2, 2, 1024)])
def test_f32_tiled_splat_forced(ext, kind, nc, tile):
    """`set_splat_shear(on=0)`: the float32 call reaches splat_tiled_kernel (otherwise the sheared kernel takes every
    float32 3D call first), in its multi-channel and its channel-by-channel form."""
    u, I, go, want = fields(kind, nc)
    ext.set_splat_shear(0, *DEFAULT_SHEAR[1:])
    ext.set_splat_tile(*tile)
    try:
        for mc in ((1, 0) if nc > 1 else (1,)):
            ext.set_splat_mc(mc)
            for dt in (1.0, -0.2):
                oI, ou = want[dt]
                dI, du = ext.interp_backward(dev(go), dev(I), dev(u), dt, True, True)
                assert_bits(du, ou, f"tiled f32 d_u {kind} C={nc} dt={dt} mc={mc} tile={tile}")
                assert_close(dI, oI, torch.float32, f"tiled f32 d_I ({kind} C={nc} dt={dt} mc={mc} tile={tile})")
    finally:
        ext.set_splat_shear(*DEFAULT_SHEAR)
        ext.set_splat_tile(0, 8, 0, 1, 1, 4, 512)
        ext.set_splat_mc(1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("sp", [(20, 12, 40), (9, 11, 33)])
@pytest.mark.parametrize("shear,mode", [(1, 1), (0, 1), (1, 0)])
@pytest.mark.parametrize("bc", [False, True])
def test_fused_backward_production_combination(ext, dtype, sp, shear, mode, bc):
    """The calls of `_shoot_reverse` (lagomorph_amd/lddmm.py): need_I with addgo (u_mode 2), and need_I with a running
    d_u AND a running d_I (u_mode 1, i_mode 1), three channels, through the sheared kernel, the tiled kernel and the
    plain-atomic kernel, with and without a broadcast I."""
    rng = np.random.default_rng(hash((sp, shear, mode, bc)) % 2**31)
    u = _disp(rng, 2, sp, dtype)
    I = rnd(rng, (1 if bc else 2, 3) + sp, dtype)
    go = rnd(rng, (2, 3) + sp, dtype)
    startu = rnd(rng, (2, 3) + sp, dtype)
    startI = rnd(rng, I.shape, dtype)
    k = go.dtype.type
    ext.set_splat_shear(shear, *DEFAULT_SHEAR[1:])
    ext.set_splat_mode(mode)
    try:
        for dt in (1.0, -0.25):
            oI, ou = orc.interp_backward(go, I, u, dt, True, True)
            for mc in (2, 1, 0):
                ext.set_splat_shear_mc(mc)
                ext.set_splat_mc(1 if mc else 0)
                dI, du = ext.interp_backward_fused(dev(go), dev(I), dev(u), dt, True, addgo=-0.2)
                assert_close(dI, oI, dtype, f"fused d_I (need_I + addgo, dt={dt} mc={mc})")
                assert_close(du, (k(-0.2) * go).astype(np.float64) + ou, dtype, f"fused d_u (need_I + addgo, dt={dt} mc={mc})",
                             scale=np.abs(ou).max() + 0.2 * np.abs(go).max())
                run_I = dev(startI)
                dI, du = ext.interp_backward_fused(dev(go), dev(I), dev(u), dt, True, d_u=dev(startu), d_I=run_I)
                assert dI.data_ptr() == run_I.data_ptr()
                assert_close(dI, startI.astype(np.float64) + oI, dtype, f"fused d_I accumulated (dt={dt} mc={mc})",
                             scale=np.abs(oI).max() + np.abs(startI).max())
                assert_close(du, startu.astype(np.float64) + ou, dtype, f"fused d_u accumulated (dt={dt} mc={mc})",
                             scale=np.abs(ou).max() + np.abs(startu).max())
    finally:
        ext.set_splat_shear(*DEFAULT_SHEAR)
        ext.set_splat_mode(1)
        ext.set_splat_shear_mc(2)
        ext.set_splat_mc(1)
