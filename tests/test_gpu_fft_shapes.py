"""GPU: the hand-written FFT passes of the fluid metric over the stage plans they are instantiated for -- 256 points
(two groups of four levels), 192 (radix 3 in front of 3 + 3), 96 / 160 (radix-6 / radix-10 stage), 64 and 128, 176 / 208
(radix 11 / 13 in front of four levels; half lengths 88 / 104), 112 / 224, 144, 240 (odd factors 7, 9, 15), mixed
planes, persistent and one-shot forms -- against the float64 path (rocFFT, spot-checked on first use) at 2e-6 of the
result's maximum (observed 3-5e-7).  The host emulation (tests/test_fft_emulation.py) checks the same code thread by
thread without a GPU; this is the hardware side of it."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(256, 64, 128), (64, 256, 64), (256, 128, 256), (192, 192, 192), (96, 96, 96), (64, 64, 64), (128, 32, 64),
          (160, 96, 64), (96, 160, 192), (128, 128, 160),
          # radix 11 and 13 (round 6): the 176 x 208 x 176 brain grid and its permutations, mixed with the older lengths
          (176, 208, 176), (208, 176, 176), (176, 176, 208), (176, 64, 64), (208, 96, 128), (64, 208, 176), (160, 176, 176),
          # odd factors 7, 9, 15: cropped MNI grids (96 x 112 x 96, 192 x 224 x 160), 144^3, 144 x 176 x 144, 240-point axes
          (96, 112, 96), (112, 112, 112), (112, 128, 112), (112, 112, 128), (192, 224, 160), (224, 160, 224), (128, 224, 128),
          (144, 144, 144), (144, 176, 144), (176, 144, 176), (240, 160, 240), (160, 240, 160),
          # 8 x odd: half a tile at the end of the Nyquist plane -- half-resolution grids of 176 x 208 x 176 and 160^3, 120^3
          (88, 104, 88), (104, 88, 88), (88, 88, 104), (120, 120, 120), (80, 80, 80), (120, 88, 88), (88, 120, 120), (104, 104, 88),
          # planes above the LDS: rows + columns around the x pass (five launches)
          (256, 256, 256), (64, 256, 256), (160, 192, 224), (192, 224, 192), (224, 224, 224), (96, 256, 192), (128, 224, 256),
          (64, 176, 160), (112, 144, 128), (176, 208, 192), (240, 240, 224), (80, 128, 224),
          (96, 208, 208), (64, 256, 176), (240, 240, 240), (128, 224, 144), (176, 192, 208)]


@pytest.mark.parametrize("shape", SHAPES, ids=[f"{a}x{b}x{c}" for a, b, c in SHAPES])
def test_fft_passes_match_the_float64_path(shape):
    import lagomorph_amd as lm

    ext = lm.lagomorph_ext
    met = lm.FluidMetric([0.1, 0.05, 0.01])
    g = torch.Generator(device="cuda").manual_seed(sum(shape))
    m = torch.randn((2, 3) + shape, device="cuda", generator=g)
    for f in (met.sharp, met.flat):
        before = ext.path_launches("fluid_lds")
        out = f(m)
        assert ext.path_launches("fluid_lds") == before + 1   # the three LDS-tiled passes, not a fallback
        ref = f(m.double())
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        assert err <= 2e-6, (shape, err)


def test_every_multiple_of_16_from_64_to_256_runs_the_lds_passes():
    """The rule the instantiation lists add up to (round 6): any float32 volume whose three extents are multiples of 16 between
    64 and 256 takes the LDS-tiled passes (three launches where its plane has a one-kernel instantiation, rows + columns
    otherwise), never the generic passes or rocFFT.  A seeded sample of 40 such shapes against the float64 path."""
    import random

    import lagomorph_amd as lm

    ext = lm.lagomorph_ext
    met = lm.FluidMetric([0.1, 0.05, 0.01])
    rnd = random.Random(16)
    sizes = list(range(64, 257, 16))
    g = torch.Generator(device="cuda").manual_seed(16)
    for _ in range(40):
        shape = tuple(rnd.choice(sizes) for _ in range(3))
        m = torch.randn((1, 3) + shape, device="cuda", generator=g)
        before = ext.path_launches("fluid_lds")
        out = met.sharp(m)
        assert ext.path_launches("fluid_lds") == before + 1, shape
        ref = met.sharp(m.double())
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        assert err <= 2e-6, (shape, err)
        del m, out, ref


@pytest.mark.parametrize("sp,dtype,mode", [
    ((128, 128, 128), torch.float32, 3), ((160, 160, 160), torch.float32, 3),   # tuned passes (persistent zy at 160)
    ((64, 96, 128), torch.float32, 3), ((32, 32, 64), torch.float32, 3),
    ((128, 128), torch.float32, 3), ((64, 256), torch.float32, 3),              # the fused 2D kernel
    ((24, 20, 28), torch.float32, 3), ((100, 100), torch.float32, 3),           # generic passes
    ((24, 20, 28), torch.float64, 3), ((30, 36), torch.float64, 3),
    ((64, 40, 40), torch.float32, 1), ((24, 20, 28), torch.float32, 0), ((24, 20, 28), torch.float64, 0)])   # rocFFT forms
def test_scaled_operator_has_the_bits_of_a_separate_multiply(sp, dtype, mode):
    """`lago_fluid_metric_scaled`: out_scale * K(m) is the finished value times the factor -- bit for bit what
    `K(m) * out_scale` gives, on every implementation of the operator (the factor rides in the last kernel of the
    tuned 3D passes and of the 2D kernel; the other paths run one in-place pass), both directions, and through the
    autograd node (the backward of s K is s K)."""
    import lagomorph_amd as lm

    g = torch.Generator(device="cuda").manual_seed(11)
    m = torch.randn((3, len(sp)) + sp, device="cuda", dtype=dtype, generator=g)
    met = lm.FluidMetric([0.1, 0.05, 0.01])
    lm.lagomorph_ext.set_fluid_mode(mode)
    try:
        for s in (-0.1, 1.0 / 3.0, -1.0, 7.5):
            v = met.sharp(m)
            assert torch.equal(met.sharp(m, out_scale=s), v * s), (sp, dtype, mode, s)
            f = met.operator(m, inverse=False)
            assert torch.equal(met.operator(m, inverse=False, out_scale=s), f * s), (sp, dtype, mode, s)
        assert torch.equal(met.sharp(m, out_scale=1.0), met.sharp(m))
        mr = m.clone().requires_grad_(True)
        go = torch.randn(m.shape, device="cuda", dtype=dtype, generator=g)
        met.sharp(mr, out_scale=-0.25).backward(go)
        assert torch.equal(mr.grad, met.sharp(go) * -0.25)
    finally:
        lm.lagomorph_ext.set_fluid_mode(3)


def test_first_euler_step_uses_the_scaled_operator():
    """`expmap` from the identity: phi_1 = -dt sharp(m0) comes out of ONE call of the operator (no multiply pass), with
    the bits of the two-call form."""
    import lagomorph_amd as lm
    from lagomorph_amd import lddmm

    g = torch.Generator(device="cuda").manual_seed(5)
    m = torch.randn((2, 3, 32, 32, 64), device="cuda", generator=g)
    met = lm.FluidMetric([0.1, 0.0, 0.01])
    one = lddmm._first_step(met, m, 0.1)
    assert torch.equal(one, met.sharp(m) * -0.1)
    v0 = met.sharp(m)
    assert torch.equal(lddmm._first_step(met, m, 0.1, v0=v0), one)
    assert torch.equal(lm.expmap(met, m, num_steps=1), met.sharp(m) * -1.0)   # (dt = 1)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("sp,B", [((64, 48, 80), 3), ((33, 29, 31), 2), ((100, 120, 60), 2), ((91, 77, 55), 2), ((7, 9, 6), 3),
                                  ((2, 3, 4), 2), ((59, 64, 64), 2), ((128, 30, 34), 2), ((26, 40, 44), 1),
                                  ((256, 192), 3), ((512, 40), 2), ((1024, 24), 2), ((2048, 10), 1), ((256, 14, 18), 2), ((512, 6, 10), 1),
                                  ((100, 90), 2), ((33, 21), 3), ((59, 40), 2), ((26, 30), 2), ((5, 4), 2)])
def test_generic_passes_fused_x_pass_has_the_bits_of_the_separate_launches(sp, B, dtype):
    """csrc/fftg.hip: fft_xop_kernel (x forward + operator + x inverse in one launch, round 6; `fluid_mode` 3) against the three
    separate launches (`fluid_mode` 4) on the generic path: the same stages on the same roots and the operator of
    fluid_bin.hpp, hence the same bits -- sharp and flat, with and without the beta term, 3D (along x) and 2D (along the
    fields' first axis); lengths with a prime factor of 29
    or more (59: a Bluestein line) and the radix-13 instantiation (26 = 2 x 13) keep the separate launches in both modes;
    power-of-two lines of 256 ... 1024 points run the fused pass with in-place stages (2048: one column only, ping-pong)."""
    import lagomorph_amd as lm

    ext = lm.lagomorph_ext
    g = torch.Generator(device="cuda").manual_seed(sum(sp))
    x = torch.randn((B, len(sp)) + sp, device="cuda", generator=g, dtype=dtype)
    try:
        for params in ([0.1, 0.0, 0.01], [0.1, 0.05, 0.01]):
            met = lm.FluidMetric(params)
            res = {}
            for mode in (3, 4):
                ext.set_fluid_mode(mode)
                before = ext.path_launches("fluid_generic")
                res[mode] = (met.sharp(x), met.flat(x), met.sharp(x, out_scale=-0.25))
                if dtype == torch.float64 or (len(sp) == 3 and sp[0] not in (128, 64)):   # (float32 2D planes up to 128 x 128: fused 2D kernel)
                    assert ext.path_launches("fluid_generic") == before + 3, "not the generic passes"
            for a, b in zip(res[3], res[4]):
                assert torch.equal(a, b), (sp, dtype, params)
            ref = met.sharp(x.double())
            assert float((res[3][0].double() - ref).abs().max() / ref.abs().max()) <= (2e-6 if dtype == torch.float32 else 1e-12)
    finally:
        ext.set_fluid_mode(3)
