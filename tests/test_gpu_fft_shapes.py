"""GPU: the hand-written FFT passes of the fluid metric over the stage plans they are instantiated for -- 256 points
(two groups of four levels), 192 (radix 3 in front of 3 + 3), 96 / 160 (radix-6 / radix-10 stage), 64 and 128, mixed
planes, persistent and one-shot forms -- against the float64 path (rocFFT, spot-checked on first use) at 2e-6 of the
result's maximum (observed 3-5e-7).  The host emulation (tests/test_fft_emulation.py) checks the same code thread by
thread without a GPU; this is the hardware side of it."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(256, 64, 128), (64, 256, 64), (256, 128, 256), (192, 192, 192), (96, 96, 96), (64, 64, 64), (128, 32, 64),
          (160, 96, 64), (96, 160, 192), (128, 128, 160)]


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
