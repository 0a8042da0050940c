"""Fluid (and other) LDDMM metrics.

Host-side mirror of ``/root/reference/lagomorph/metric.py``.  The FFTs are
rocFFT through ``torch.fft`` (the reference used ``torch.rfft``/``irfft``,
removed in torch 1.8; ``rfftn(norm="ortho")`` + ``view_as_real`` is the same
transform and layout); the per-frequency operator is ``csrc/metric.hip``.
"""
import itertools

import numpy as np
import torch

from . import lagomorph_ext


def _rfft(x, spatial_dim):
    dims = tuple(range(-spatial_dim, 0))
    return torch.view_as_real(torch.fft.rfftn(x, dim=dims, norm="ortho").contiguous())


def _irfft(Fx, spatial_dim, signal_sizes):
    dims = tuple(range(-spatial_dim, 0))
    return torch.fft.irfftn(torch.view_as_complex(Fx), s=tuple(signal_sizes), dim=dims, norm="ortho")


def _apply(mv, inverse, luts, params, out_scale=1.0):
    """rFFT -> per-frequency operator -> inverse rFFT (times `out_scale`).  On the GPU this is one C-ABI call
    (hipFFT on caller buffers + csrc/metric.hip, 1/N folded into the kernel); the three-call
    form below is the reference's literal sequence and is what the extension surface exposes."""
    if USE_FUSED_FLUID and hasattr(lagomorph_ext, "fluid_metric"):
        if out_scale != 1.0:
            return lagomorph_ext.fluid_metric(mv.contiguous(), inverse, luts["cos"], luts["sin"], *params,
                                              lut_generation=luts.get("gen", 0), out_scale=out_scale)
        return lagomorph_ext.fluid_metric(mv.contiguous(), inverse, luts["cos"], luts["sin"], *params,
                                          lut_generation=luts.get("gen", 0))
    sh = mv.shape
    spatial_dim = len(sh) - 2
    Fmv = _rfft(mv.contiguous(), spatial_dim)
    lagomorph_ext.fluid_operator(Fmv, inverse, luts["cos"], luts["sin"], *params)
    out = _irfft(Fmv, spatial_dim, sh[2:])
    return out if out_scale == 1.0 else out * out_scale


USE_FUSED_FLUID = True


class FluidMetricOperator(torch.autograd.Function):
    """rFFT -> per-frequency L^2 (flat) or L^-2 (sharp) -> inverse rFFT (metric.py:9-34).
    The operator is self-adjoint, so backward applies the same operator to the output gradient."""

    @staticmethod
    def forward(ctx, params, luts, inverse, mv, out_scale=1.0):
        ctx.params = params
        ctx.luts = luts
        ctx.inverse = inverse
        ctx.out_scale = out_scale
        return _apply(mv, inverse, luts, params, out_scale)

    @staticmethod
    def backward(ctx, outgrad):
        return None, None, None, _apply(outgrad, ctx.inverse, ctx.luts, ctx.params, ctx.out_scale), None


def fluid_luts(spatial_shape, dtype, device):
    """cos_k = 2(1 - cos 2 pi k / N), sin_k = sin 2 pi k / N per axis; the last axis has
    N//2+1 entries.  Built in float64 and rounded through float32 before the cast to
    `dtype`, because the reference passes them through ``torch.Tensor(...)`` (metric.py:66-75)."""
    cos, sin = [], []
    nd = len(spatial_shape)
    for d, N in enumerate(spatial_shape):
        Nf = N // 2 + 1 if d == nd - 1 else N
        k = np.arange(Nf)
        c = (2.0 * (1.0 - np.cos(2 * np.pi * k / N))).astype(np.float32)
        s = np.sin(2.0 * np.pi * k / N).astype(np.float32)
        cos.append(torch.from_numpy(c).to(dtype).to(device))
        sin.append(torch.from_numpy(s).to(dtype).to(device))
    # "gen" names these LUT contents for the library's coefficient-table cache (include/lagomorph_hip.h):
    # a fresh number per LUT set, so a later set that lands on recycled device addresses cannot hit a stale table
    return {"cos": cos, "sin": sin, "gen": next(_LUT_GENERATION)}


_LUT_GENERATION = itertools.count(1)


class FluidMetric(object):
    """L'L = -alpha lap - beta grad div + gamma  (as coded in the reference kernel; metric.py:37-97).

    One deliberate difference: the reference builds its LUTs once and silently reuses them
    for later inputs of another shape/dtype/device (``if self.luts is None``, metric.py:63);
    here they are cached per (spatial shape, dtype, device)."""

    def __init__(self, params=[0.1, 0.0, 0.001]):
        assert len(params) == 3
        self.params = params
        self.shape = None
        self.complexshape = None
        self.luts = None
        self._lut_cache = {}

    def initialize_luts(self, shape, dtype, device="cuda"):
        shape = tuple(shape)
        if self.shape != shape:
            self.shape = shape
            cs = list(shape)
            cs[-1] = cs[-1] // 2 + 1
            self.complexshape = tuple(cs)
        key = (shape[2:], dtype, str(device))
        if key not in self._lut_cache:
            self._lut_cache[key] = fluid_luts(shape[2:], dtype, device)
        self.luts = self._lut_cache[key]

    def operator(self, mv, inverse, out_scale=1.0):
        self.initialize_luts(shape=mv.shape, dtype=mv.dtype, device=mv.device)
        return FluidMetricOperator.apply(self.params, self.luts, inverse, mv, float(out_scale))

    def sharp(self, m, out_scale=1.0):
        """momentum -> velocity (apply the Green's function).  `out_scale` (not in the reference): a factor on the
        result, the bits of `sharp(m) * out_scale` without the extra pass (lago_fluid_metric_scaled)."""
        return self.operator(m, inverse=True, out_scale=out_scale)

    def flat(self, m, out=None):
        """velocity -> momentum (apply the differential operator)."""
        return self.operator(m, inverse=False)


class Metric:
    """Metric factory from parsed arguments (metric.py:100-135)."""

    @staticmethod
    def add_args(parser):
        parser.add_argument("--metric_type", default="fluid", type=str,
                            help="Type of metric. Currently only 'fluid' is supported.")
        parser.add_argument("--fluid_alpha", default=0.1, type=float, help="Fluid parameter for vector Laplacian term")
        parser.add_argument("--fluid_beta", default=0.0, type=float, help="Fluid parameter for gradient divergence term")
        parser.add_argument("--fluid_gamma", default=0.01, type=float, help="Fluid parameter for L2 term")

    @classmethod
    def from_args(cls, args):
        if args.metric_type.lower() == "fluid":
            return FluidMetric(params=[args.fluid_alpha, args.fluid_beta, args.fluid_gamma])
