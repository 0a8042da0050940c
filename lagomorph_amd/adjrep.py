r"""Adjoint representation of Diff(R^d) on velocities and momenta.

Host-side mirror of ``/root/reference/lagomorph/adjrep.py`` (same formulas).
"""
import torch

from . import lagomorph_ext
from .deform import interp
from .diff import jacobian_times_vectorfield, jacobian_times_vectorfield_adjoint

USE_FUSED_AD_STAR = True


class AdStarFunction(torch.autograd.Function):
    """(D phiinv + I) (m o (id + phiinv)) as ONE kernel (csrc/fused.hip), bit-identical to the
    two-call sequence of adjrep.py:86-97.  When a gradient is wanted the kernel also stores the resampled
    momentum (12 bytes per voxel of the 288 GB instead of a 36-byte-per-voxel recomputation in the backward);
    the backward is the chain of the two reference backward kernels, the second one adding its displacement
    gradient straight onto the first one's (lago_interp_backward_fused: no separate add pass)."""

    @staticmethod
    def forward(ctx, phiinv, m):
        phiinv, m = phiinv.contiguous(), m.contiguous()
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            out, mphiinv = lagomorph_ext.Ad_star(phiinv, m, save_resampled=True)
            ctx.save_for_backward(phiinv, m, mphiinv)
            return out
        return lagomorph_ext.Ad_star(phiinv, m)

    @staticmethod
    def backward(ctx, gradout):
        phiinv, m, mphiinv = ctx.saved_tensors
        gradout = gradout.contiguous()
        need_phi, need_m = ctx.needs_input_grad
        d_v, d_w = lagomorph_ext.jacobian_times_vectorfield_backward(gradout, phiinv, mphiinv, True, False, need_phi, True)
        if need_phi and hasattr(lagomorph_ext, "interp_backward_fused"):
            # d_phi = d_v + d_u, the second term accumulated onto the first inside the splat kernel
            d_m, d_phi = lagomorph_ext.interp_backward_fused(d_w, m, phiinv, 1.0, need_m, d_u=d_v)
            return d_phi, d_m if need_m else None
        d_m, d_u = lagomorph_ext.interp_backward(d_w, m, phiinv, 1.0, need_m, need_phi)
        return d_v.add_(d_u) if need_phi else None, d_m if need_m else None


def ad(v, w):
    r"""ad(v, w) = Dv w - Dw v   (adjrep.py:37-47)"""
    return jacobian_times_vectorfield(v, w, displacement=False) - jacobian_times_vectorfield(
        w, v, displacement=False
    )


def Ad(phi, v):
    """Big adjoint action; not implemented in the reference either (adjrep.py:50-66)."""
    raise NotImplementedError


class AdStarSmallFunction(torch.autograd.Function):
    """ad^*(v, m) as ONE stencil kernel (csrc/diff.hip), bit-identical to the three-call sequence of
    adjrep.py:69-83; the backward is the difference of the two reference backward kernels."""

    @staticmethod
    def forward(ctx, v, m):
        ctx.save_for_backward(v, m)
        return lagomorph_ext.ad_star(v.contiguous(), m.contiguous())

    @staticmethod
    def backward(ctx, gradout):
        v, m = ctx.saved_tensors
        v, m, gradout = v.contiguous(), m.contiguous(), gradout.contiguous()
        need_v, need_m = ctx.needs_input_grad
        dA_v, dA_m = lagomorph_ext.jacobian_times_vectorfield_backward(gradout, v, m, False, True, need_v, need_m)
        dB_m, dB_v = lagomorph_ext.jacobian_times_vectorfield_adjoint_backward(gradout, m, v, need_m, need_v)
        return dA_v.sub_(dB_v) if need_v else None, dA_m.sub_(dB_m) if need_m else None


def ad_star(v, m):
    r"""ad^*(v, m) = (Dv)^T m + Dm v + m div v, as the numerical adjoint of ad(v, .)  (adjrep.py:69-83)"""
    if (USE_FUSED_AD_STAR and hasattr(lagomorph_ext, "ad_star") and v.shape == m.shape
            and m.size(1) == m.dim() - 2 and v.dtype == m.dtype):
        return AdStarSmallFunction.apply(v, m)
    return jacobian_times_vectorfield(v, m, displacement=False, transpose=True) - jacobian_times_vectorfield_adjoint(
        m, v
    )


def Ad_star(phiinv, m):
    r"""Ad^*(phi, m)(x) = (D phi(x)) m(phi(x)); note the non-transposed product (adjrep.py:86-97)"""
    if (USE_FUSED_AD_STAR and hasattr(lagomorph_ext, "Ad_star") and phiinv.shape == m.shape
            and m.size(1) == m.dim() - 2 and phiinv.dtype == m.dtype):
        return AdStarFunction.apply(phiinv, m)
    mphiinv = interp(m, phiinv)
    return jacobian_times_vectorfield(phiinv, mphiinv, displacement=True)


def ad_dagger(x, y, metric):
    r"""ad^dagger(x, y) = ad^*(x, y^flat)^sharp   (adjrep.py:104-113)"""
    return metric.sharp(ad_star(x, metric.flat(y)))


def Ad_dagger(phi, y, metric):
    r"""Ad^dagger(phi, y) = Ad^*(phi, y^flat)^sharp   (adjrep.py:116-122)"""
    return metric.sharp(Ad_star(phi, metric.flat(y)))


def sym(x, y, metric):
    r"""sym(x, y) = -(ad^dagger(x, y) + ad^dagger(y, x))   (adjrep.py:125-135)"""
    return -(ad_dagger(x, y, metric) + ad_dagger(y, x, metric))


def sym_dagger(x, y, metric):
    r"""sym^dagger(x, y) = ad^dagger(y, x) - ad(x, y)   (adjrep.py:138-145)"""
    return ad_dagger(y, x, metric) - ad(x, y)
