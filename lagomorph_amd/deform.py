"""Displacement-field interpolation and composition.

Host-side mirror of ``/root/reference/lagomorph/deform.py`` (same names,
arguments and semantics); the heavy lifting is ``lagomorph_ext.interp_*``
(HIP kernels, ``csrc/interp.hip`` / ``csrc/splat.hip``).
"""
import numpy as np
import torch

from . import lagomorph_ext


def identity(defshape, dtype=np.float32):
    """Identity map x -> x as a numpy array of shape N, d, *spatial (deform.py:10-21)."""
    dim = len(defshape) - 2
    ix = np.empty(defshape, dtype=dtype)
    for d in range(dim):
        bshape = [1] * len(defshape)
        bshape[d + 2] = defshape[d + 2]
        ix[:, d, ...] = np.arange(defshape[d + 2], dtype=dtype).reshape(bshape)
    return ix


class InterpFunction(torch.autograd.Function):
    """out(x) = I(x + dt*u(x)); backward = splat (d_I) + analytic gradient (d_u).  deform.py:24-41"""

    @staticmethod
    def forward(ctx, I, u, dt):
        ctx.dt = dt
        ctx.save_for_backward(I, u)
        return lagomorph_ext.interp_forward(I.contiguous(), u.contiguous(), dt)

    @staticmethod
    def backward(ctx, gradout):
        I, u = ctx.saved_tensors
        d_I, d_u = lagomorph_ext.interp_backward(
            gradout.contiguous(), I.contiguous(), u.contiguous(), ctx.dt, *ctx.needs_input_grad[:2]
        )
        return d_I, d_u, None


def interp(I, u, dt=1.0):
    return InterpFunction.apply(I, u, dt)


def interp_hessian_diagonal_image(I, u, dt=1.0):
    """Hessian diagonal w.r.t. I of interp(I, u, dt) (deform.py:48-50; 2D only)."""
    return lagomorph_ext.interp_hessian_diagonal_image(I, u, dt)


class ComposeFunction(torch.autograd.Function):
    """ds*u + dt*interp(v, u, ds) as ONE kernel (csrc/fused.hip).  Used by compose() when both
    arguments are vector fields of the same shape; bit-identical to the unfused expression."""

    @staticmethod
    def forward(ctx, u, v, ds, dt):
        ctx.ds, ctx.dt = ds, dt
        ctx.save_for_backward(u, v)
        return lagomorph_ext.compose(u.contiguous(), v.contiguous(), ds, dt)

    @staticmethod
    def backward(ctx, gradout):
        u, v = ctx.saved_tensors
        gradout = gradout.contiguous()
        need_u, need_v = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if need_u and ctx.dt == 1.0 and hasattr(lagomorph_ext, "interp_backward_fused"):
            # d_u = (D_u interp)^T grad + ds * grad in one kernel (the sum of each voxel's d_u starts from ds * grad)
            d_v, d_u = lagomorph_ext.interp_backward_fused(gradout, v.contiguous(), u.contiguous(), ctx.ds, need_v,
                                                           addgo=ctx.ds)
            return d_u, d_v if need_v else None, None, None
        gi = gradout if ctx.dt == 1.0 else ctx.dt * gradout  # gradient reaching the interp output
        d_v, d_u = lagomorph_ext.interp_backward(gi, v.contiguous(), u.contiguous(), ctx.ds, need_v, need_u)
        if need_u:
            d_u.add_(gradout, alpha=ctx.ds)  # d_u is a fresh tensor owned by this call
        return d_u, d_v, None, None


def compose(u, v, ds=1.0, dt=1.0):
    """ds*u(x) + dt*v(x + ds*u(x))   (deform.py:53-55)"""
    if u.shape == v.shape and u.size(1) == u.dim() - 2 and u.dtype == v.dtype:
        return ComposeFunction.apply(u, v, ds, dt)
    return ds * u + dt * interp(v, u, dt=ds)


def compose_disp_vel(u, v, dt=1.0):
    """dt*v(x) + u(x + dt*v(x))   (deform.py:58-62)"""
    return compose(v, u, ds=dt, dt=1.0)


def compose_vel_disp(v, u, dt=1.0):
    """u(x) + dt*v(x + u(x))   (deform.py:65-70)"""
    return compose(u, v, ds=1.0, dt=dt)
