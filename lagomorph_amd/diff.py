"""Finite-difference Jacobian-times-vector-field operators.

Host-side mirror of ``/root/reference/lagomorph/diff.py``; kernels in
``csrc/diff.hip``.
"""
import torch

from . import lagomorph_ext


class JacobianTimesVectorFieldFunction(torch.autograd.Function):
    """(Dv [+ I]) w, or its transpose, with clamped central differences (diff.py:7-35)."""

    @staticmethod
    def forward(ctx, v, w, displacement, transpose):
        ctx.displacement = displacement
        ctx.transpose = transpose
        ctx.save_for_backward(v, w)
        return lagomorph_ext.jacobian_times_vectorfield_forward(v, w, displacement, transpose)

    @staticmethod
    def backward(ctx, gradout):
        v, w = ctx.saved_tensors
        d_v, d_w = lagomorph_ext.jacobian_times_vectorfield_backward(
            gradout, v, w, ctx.displacement, ctx.transpose, *ctx.needs_input_grad[:2]
        )
        return d_v, d_w, None, None


def jacobian_times_vectorfield(v, w, displacement=True, transpose=False):
    """Note the reference's default displacement=True (diff.py:38)."""
    return JacobianTimesVectorFieldFunction.apply(v, w, displacement, transpose)


class JacobianTimesVectorFieldAdjointFunction(torch.autograd.Function):
    """T(w)^dagger v, the adjoint of v -> (Dv) w (diff.py:42-58)."""

    @staticmethod
    def forward(ctx, v, w):
        ctx.save_for_backward(v, w)
        return lagomorph_ext.jacobian_times_vectorfield_adjoint_forward(v, w)

    @staticmethod
    def backward(ctx, gradout):
        v, w = ctx.saved_tensors
        d_v, d_w = lagomorph_ext.jacobian_times_vectorfield_adjoint_backward(
            gradout, v, w, *ctx.needs_input_grad[:2]
        )
        return d_v, d_w


jacobian_times_vectorfield_adjoint = JacobianTimesVectorFieldAdjointFunction.apply
