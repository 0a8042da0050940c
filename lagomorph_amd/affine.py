"""Affine interpolation, regridding and small affine-group helpers.

Host-side mirror of the operator part of ``/root/reference/lagomorph/affine.py``
(``AffineInterpFunction`` :11-36, helpers :49-148, ``RegridFunction`` / ``regrid``
:151-285).  The HDF5-driven ``affine_atlas`` driver is outside this build's scope.
Kernels: ``csrc/affine.hip``.
"""
import torch

from . import lagomorph_ext


class AffineInterpFunction(torch.autograd.Function):
    """out(x) = I(A (x - c) + T + c), c = (shape - 1)/2   (affine.py:11-33)"""

    @staticmethod
    def forward(ctx, I, A, T):
        ctx.save_for_backward(I, A, T)
        return lagomorph_ext.affine_interp_forward(I.contiguous(), A.contiguous(), T.contiguous())

    @staticmethod
    def backward(ctx, grad_out):
        I, A, T = ctx.saved_tensors
        d_I, d_A, d_T = lagomorph_ext.affine_interp_backward(
            grad_out.contiguous(), I.contiguous(), A.contiguous(), T.contiguous(), *ctx.needs_input_grad
        )
        need = ctx.needs_input_grad
        return (d_I if need[0] else None, d_A if need[1] else None, d_T if need[2] else None)


affine_interp = AffineInterpFunction.apply


class AffineInterp(torch.nn.Module):
    def forward(self, I, A, T):
        return AffineInterpFunction.apply(I, A, T)


def det_2x2(A):
    a, b, c, d = A[:, 0, 0], A[:, 0, 1], A[:, 1, 0], A[:, 1, 1]
    return a * d - b * c


def invert_2x2(A):
    """Batched closed-form inverse of 2x2 matrices (adjugate / determinant)."""
    adj = torch.stack((A[:, 1, 1], -A[:, 0, 1], -A[:, 1, 0], A[:, 0, 0]), dim=1).view(-1, 2, 2)
    return adj / det_2x2(A).view(-1, 1, 1)


def minor(A, i, j):
    n = A.shape[1]
    rows = [r for r in range(n) if r != i]
    cols = [c for c in range(n) if c != j]
    return A[:, rows][:, :, cols]


def invert_3x3(A):
    """Batched closed-form inverse of 3x3 matrices: transposed cofactor matrix over the determinant."""
    cof = A.new_empty(A.shape)
    for i in range(3):
        for j in range(3):
            cof[:, i, j] = (-1) ** (i + j) * det_2x2(minor(A, i, j))
    det = (A[:, 0, :] * cof[:, 0, :]).sum(dim=1)
    return cof.transpose(1, 2) / det.view(-1, 1, 1)


def affine_inverse(A, T):
    """(A, T)^-1 = (A^-1, -A^-1 T)   (affine.py:108-122)"""
    assert A.shape[1] == A.shape[2] == T.shape[1]
    dim = A.shape[1]
    assert dim in (2, 3)
    Ainv = invert_2x2(A) if dim == 2 else invert_3x3(A)
    Tinv = -torch.matmul(Ainv, T.unsqueeze(2)).squeeze(2)
    return Ainv, Tinv


def rotation_exp_map(v):
    """Tangent vectors -> rotation matrices; 2D (vector of angles) only, like the reference (affine.py:125-141)."""
    if v.dim() == 1:
        c, s = torch.cos(v), torch.sin(v)
        return torch.stack((c, -s, s, c), dim=1).view(-1, 2, 2)
    if v.dim() == 2 and v.size(1) == 3:
        raise NotImplementedError()
    raise Exception(f"Cannot infer dimension from v shape {v.shape}")


def rigid_inverse(v, T):
    """(R(v), T)^-1 = (R(-v), -R(-v) T)   (affine.py:144-151)"""
    Rinv = rotation_exp_map(-v)
    return -v, -torch.matmul(Rinv, T.unsqueeze(2)).squeeze(2)


class RegridFunction(torch.autograd.Function):
    """Resample from one regular grid to another (affine.py:151-187).  In displacement mode the
    values are additionally divided by the spacing (affine.py:165-173)."""

    @staticmethod
    def forward(ctx, I, outshape, origin, spacing, displacement):
        outshape = [int(s) for s in outshape]
        origin = [float(o) for o in origin]
        spacing = [float(s) for s in spacing]
        ctx.inshape = tuple(I.shape[2:])
        ctx.outshape = outshape
        ctx.outorigin = origin
        ctx.outspacing = spacing
        ctx.displacement = displacement
        reg = lagomorph_ext.regrid_forward(I.contiguous(), outshape, origin, spacing)
        if displacement:
            dim = I.dim() - 2
            if I.shape[1] != dim:
                raise ValueError("Incorrect num channels for regridding displacement")
            # torch.Tensor(spacing) is float32 in the reference: 1/spacing is rounded through float32
            ctx.spacing_tensor = 1.0 / torch.tensor(spacing, dtype=torch.float32).to(reg.dtype).to(
                reg.device
            ).view(1, dim, *[1] * dim)
            reg.mul_(ctx.spacing_tensor)
        return reg

    @staticmethod
    def backward(ctx, grad_out):
        d_I = lagomorph_ext.regrid_backward(grad_out.contiguous(), ctx.inshape, ctx.outshape, ctx.outorigin,
                                            ctx.outspacing)
        if ctx.displacement:
            d_I.mul_(ctx.spacing_tensor)
        return d_I, None, None, None, None


def regrid(I, shape=None, origin=None, spacing=None, displacement=False):
    """Interpolate from one regular grid to another (affine.py:190-272).

    Only the argument combinations the reference implements are accepted: ``shape`` alone
    (origin = centre of the input, spacing = (in-1)/(out-1), so corner voxels coincide); every
    other combination raises exactly as the reference does."""
    if shape is None:
        if origin is None:
            if spacing is None:
                raise ValueError("At least one of shape, origin, or spacing required")
            raise NotImplementedError
        if spacing is None:
            raise NotImplementedError
        raise ValueError("Shape is required if specifying origin and spacing")
    d = I.dim() - 2
    if not isinstance(shape, (list, tuple, torch.Size)):
        shape = tuple([shape] * d)
    if origin is not None:
        raise NotImplementedError
    origin = tuple((s - 1) * 0.5 for s in I.shape[2:])
    if spacing is None:
        spacing = tuple((sI - 1) / (s - 1) for sI, s in zip(I.shape[2:], shape))
    if not isinstance(spacing, (list, tuple)):
        spacing = tuple([spacing] * d)
    assert len(shape) == d and len(origin) == d and len(spacing) == d
    return RegridFunction.apply(I, shape, origin, spacing, displacement)


class RegridModule(torch.nn.Module):
    def __init__(self, shape, origin, spacing):
        super().__init__()
        self.shape, self.origin, self.spacing = shape, origin, spacing

    def forward(self, I):
        return regrid(I, self.shape, self.origin, self.spacing)
