"""Affine interpolation, regridding and small affine-group helpers.

Host-side mirror of the operator part of ``/root/reference/lagomorph/affine.py``
(``AffineInterpFunction`` :11-36, helpers :49-148, ``RegridFunction`` / ``regrid``
:151-285) and of the compute loop of ``affine_atlas`` / ``StandardizedDataset`` (:288-438) on
device-resident tensors (the DataLoader / HDF5 plumbing around it is outside this build's scope).
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


# --------------------------------------------------------------------------- affine atlas


def batch_average(images, batch_size=50):
    """Mean over the first axis as the reference computes the initial atlas (`data.batch_average`, data.py:308-336,
    called at affine.py:328-333): minibatch sums in float64 folded into a running average, returned in the images'
    dtype, shape (1, *images.shape[1:]); on the device the images live on."""
    avg, seen = None, 0
    for b in range(0, images.size(0), batch_size):
        img = images[b:b + batch_size]
        sz = img.size(0)
        avi = img.to(torch.float64).sum(dim=0)
        avg = avi / sz if avg is None else avg * (seen / (seen + sz)) + avi / (seen + sz)
        seen += sz
    if avg is None:
        return torch.zeros_like(images[:1])
    if images.dtype in (torch.float32, torch.float64):
        avg = avg.to(images.dtype)
    return avg.unsqueeze(0)


def affine_atlas(images, As, Ts, I=None, num_epochs=1000, batch_size=50, image_update_freq=0, affine_steps=1,
                 reg_weightA=0e1, reg_weightT=0e1, learning_rate_A=1e-3, learning_rate_T=1e-2, learning_rate_I=1e5,
                 world_size=1, rank=0, dataset_size=None):
    """Affine atlas building on device-resident data: the compute loop of ``affine_atlas``
    (affine.py:288-415) without its DataLoader / HDF5 plumbing.

    images: (N, 1, *spatial) -- this rank's contiguous shard of the subjects (the reference shards
    with a DistributedSampler); As (N, d, d) and Ts (N, d): the matching shard of the affine
    parameters, stored as in the reference as the deviation from the identity (the map applied is
    A + eye, T).  They are updated in place and returned.  ``dataset_size`` is the total number of
    subjects over all ranks (default N * world_size).

    Same update rules as the reference: per minibatch ``affine_steps`` gradient steps on (A, T) with
    the image gradient accumulated on the last one; the image takes an SGD step every
    ``image_update_freq`` minibatches (0: once per epoch) on the gradient averaged over minibatches
    and ranks -- ONE all-reduce of the (1, 1, *spatial) gradient per image update.  Differences: the
    losses stay on the device until the end (no per-iteration host synchronisation) and the epoch
    loss is all-reduced once per epoch.

    Returns (I, As, Ts, epoch_losses, iter_losses) like the reference."""
    import torch.distributed as dist

    N = images.size(0)
    if dataset_size is None:
        dataset_size = N * world_size
    dev, dt = images.device, As.dtype
    sp = tuple(images.shape[2:])
    dim = len(sp)
    nvox = 1
    for s in sp:
        nvox *= s
    if I is None:
        with torch.no_grad():
            I = batch_average(images, batch_size).to(dt)
            if world_size > 1:
                dist.all_reduce(I)
                I /= world_size
    else:
        I = I.clone().to(dev)
    I = I.to(dt).view(1, 1, *sp).contiguous()
    eye = torch.eye(dim, dtype=dt, device=dev).view(1, dim, dim)
    epoch_losses, iter_losses = [], []
    grad_I = torch.zeros_like(I)

    def image_step(image_iters):
        # affine.py:389-396, 404-409: average over accumulated minibatches and ranks, plain SGD
        if world_size > 1:
            dist.all_reduce(grad_I)
        I.sub_(grad_I, alpha=learning_rate_I / (image_iters * world_size))

    for epoch in range(num_epochs):
        epoch_loss = torch.zeros((), dtype=dt, device=dev)
        if image_update_freq == 0 or epoch == 0:
            grad_I.zero_()
        image_iters = 0
        for b in range(0, N, batch_size):
            img = images[b:b + batch_size].to(dt)
            A = As[b:b + batch_size].detach().clone().contiguous()
            T = Ts[b:b + batch_size].detach().clone().contiguous()
            nb = img.size(0)
            for affit in range(affine_steps):
                A.requires_grad_(True)
                T.requires_grad_(True)
                A.grad = None
                T.grad = None
                last = affit == affine_steps - 1
                Iv = I.detach().requires_grad_(last)  # the image gradient is accumulated on the last affine step only
                Idef = affine_interp(Iv, A + eye, T)
                regloss = 0.0
                if reg_weightA > 0:
                    regloss = regloss + 0.5 * reg_weightA * (A * A).sum()
                if reg_weightT > 0:
                    regloss = regloss + 0.5 * reg_weightT * (T * T).sum()
                loss = (((Idef - img) ** 2).sum() * (1.0 / nvox) + regloss) / nb
                loss.backward()
                with torch.no_grad():
                    li = loss.detach() * (nb / dataset_size)
                    iter_losses.append(li)
                    A.sub_(A.grad, alpha=learning_rate_A)
                    T.sub_(T.grad, alpha=learning_rate_T)
                    if last:
                        grad_I.add_(Iv.grad)
            image_iters += 1
            with torch.no_grad():
                if image_iters == image_update_freq:
                    image_step(image_iters)
                    grad_I.zero_()
                    image_iters = 0
                epoch_loss = epoch_loss + li
                As[b:b + batch_size] = A.detach()
                Ts[b:b + batch_size] = T.detach()
        with torch.no_grad():
            if image_iters > 0:
                image_step(image_iters)
            if world_size > 1:
                dist.all_reduce(epoch_loss)
        epoch_losses.append(epoch_loss)
    epoch_losses = [float(x) for x in epoch_losses]
    iter_losses = [float(x) for x in iter_losses]
    return I.detach(), As.detach(), Ts.detach(), epoch_losses, iter_losses


def save_affine_atlas(path, I, As, Ts, epoch_losses, iter_losses):
    """The result file of the reference's affine atlas tool (affine.py:581-587): HDF5 datasets `atlas`, `A`, `T`,
    `epoch_losses`, `iter_losses` when h5py is importable, the same arrays as an .npz archive otherwise."""
    import numpy as np

    st = {"atlas": I.detach().cpu().numpy(), "A": As.detach().cpu().numpy(), "T": Ts.detach().cpu().numpy(),
          "epoch_losses": np.asarray(epoch_losses), "iter_losses": np.asarray(iter_losses)}
    try:
        import h5py
    except ImportError:
        with open(path, "wb") as fh:
            np.savez(fh, **st)
        return path
    with h5py.File(path, "w") as f:
        for k, v in st.items():
            f.create_dataset(k, data=v)
    return path


def load_affine_atlas(path):
    """Reads what `save_affine_atlas` (or the reference's tool) wrote: (I, As, Ts, epoch_losses, iter_losses), tensors
    on the CPU -- the inputs `StandardizedDataset` needs (affine.py:589-600)."""
    import numpy as np

    with open(path, "rb") as fh:
        is_h5 = fh.read(8) == b"\x89HDF\r\n\x1a\n"
    if is_h5:
        try:
            import h5py
        except ImportError:
            raise RuntimeError(f"{path} is an HDF5 file but h5py is not importable")
        with h5py.File(path, "r") as f:
            st = {k: np.asarray(f[k]) for k in ("atlas", "A", "T", "epoch_losses", "iter_losses")}
    else:
        with np.load(path, allow_pickle=False) as z:
            st = {k: z[k] for k in z.files}
    return (torch.from_numpy(np.asarray(st["atlas"])), torch.from_numpy(np.asarray(st["A"])),
            torch.from_numpy(np.asarray(st["T"])), [float(x) for x in st["epoch_losses"]],
            [float(x) for x in st["iter_losses"]])


class StandardizedDataset:
    """Subjects resampled into atlas space by the inverse of their fitted affine map
    (affine.py:418-438); ``dataset[idx]`` is one (C, *spatial) image."""

    def __init__(self, dataset, As, Ts, device="cuda"):
        self.dataset, self.As, self.Ts, self.device = dataset, As, Ts, device
        dim = Ts.shape[1]
        self.eye = torch.eye(dim, dtype=As.dtype, device=device).view(1, dim, dim)

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        J = self.dataset[idx].to(self.device).unsqueeze(0)
        A = self.As[[idx], ...].to(self.device)
        T = self.Ts[[idx], ...].to(self.device)
        Ainv, Tinv = affine_inverse(A + self.eye, T)
        if J.dtype not in (torch.float32, torch.float64):
            J = J.to(torch.float32)
        return affine_interp(J.contiguous(), Ainv.contiguous(), Tinv.contiguous()).squeeze(0)
