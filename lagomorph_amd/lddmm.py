"""LDDMM geodesic shooting (vector momentum EPDiff) and the atlas step.

Host-side mirror of ``/root/reference/lagomorph/lddmm.py:20-105`` (shooting)
and of the per-minibatch arithmetic of ``LDDMMAtlasBuilder``
(``lddmm.py:287-341``).  Data loading, HDF5 checkpoints and the CLI of the
reference are outside this build's scope.
"""
import torch
import torch.distributed as dist

from . import adjrep, deform
from .affine import regrid
from .metric import FluidMetric


def expmap_advect(metric, m, T=1.0, num_steps=10, phiinv=None):
    """Euler integration of d/dt m = -ad_v^* m without the integrated form (lddmm.py:20-36)."""
    if phiinv is None:
        phiinv = torch.zeros_like(m)
    dt = T / num_steps
    v = metric.sharp(m)
    phiinv = deform.compose_disp_vel(phiinv, v, dt=-dt)
    for _ in range(num_steps - 1):
        m = m - dt * adjrep.ad_star(v, m)
        v = metric.sharp(m)
        phiinv = deform.compose_disp_vel(phiinv, v, dt=-dt)
    return phiinv


def EPDiff_step(metric, m0, dt, phiinv, mommask=None):
    """One Euler step of the integrated EPDiff equation (lddmm.py:39-44):
    m = Ad^*_{phi^-1} m0;  v = m^sharp;  phi^-1 <- phi^-1 o (x - dt v)."""
    m = adjrep.Ad_star(phiinv, m0)
    if mommask is not None:
        m = m * mommask
    v = metric.sharp(m)
    return deform.compose_disp_vel(phiinv, v, dt=-dt)


def expmap(metric, m0, T=1.0, num_steps=10, phiinv=None, mommask=None, checkpoints=False):
    """Exponential map: returns the displacement of phi^-1 (lddmm.py:73-105).

    Only ``checkpoints=False`` is supported.  The reference's checkpointed branch is dead
    code (``EPDiffStepsFunction`` swaps dt/phiinv, lddmm.py:56,64; the integer branch
    computes step counts and then integrates nothing, lddmm.py:93-95), so it is rejected
    here instead of being reproduced."""
    if phiinv is None:
        phiinv = torch.zeros_like(m0)
    if checkpoints:
        raise NotImplementedError("expmap(checkpoints=...) is broken in the reference and not provided")
    dt = T / num_steps
    for _ in range(num_steps):
        phiinv = EPDiff_step(metric, m0, dt, phiinv, mommask=mommask)
    return phiinv


# --------------------------------------------------------------------------- atlas step


def lddmm_step(I, m, img, metric, dataset_size, integration_steps=5, reg_weight=1e2, learning_rate_pose=2e2,
               momentum_preconditioning=False):
    """One matching step of the atlas builder for a minibatch (lddmm.py:300-325).

    I: atlas image (1, 1, *sp) with requires_grad as the caller wishes (its .grad accumulates),
    m: momenta (B, d, *msp) -- updated in place by gradient descent, img: (B, 1, *sp).
    Returns (m, loss, reg_term) with loss/reg already scaled by B / dataset_size, all on device
    (no host synchronisation)."""
    m.requires_grad_(True)
    if m.grad is not None:
        m.grad.detach_()
        m.grad.zero_()
    regrid_momenta = tuple(m.shape[2:]) != tuple(I.shape[2:])
    h = expmap(metric, m, num_steps=integration_steps)
    if regrid_momenta:
        h = regrid(h, shape=I.shape[2:])
    Idef = deform.interp(I, h)
    v = metric.sharp(m)
    reg_term = reg_weight * (v * m).sum() / img.numel()
    if regrid_momenta:
        reg_term = reg_term * (I.numel() / v[0, 0, ...].numel())
    loss = ((Idef - img) ** 2).sum() / img.numel() + reg_term
    loss.backward()
    with torch.no_grad():
        norm_factor = img.shape[0] / dataset_size
        loss = (loss * norm_factor).detach()
        reg_term = (reg_term * norm_factor).detach()
        p = m.grad
        if momentum_preconditioning:
            p = metric.flat(p)
        m.add_(p, alpha=-learning_rate_pose)
    return m.detach(), loss, reg_term


class LDDMMAtlasBuilder:
    """Batch-sharded atlas building over in-memory volumes (lddmm.py:108-375, compute path only).

    Each rank owns a contiguous shard of the subjects and of their momenta, both resident in HBM
    (the reference parks momenta in pinned host memory and copies them every iteration,
    lddmm.py:236,328,337).  The only collectives are the SUM all-reduce of the atlas gradient per
    image update (lddmm.py:292-297), one all-reduce of the initial mean image (lddmm.py:196-198)
    and one of the two scalar losses per epoch (the reference reduces them every iteration and
    then calls .item(), lddmm.py:333-341)."""

    def __init__(self, images, batch_size=8, lddmm_integration_steps=5, image_update_freq=0, reg_weight=1e2,
                 learning_rate_pose=2e2, learning_rate_image=1e4, metric=None, momentum_shape=None,
                 momentum_preconditioning=False, I0=None, world_size=1, rank=0, dataset_size=None):
        self.images = images  # this rank's shard: (n_local, 1, *sp) on the device
        self.batch_size = batch_size
        self.lddmm_integration_steps = lddmm_integration_steps
        self.image_update_freq = image_update_freq
        self.reg_weight = reg_weight
        self.learning_rate_pose = learning_rate_pose
        self.learning_rate_image = learning_rate_image
        self.metric = metric if metric is not None else FluidMetric([0.1, 0, 0.01])  # lddmm.py:213
        self.momentum_preconditioning = momentum_preconditioning
        self.world_size = world_size
        self.rank = rank
        n_local = images.shape[0]
        self.dataset_size = dataset_size if dataset_size is not None else n_local * world_size
        dim = images.dim() - 2
        with torch.no_grad():
            if I0 is None:  # lddmm.py:186-198: mean image, all-reduced and averaged over ranks
                I0 = images.mean(dim=0, keepdim=True)
                if world_size > 1:
                    dist.all_reduce(I0)
                    I0 /= world_size
            self.I = I0.detach().clone().view(1, 1, *images.shape[2:])
        self.I.requires_grad_(True)
        self.image_optimizer = torch.optim.SGD([self.I], lr=learning_rate_image, weight_decay=0)
        self.image_optimizer.zero_grad()
        msp = tuple(momentum_shape) if momentum_shape is not None else tuple(images.shape[2:])
        self.ms = [
            torch.zeros((min(batch_size, n_local - b), dim) + msp, dtype=images.dtype, device=images.device)
            for b in range(0, n_local, batch_size)
        ]
        self.image_iters = 0
        self.epoch_losses, self.epoch_reg_terms = [], []

    def update_base_image(self, force=False):
        """lddmm.py:287-298"""
        if (self.image_iters < self.image_update_freq and not force) or self.image_iters == 0:
            return
        with torch.no_grad():
            if self.world_size > 1:
                dist.all_reduce(self.I.grad)
            self.I.grad = self.I.grad / (self.image_iters * self.world_size)
            self.image_optimizer.step()
            self.image_optimizer.zero_grad()
        self.image_iters = 0

    def iteration(self, b):
        img = self.images[b * self.batch_size:(b + 1) * self.batch_size]
        m, loss, reg = lddmm_step(self.I, self.ms[b], img, self.metric, self.dataset_size,
                                  integration_steps=self.lddmm_integration_steps, reg_weight=self.reg_weight,
                                  learning_rate_pose=self.learning_rate_pose,
                                  momentum_preconditioning=self.momentum_preconditioning)
        self.ms[b] = m
        self.image_iters += 1
        self.update_base_image()
        return loss, reg

    def epoch(self):
        """lddmm.py:343-362; returns (epoch_loss, epoch_reg_term) as 0-dim device tensors."""
        if self.image_update_freq == 0:
            self.image_optimizer.zero_grad()
        self.image_iters = 0
        tot = torch.zeros(2, dtype=self.images.dtype, device=self.images.device)
        for b in range(len(self.ms)):
            loss, reg = self.iteration(b)
            tot[0] += loss
            tot[1] += reg
        self.update_base_image(force=True)
        if self.world_size > 1:
            dist.all_reduce(tot)
        return tot[0], tot[1]

    def run(self, num_epochs=1):
        for _ in range(num_epochs):
            l, r = self.epoch()
            self.epoch_losses.append(l)
            self.epoch_reg_terms.append(r)
        return self.I.detach()

    # ---- checkpoint / resume (the reference writes the same fields to HDF5, lddmm.py:238-285) ----

    def state_dict(self):
        """Atlas, this rank's momenta (one tensor per minibatch, as the reference stores them with a
        `batch_sizes` attribute) and the loss histories -- what `LDDMMAtlasBuilder.save` of the
        reference writes (lddmm.py:238-262), as a plain dict of CPU tensors."""
        return {
            "atlas": self.I.detach().cpu().clone(),
            "momenta": [m.detach().cpu().clone() for m in self.ms],
            "batch_sizes": [int(m.shape[0]) for m in self.ms],
            "epoch_losses": [float(x) for x in self.epoch_losses],
            "epoch_reg_terms": [float(x) for x in self.epoch_reg_terms],
            "rank": self.rank, "world_size": self.world_size,
        }

    def load_state_dict(self, state):
        """Resume from `state_dict()` (reference: `load`, lddmm.py:264-285)."""
        if state["batch_sizes"] != [int(m.shape[0]) for m in self.ms]:
            raise ValueError("checkpoint was written with a different shard / batch size")
        with torch.no_grad():
            self.I.copy_(state["atlas"].to(self.I.device, self.I.dtype))
            for m, s in zip(self.ms, state["momenta"]):
                m.detach_().copy_(s.to(m.device, m.dtype))
        self.epoch_losses = list(state["epoch_losses"])
        self.epoch_reg_terms = list(state["epoch_reg_terms"])
        self.image_iters = 0
        self.image_optimizer.zero_grad()

    def save(self, path):
        torch.save(self.state_dict(), path)

    def load(self, path):
        self.load_state_dict(torch.load(path, map_location="cpu"))
