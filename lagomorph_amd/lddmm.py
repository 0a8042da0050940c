"""LDDMM geodesic shooting (vector momentum EPDiff) and the atlas step.

Host-side mirror of ``/root/reference/lagomorph/lddmm.py:20-105`` (shooting)
and of the per-minibatch arithmetic of ``LDDMMAtlasBuilder``
(``lddmm.py:108-375``) on device-resident tensors.  Data loading and the CLI of
the reference are outside this build's scope.
"""
import threading

import numpy as np
import torch
import torch.distributed as dist

from . import adjrep, deform, lagomorph_ext
from .affine import regrid
from .metric import FluidMetric

USE_FUSED_EXPMAP = True


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


class ExpmapFunction(torch.autograd.Function):
    """The whole Euler integration of `expmap` as ONE autograd node with a hand-written reverse sweep.

    Forward: the loop of `EPDiff_step` (lddmm.py:39-44, :73-105) through the fused kernels, keeping per step
    phi_k, v_k and the resampled momentum m0 o (id + phi_k) -- the tensors autograd would keep as well.
    Backward, for k = N-1 .. 0 with G = dL/dphi_{k+1}:
        (d_phi_k, d_v_k) = compose^T: splat of G at x + ds v_k, and (D_u interp)^T G + ds G      (one kernel)
        d_m_k            = sharp(d_v_k)                                        (the operator is self-adjoint)
        d_phi_k         += jtv_backward(d_m_k; phi_k, m0 o (id + phi_k))        (added in place), giving d_w
        d_m0            += splat of d_w at x + phi_k;   d_phi_k += (D_u interp)^T d_w            (one kernel)
    Every chain-rule sum happens inside a kernel (lago_interp_backward_fused, lago_jtv_backward_acc): the thirteen
    elementwise add passes and nine memsets per five-step shoot that autograd's own accumulation costs (6.6 % + 1.7 %
    of the 160^3 atlas step, profiles/r02_atlas160_kernel_stats.md) are gone.  Same kernels and formulas as
    differentiating the loop, so the gradients agree to the rounding of the summation order."""

    @staticmethod
    def forward(ctx, metric, m0, phiinv, dt, num_steps, v0):
        m0 = m0.contiguous()
        keep = ctx.needs_input_grad[1] or ctx.needs_input_grad[2] or ctx.needs_input_grad[5]
        phi0 = None if phiinv is None else phiinv.contiguous()
        phi, steps, first = _shoot(metric, m0, phi0, dt, num_steps, v0, keep)
        # the two input tensors the reverse sweep reads go through save_for_backward, so that an in-place edit
        # between forward and backward raises instead of giving silently wrong gradients (phi0 may be the caller's
        # own tensor); the per-step tensors are intermediates nobody else holds
        ctx.save_for_backward(*([m0] if phi0 is None else [m0, phi0]))
        ctx.metric, ctx.dt, ctx.steps, ctx.first, ctx.has_v0 = metric, dt, steps, first, v0 is not None
        return phi

    @staticmethod
    @torch.autograd.function.once_differentiable   # the reverse sweep is not itself differentiable
    def backward(ctx, G):
        metric, dt = ctx.metric, ctx.dt
        saved = ctx.saved_tensors   # (version-checked)
        m0 = saved[0]
        need_m, need_phi, need_v0 = ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[5]
        d_m0, G = _shoot_reverse(metric, m0, dt, ctx.steps, G.contiguous())
        ctx.steps = None
        d_v0 = None
        if ctx.first:
            # phi_1 = -dt sharp(m0): what the general step's kernels return at phi_0 = 0 (the splats are identities,
            # the position gradients vanish), without running them
            d_v0 = G * (-dt)
            if not ctx.has_v0:   # sharp(m0) was computed here: through the (self-adjoint) operator onto d_m0
                d_m = metric.sharp(d_v0)
                d_m0 = d_m if d_m0 is None else d_m0.add_(d_m)
                d_v0 = None
            elif d_m0 is None:
                d_m0 = torch.zeros_like(m0)
            G = None
        return (None, d_m0 if need_m else None, G if need_phi else None, None, None,
                d_v0 if (need_v0 and ctx.has_v0) else None)


def _shoot(metric, m0, phiinv, dt, num_steps, v0, keep, out=None):
    """The Euler loop of `expmap` through the fused kernels.  Returns (phi, steps, first): steps = per general step
    (phi_k, v_k, m0 o (id + phi_k)) when `keep`, first = the first step was taken in closed form (`_first_step`).
    `out`: where the final displacement is to be written (a contiguous tensor of m0's shape)."""
    steps = []
    first = phiinv is None   # shooting from the identity
    phi = _first_step(metric, m0, dt, v0) if first else phiinv.contiguous()
    general = num_steps - 1 if first else num_steps
    for k in range(general):
        if keep:
            m, mphi = lagomorph_ext.Ad_star(phi, m0, save_resampled=True)
        else:
            m, mphi = lagomorph_ext.Ad_star(phi, m0), None
        v = metric.sharp(m)
        del m
        nxt = lagomorph_ext.compose(v, phi, -dt, 1.0, out=out if k + 1 == general else None)
        if keep:
            steps.append((phi, v, mphi))
        phi = nxt
    if out is not None and general == 0:
        out.copy_(phi)
        phi = out
    return phi, steps, first


# A forward-only shoot (no gradient wanted) of at least EXPMAP_MIN_ITEMS * EXPMAP_STREAMS batch items is cut into
# EXPMAP_STREAMS contiguous sub-batches (fewer when the batch does not have that many items) that run on HIP streams of their own: batch items are independent, and the tail of one part's
# kernel (the last workgroups of a launch leave most CUs idle) then overlaps with the head of another part's next
# kernel.  Measured (tools/ab_streams.py, profiles/r04_stream_split.md): 22.28 -> 21.66 ms at 32 x 128^3, 5.84 -> 5.55
# at 8 x 128^3; the driver's round-4 run 21.32 -> 21.14 ms; same bits (tests/test_gpu_lddmm_step.py:
# test_expmap_stream_split_same_bits).  ON by default since round 5 (2 parts, i.e. batches of 2 and more; single items
# and every shoot that keeps its steps for a backward pass run on the caller's stream alone); `EXPMAP_STREAMS = 1`
# switches it off -- bench.py does so for the pass it takes its per-kernel roofline from, because with two parts in
# flight a launch's duration is no longer the time the kernel needs by itself.
# Memory: the temporaries of a part (Ad_star / sharp / compose outputs, the FFT work buffer) are allocated on its side
# stream, i.e. in that stream's pool of torch's caching allocator, and a block cached there is not handed to an
# allocation of another stream.  A loop that alternates forward-only shoots with work on the caller's stream therefore
# keeps both sets of blocks reserved: measured with tools/measure_split_memory.py (profiles/r06_split_memory.md) --
# peak ALLOCATED is the same with and without the split, peak RESERVED grows by what one part's temporaries take.
# Near the capacity of the device set EXPMAP_STREAMS = 1 (or call torch.cuda.empty_cache() between the phases).
EXPMAP_STREAMS = 2
EXPMAP_MIN_ITEMS = 1   # batch items a part must have (1: a batch of two shoots as 1 + 1: 1.61-1.71 -> 1.52-1.56 ms at 128^3)
_side_streams = threading.local()   # per host thread: concurrent callers do not serialise on one pair of streams


def _streams_for(device, n):
    pools = getattr(_side_streams, "pools", None)
    if pools is None:
        pools = _side_streams.pools = {}
    pool = pools.setdefault(device, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


def _shoot_forward_split(metric, m0, phiinv, dt, num_steps, v0):
    """`_shoot(..., keep=False)` over sub-batches on side streams; None when the split does not apply."""
    parts = EXPMAP_STREAMS
    B = m0.size(0)
    parts = min(parts, B // EXPMAP_MIN_ITEMS)
    if parts < 2 or not m0.is_cuda or num_steps < 2:
        return None
    m0 = m0.contiguous()
    phiinv = None if phiinv is None else phiinv.contiguous()
    v0 = None if v0 is None else v0.contiguous()
    out = torch.empty_like(m0)
    main = torch.cuda.current_stream(m0.device)
    streams = _streams_for(m0.device, parts)
    bounds = [B * i // parts for i in range(parts + 1)]
    metric.initialize_luts(shape=m0.shape, dtype=m0.dtype, device=m0.device)   # (made once, on the main stream)
    for i, s in enumerate(streams):
        s.wait_stream(main)   # the inputs (and `out`'s memory) are the main stream's
        sl = slice(bounds[i], bounds[i + 1])
        with torch.cuda.stream(s):
            _shoot(metric, m0[sl], None if phiinv is None else phiinv[sl], dt, num_steps, None if v0 is None else v0[sl],
                   False, out=out[sl])
    for s in streams:
        main.wait_stream(s)
    return out


def _shoot_reverse(metric, m0, dt, steps, G):
    """Reverse sweep over the general steps of `_shoot` (see ExpmapFunction).  G = dL/dphi_N.  Returns (d_m0, G) with
    G = dL/d(phi at the start of the first general step) and d_m0 the momentum gradient collected over these steps
    (None when there were none)."""
    d_m0 = torch.zeros_like(m0) if steps else None
    for phi, v, mphi in reversed(steps):
        d_phi, d_v = lagomorph_ext.interp_backward_fused(G, phi, v, -dt, True, addgo=-dt)
        d_m = metric.sharp(d_v)
        del d_v
        d_phi, d_w = lagomorph_ext.jacobian_times_vectorfield_backward(d_m, phi, mphi, True, False, True, True,
                                                                        d_v=d_phi)
        del d_m
        _, G = lagomorph_ext.interp_backward_fused(d_w, m0, phi, 1.0, True, d_u=d_phi, d_I=d_m0)
    return d_m0, G


def _first_step(metric, m0, dt, v0=None, mommask=None):
    """The Euler step from the identity in closed form.  With phi^-1 = id (zero displacement) `Ad_star` returns m0
    itself (the resampling is the identity and the Jacobian of the zero field vanishes: m0 + 0) and
    `compose_disp_vel(0, v, -dt)` returns 0 + (-dt) v: the same bits as `EPDiff_step` (lddmm.py:39-44) produces
    through the Ad_star and compose kernels (up to the sign of exact zeros), for finite m0.  v0: sharp(m0) if the
    caller already has it."""
    m = m0 if mommask is None else m0 * mommask
    if v0 is None or mommask is not None:
        if isinstance(metric, FluidMetric) and isinstance(dt, (int, float)):   # the factor rides in the operator's last kernel: same bits, one pass less
            return metric.sharp(m, out_scale=-dt)
        return metric.sharp(m) * (-dt)
    return v0 * (-dt)


def _fused_expmap_ok(metric, m0, phiinv, mommask, v0):
    grads = m0.requires_grad or (phiinv is not None and phiinv.requires_grad) or (v0 is not None and v0.requires_grad)
    return (USE_FUSED_EXPMAP and mommask is None and isinstance(metric, FluidMetric) and m0.is_cuda
            and m0.size(1) == m0.dim() - 2 and m0.dtype in (torch.float32, torch.float64)
            and (phiinv is None or (phiinv.shape == m0.shape and phiinv.dtype == m0.dtype))
            and all(hasattr(lagomorph_ext, n) for n in ("Ad_star", "compose", "interp_backward_fused"))
            and torch.is_grad_enabled() and grads)


def _forward_split_ok(metric, m0, phiinv, mommask, v0):
    grads = torch.is_grad_enabled() and (m0.requires_grad or (phiinv is not None and phiinv.requires_grad)
                                          or (v0 is not None and v0.requires_grad))
    return (USE_FUSED_EXPMAP and not grads and mommask is None and isinstance(metric, FluidMetric) and m0.is_cuda
            and m0.size(1) == m0.dim() - 2 and m0.dtype in (torch.float32, torch.float64)
            and (phiinv is None or (phiinv.shape == m0.shape and phiinv.dtype == m0.dtype))
            and all(hasattr(lagomorph_ext, n) for n in ("Ad_star", "compose")))


def expmap(metric, m0, T=1.0, num_steps=10, phiinv=None, mommask=None, checkpoints=False, v0=None):
    """Exponential map: returns the displacement of phi^-1 (lddmm.py:73-105).

    Only ``checkpoints=False`` is supported.  The reference's checkpointed branch is dead
    code (``EPDiffStepsFunction`` swaps dt/phiinv, lddmm.py:56,64; the integer branch
    computes step counts and then integrates nothing, lddmm.py:93-95), so it is rejected
    here instead of being reproduced.

    When no ``phiinv`` is given the shoot starts at the identity and its first Euler step is evaluated in closed
    form, phi^-1_1 = -dt sharp(m0) (`_first_step`): the values `EPDiff_step` would return, without its two gather
    kernels (and, in the backward pass, without their two splats and the Jacobian adjoint).  ``v0`` (not in the
    reference): sharp(m0) when the caller has computed it already -- `lddmm_step` needs it for its regulariser."""
    if checkpoints:
        raise NotImplementedError("expmap(checkpoints=...) is broken in the reference and not provided")
    if num_steps <= 0:
        return torch.zeros_like(m0) if phiinv is None else phiinv
    dt = T / num_steps
    if v0 is not None and (phiinv is not None or mommask is not None):
        v0 = None
    if _fused_expmap_ok(metric, m0, phiinv, mommask, v0):
        return ExpmapFunction.apply(metric, m0, phiinv, dt, num_steps, v0)
    if _forward_split_ok(metric, m0, phiinv, mommask, v0):
        with torch.no_grad():
            h = _shoot_forward_split(metric, m0, phiinv, dt, num_steps, v0)
        if h is not None:
            return h
    first = phiinv is None
    if first:
        phiinv = _first_step(metric, m0, dt, v0, mommask)
    for _ in range(num_steps - 1 if first else num_steps):
        phiinv = EPDiff_step(metric, m0, dt, phiinv, mommask=mommask)
    return phiinv


# --------------------------------------------------------------------------- atlas step


USE_FUSED_STEP = True


def _fused_step_ok(I, m, img, metric, integration_steps):
    return (USE_FUSED_STEP and USE_FUSED_EXPMAP and integration_steps >= 1 and isinstance(metric, FluidMetric)
            and m.is_cuda and I.is_cuda and img.is_cuda and m.dtype in (torch.float32, torch.float64)
            and m.dtype == I.dtype == img.dtype and m.size(1) == m.dim() - 2 and m.numel() < 2 ** 31
            and m.is_contiguous() and torch.is_grad_enabled()
            and all(hasattr(lagomorph_ext, n) for n in ("Ad_star", "compose", "interp_backward_fused", "lincomb")))


def _lddmm_step_fused(I, m, img, metric, dataset_size, integration_steps, reg_weight, learning_rate_pose,
                      momentum_preconditioning, whole=None, after_image_backward=None):
    """`lddmm_step` with the momentum side written out by hand: the shoot and its reverse sweep (`_shoot`,
    `_shoot_reverse`), the regulariser <sharp(m), m> and its gradient, and the descent step, with every elementwise
    sum in one pass (`lagomorph_ext.lincomb`).  Autograd is kept for the image side -- interp of the atlas, the
    regrid of multiscale momenta, the loss -- so that I.grad and its hooks (the atlas builder's all-reduce) behave
    as in the plain form.  Same formulas; sums in a different order (rounding only).
    `whole` = (numel, items) of the WHOLE minibatch when (m, img) is a sub-batch of it (`_lddmm_step_split`): the
    normalisers of the loss are the minibatch's (lddmm.py:308-313), so the parts' losses and gradients simply add.
    `after_image_backward`: called once I.grad holds this call's splat (the split records a stream event there)."""
    numel, items = (img.numel(), img.shape[0]) if whole is None else whole
    regrid_momenta = tuple(m.shape[2:]) != tuple(I.shape[2:])
    dt = 1.0 / integration_steps
    with torch.no_grad():
        m = m.detach()
        v = metric.sharp(m)
        h, steps, _ = _shoot(metric, m, None, dt, integration_steps, v, True)
    h.requires_grad_(True)
    hh = regrid(h, shape=I.shape[2:]) if regrid_momenta else h
    img_term = torch.nn.functional.mse_loss(deform.interp(I, hh), img, reduction="sum") / numel
    img_term.backward()
    if after_image_backward is not None:
        after_image_backward()
    with torch.no_grad():
        c = reg_weight / numel
        if regrid_momenta:  # account for downscaling in averaging (lddmm.py:311-312)
            c = c * (I.numel() / v[0, 0, ...].numel())
        reg_term = c * torch.dot(v.reshape(-1), m.reshape(-1))
        loss = img_term.detach() + reg_term
        d_m0, G = _shoot_reverse(metric, m, dt, steps, h.grad.contiguous())
        own = len(steps) > 0   # G came out of the reverse sweep (ours to overwrite), not out of autograd
        del steps
        # gradient with respect to v = sharp(m): -dt G from the first Euler step, c m from the regulariser
        d_v = lagomorph_ext.lincomb([(-dt, G), (c, m)], out=G if own else None)
        d_mK = metric.sharp(d_v)
        del d_v, G
        norm_factor = items / dataset_size
        terms = [(1.0, d_mK), (c, v)] + ([(1.0, d_m0)] if d_m0 is not None else [])
        if momentum_preconditioning:
            p = metric.flat(lagomorph_ext.lincomb(terms, out=d_mK))
            lagomorph_ext.lincomb([(1.0, m), (-learning_rate_pose, p)], out=m)
        else:   # m <- m - lr (d_mK + c v + d_m0)
            lr = learning_rate_pose
            lagomorph_ext.lincomb([(1.0, m)] + [(-lr * a, x) for a, x in terms], out=m)
        return m, (loss * norm_factor), (reg_term * norm_factor)


# LDDMM_STEP_STREAMS = k >= 2 (an OPTION; default 1 = off): the matching step of a minibatch of at least 2 k subjects is
# cut into k contiguous sub-batches that run -- forward, backward, momentum update -- on HIP streams of their own, like
# the forward shoot (EXPMAP_STREAMS above): subjects are independent up to the two sums over the minibatch, the loss and
# the atlas gradient.  Each part splats into an atlas gradient of its own (a private leaf view of I), records an event
# when that splat has run, and the caller's stream adds the parts and hands the sum to I through autograd -- I.grad
# accumulates and its hooks (the atlas builder's asynchronous all-reduce) fire as in the one-stream form, still while the
# parts' reverse sweeps run.  Same formulas and normalisers; the two sums are taken in a different order (rounding only).
# Why it is off (profiles/r05_stream_split.md): a lone `lddmm_step` called back to back gains 4-6 % at 4-8 subjects
# (tools/ab_step_streams.py), but inside LDDMMAtlasBuilder's loop -- a different minibatch every iteration, the image
# update in between -- the same split measures -1 ... -2.5 % at 4 subjects, +1 ... +3 % at 8 and +-1 % at 32 per GPU
# (tools/ab_atlas_streams.py): not a robust win where it would matter.
LDDMM_STEP_STREAMS = 1


def _lddmm_step_split(I, m, img, metric, dataset_size, integration_steps, reg_weight, learning_rate_pose,
                      momentum_preconditioning):
    """`_lddmm_step_fused` over sub-batches on side streams; None when the split does not apply."""
    parts = LDDMM_STEP_STREAMS
    B = m.size(0)
    if parts < 2 or B < 2 * parts or img.size(0) != B:
        return None
    main = torch.cuda.current_stream(m.device)
    streams = _streams_for(m.device, parts)
    bounds = [B * i // parts for i in range(parts + 1)]
    whole = (img.numel(), B)
    metric.initialize_luts(shape=m.shape, dtype=m.dtype, device=m.device)   # (made once, on the main stream)
    want_I = I.requires_grad
    leaves, events, res = [], [], []
    for i, s in enumerate(streams):
        s.wait_stream(main)   # the inputs are the main stream's
        sl = slice(bounds[i], bounds[i + 1])
        with torch.cuda.stream(s):
            Ik = I.detach().requires_grad_(want_I)   # a leaf of its own: the parts' splats do not race on I.grad
            ev = torch.cuda.Event()
            res.append(_lddmm_step_fused(Ik, m[sl], img[sl], metric, dataset_size, integration_steps, reg_weight,
                                         learning_rate_pose, momentum_preconditioning, whole=whole,
                                         after_image_backward=ev.record))
            leaves.append(Ik)
            events.append(ev)
    if want_I:
        # as soon as every part's splat has run (the parts continue with their reverse sweeps): the minibatch's atlas
        # gradient, accumulated into I.grad through autograd so that post-accumulate hooks fire
        for ev in events:
            main.wait_event(ev)
        for Ik in leaves:   # (made on a side stream, read on this one: keep the allocator from handing the memory out
            Ik.grad.record_stream(main)   # again before this stream's reads have run)
        g = leaves[0].grad
        for Ik in leaves[1:]:
            g = g + Ik.grad
        torch.autograd.backward(I, g)
    for s in streams:
        main.wait_stream(s)
    for r in res:
        r[1].record_stream(main)
        r[2].record_stream(main)
    loss, reg = res[0][1], res[0][2]
    for r in res[1:]:
        loss = loss + r[1]
        reg = reg + r[2]
    return m.detach(), loss, reg


def lddmm_step(I, m, img, metric, dataset_size, integration_steps=5, reg_weight=1e2, learning_rate_pose=2e2,
               momentum_preconditioning=False):
    """One matching step of the atlas builder for a minibatch (lddmm.py:300-325).

    I: atlas image (1, 1, *sp) with requires_grad as the caller wishes (its .grad accumulates),
    m: momenta (B, d, *msp) -- updated in place by gradient descent, img: (B, 1, *sp).
    Returns (m, loss, reg_term) with loss/reg already scaled by B / dataset_size, all on device
    (no host synchronisation)."""
    if _fused_step_ok(I, m, img, metric, integration_steps):
        out = _lddmm_step_split(I, m, img, metric, dataset_size, integration_steps, reg_weight, learning_rate_pose,
                                momentum_preconditioning)
        if out is not None:
            return out
        return _lddmm_step_fused(I, m, img, metric, dataset_size, integration_steps, reg_weight, learning_rate_pose,
                                 momentum_preconditioning)
    m.requires_grad_(True)
    if m.grad is not None:
        m.grad.detach_()
        m.grad.zero_()
    regrid_momenta = tuple(m.shape[2:]) != tuple(I.shape[2:])
    v = metric.sharp(m)   # (lddmm.py:309; computed first here: it is also the velocity of the first Euler step)
    h = expmap(metric, m, num_steps=integration_steps, v0=v)
    if regrid_momenta:  # upscale the deformation to apply to the atlas (lddmm.py:306-307; as coded there,
        h = regrid(h, shape=I.shape[2:])  # without displacement=True: values stay in coarse-grid voxels)
    Idef = deform.interp(I, h)
    reg_term = reg_weight * (v * m).sum() / img.numel()
    if regrid_momenta:  # account for downscaling in averaging (lddmm.py:311-312)
        reg_term = reg_term * (I.numel() / v[0, 0, ...].numel())
    loss = torch.nn.functional.mse_loss(Idef, img, reduction="sum") / img.numel() + reg_term
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


def streaming_batch_average(images, batch_size):
    """Mean over the first axis as the reference's `batch_average` computes it for the initial atlas
    (data.py:308-336): minibatch sums in float64 folded into a running average, returned in the images'
    dtype.  On the device the images live on; shape (*images.shape[1:])."""
    avg, seen = None, 0
    for b in range(0, images.shape[0], batch_size):
        img = images[b:b + batch_size]
        sz = img.shape[0]
        avi = img.to(torch.float64).sum(dim=0)
        if avg is None:
            avg = avi / sz
        else:
            avg = avg * (seen / (seen + sz)) + avi / (seen + sz)
        seen += sz
    if images.dtype in (torch.float32, torch.float64):
        avg = avg.to(images.dtype)
    return avg


def shard_indices(n, world_size, rank, shuffle=True, seed=0, epoch=0):
    """The subjects rank `rank` of `world_size` works on, exactly as the reference assigns them (lddmm.py:164-167:
    `DistributedSampler(dataset, num_replicas=world_size, rank=rank)` with its defaults and `set_epoch` never called):
    a seed-0 permutation of range(n), PADDED by wrapping around to a multiple of world_size (so every rank holds the
    same number of subjects, hence of minibatches -- a few subjects are then seen by two ranks), dealt round-robin.
    torch's own sampler is used, so the assignment is the reference's by construction.  shuffle=False gives the
    unshuffled variant of the same padding rule."""
    from torch.utils.data.distributed import DistributedSampler

    sampler = DistributedSampler(range(n), num_replicas=world_size, rank=rank, shuffle=shuffle, seed=seed)
    sampler.set_epoch(epoch)
    return list(iter(sampler))


class LDDMMAtlasBuilder:
    """Batch-sharded atlas building over in-memory volumes (lddmm.py:108-375, compute path only).

    Each rank owns a contiguous shard of the subjects and of their momenta, both resident in HBM
    (the reference parks momenta in pinned host memory and copies them every iteration,
    lddmm.py:236,328,337).  Every rank must hold the same number of minibatches: `from_dataset` builds the shard
    with the reference's DistributedSampler rule (padded, equal-length shards; `shard_indices`).

    Collectives -- the atlas gradient only, as lddmm.py:292-297 prescribes:
      * ONE SUM all-reduce of `I.grad` (1, 1, *image_shape) per image update.  It is issued
        asynchronously from a post-accumulate-grad hook on the atlas: `I.grad` is final as soon as the
        splat of `deform.interp`'s backward has run -- the first node of the backward pass -- so the
        reduction travels over xGMI while the backward continues through the `integration_steps` Euler
        steps of `expmap`, and is waited for just before `image_optimizer.step()`.  (With the
        reference's blocking call it starts only after the whole backward and the momentum update.)
      * one all-reduce of the initial mean image (lddmm.py:196-198);
      * one all-reduce of the stacked per-iteration (loss, reg) pairs per epoch -- the reference
        reduces both scalars every iteration and then calls .item() (lddmm.py:333-341), a host
        synchronisation per minibatch; the per-iteration histories come out identical."""

    def __init__(self, images, batch_size=8, lddmm_steps=1, lddmm_integration_steps=5, image_update_freq=0,
                 reg_weight=1e2, learning_rate_pose=2e2, learning_rate_image=1e4, metric=None, momentum_shape=None,
                 image_shape=None, momentum_preconditioning=False, I0=None, ms=None, world_size=1, rank=0,
                 dataset_size=None, checkpoint_format=None, overlap_allreduce=True, force_collectives=False):
        self.images = images  # this rank's shard: (n_local, 1, *sp) on the device
        self.batch_size = batch_size
        self.lddmm_steps = lddmm_steps
        self.lddmm_integration_steps = lddmm_integration_steps
        self.image_update_freq = image_update_freq
        self.reg_weight = reg_weight
        self.learning_rate_pose = learning_rate_pose
        self.learning_rate_image = learning_rate_image
        self.metric = metric if metric is not None else FluidMetric([0.1, 0, 0.01])  # lddmm.py:213
        self.momentum_preconditioning = momentum_preconditioning
        self.world_size = world_size
        self.rank = rank
        self.checkpoint_format = checkpoint_format
        self.overlap_allreduce = overlap_allreduce
        # the collectives of lddmm.py:196-198,292-297,333-335 are issued when there is more than one rank -- or when
        # `force_collectives` asks for them at world size 1 (a process group must exist): every branch of the N-rank
        # code then runs, over RCCL, on a single GPU (tests/test_gpu_rccl_world1.py, bench.py's `atlas_step_rccl`)
        self.collectives = world_size > 1 or bool(force_collectives)
        self.grad_reduce_op = dist.ReduceOp.SUM   # lddmm.py:294 (the tests put a non-identity operator here at world size 1)
        n_local = images.shape[0]
        self.dataset_size = dataset_size if dataset_size is not None else n_local * world_size
        dim = images.dim() - 2
        # lddmm.py:186-207: mean image (all-reduced and averaged over ranks) unless I0 is given; brought
        # onto `image_shape` by regrid when the shapes differ
        self.image_shape = tuple(image_shape) if image_shape is not None else tuple(images.shape[2:])
        with torch.no_grad():
            if I0 is None:
                I0 = streaming_batch_average(images, batch_size).unsqueeze(0)
                if self.collectives:
                    dist.all_reduce(I0)
                    I0 /= world_size
            else:
                I0 = I0.detach().to(images.device, images.dtype)
                I0 = I0.reshape(1, 1, *I0.shape[-dim:])
            if tuple(I0.shape[2:]) != self.image_shape:
                I0 = regrid(I0.contiguous(), self.image_shape)
            self.I = I0.detach().clone().view(1, 1, *self.image_shape)
        self.I.requires_grad_(True)
        self.image_optimizer = torch.optim.SGD([self.I], lr=learning_rate_image, weight_decay=0)
        self.image_optimizer.zero_grad()
        self.momentum_shape = tuple(momentum_shape) if momentum_shape is not None else self.image_shape
        self.regrid_momenta = self.momentum_shape != self.image_shape
        bsz = [min(batch_size, n_local - b) for b in range(0, n_local, batch_size)]
        if ms is None:  # lddmm.py:229-235
            ms = [torch.zeros((b, dim) + self.momentum_shape, dtype=images.dtype, device=images.device) for b in bsz]
        else:
            ms = [m.detach().to(images.device, images.dtype).contiguous() for m in ms]
            if [int(m.shape[0]) for m in ms] != bsz:
                raise ValueError("ms does not match this rank's minibatches")
        self.ms = ms
        self.image_iters = 0
        self.epoch_losses, self.epoch_reg_terms = [], []
        self.iter_losses, self.iter_reg_terms = [], []
        self._pending_hist = []    # per-epoch (iterations, 2) device tensors not yet moved to the host
        self._epoch = 0
        # async all-reduce plumbing
        self._reduce_now = False   # set by iteration(): the hook should start the reduction
        self._work = None          # outstanding all-reduce of I.grad
        self._hook = self.I.register_post_accumulate_grad_hook(self._on_image_grad)

    @classmethod
    def from_dataset(cls, images, world_size=1, rank=0, device=None, shuffle=True, **kw):
        """The builder of rank `rank` over the WHOLE dataset `images` (n, 1, *sp), sharded as the reference shards it
        (`shard_indices`: the padded DistributedSampler assignment of lddmm.py:164-167).  `dataset_size` is the
        unpadded n, as the reference's `len(dataloader.dataset)` (lddmm.py:320-323).  The rank's subjects are moved to
        `device` (default: where `images` lives); `builder.subject_indices` records which ones they are."""
        n = images.shape[0]
        idx = shard_indices(n, world_size, rank, shuffle=shuffle) if world_size > 1 else list(range(n))
        shard = images[torch.as_tensor(idx, dtype=torch.long, device=images.device)]
        if device is not None:
            shard = shard.to(device)
        kw.setdefault("dataset_size", n)
        b = cls(shard.contiguous(), world_size=world_size, rank=rank, **kw)
        b.subject_indices = idx
        return b

    # ---- atlas gradient all-reduce -------------------------------------------------------

    def _on_image_grad(self, param):
        """Runs inside the backward pass right after `I.grad` received the splat of this minibatch."""
        if self._reduce_now and self.collectives and self._work is None:
            self._work = dist.all_reduce(param.grad, op=self.grad_reduce_op, async_op=True)

    def _will_update(self, last_of_epoch):
        # the decision update_base_image() is going to take after this iteration (lddmm.py:287-291)
        return self.image_iters + 1 >= self.image_update_freq or last_of_epoch

    def update_base_image(self, force=False):
        """lddmm.py:287-298"""
        if (self.image_iters < self.image_update_freq and not force) or self.image_iters == 0:
            return
        with torch.no_grad():
            if self.collectives:
                if self._work is not None:
                    self._work.wait()
                    self._work = None
                else:
                    dist.all_reduce(self.I.grad, op=self.grad_reduce_op)
            self.I.grad = self.I.grad / (self.image_iters * self.world_size)
            self.image_optimizer.step()
            self.image_optimizer.zero_grad()
        self.image_iters = 0

    # ---- one minibatch ---------------------------------------------------------------------

    def iteration(self, b, last_of_epoch=False):
        """lddmm.py:327-341.  Returns (loss, reg_term) of this rank's minibatch as 0-dim device
        tensors (they are reduced over ranks once per epoch)."""
        img = self.images[b * self.batch_size:(b + 1) * self.batch_size]
        m = self.ms[b]
        for lit in range(self.lddmm_steps):
            last = lit == self.lddmm_steps - 1
            self.I.requires_grad_(last)  # lddmm.py:331-332: the image gradient comes from the last step only
            self._reduce_now = last and self.overlap_allreduce and self._will_update(last_of_epoch)
            m, loss, reg = lddmm_step(self.I, m, img, self.metric, self.dataset_size,
                                      integration_steps=self.lddmm_integration_steps, reg_weight=self.reg_weight,
                                      learning_rate_pose=self.learning_rate_pose,
                                      momentum_preconditioning=self.momentum_preconditioning)
        self._reduce_now = False
        self.ms[b] = m
        self.image_iters += 1
        self.update_base_image()
        return loss, reg

    def epoch(self):
        """lddmm.py:343-362; returns (epoch_loss, epoch_reg_term) as 0-dim device tensors and appends
        the per-iteration values (reduced over ranks) to iter_losses / iter_reg_terms."""
        if self.image_update_freq == 0:
            self.image_optimizer.zero_grad()
        self.image_iters = 0
        nb = len(self.ms)
        hist = torch.zeros((max(nb, 1), 2), dtype=self.images.dtype, device=self.images.device)
        for b in range(nb):
            loss, reg = self.iteration(b, last_of_epoch=b == nb - 1)
            hist[b, 0] = loss
            hist[b, 1] = reg
        self.update_base_image(force=True)
        if self.collectives:
            dist.all_reduce(hist)
        self._pending_hist.append(hist[:nb])
        tot = hist[:nb].sum(dim=0)
        if self.checkpoint_format is not None:
            self.save(self.checkpoint_format.format(epoch=self._epoch))
        return tot[0], tot[1]

    def _flush_history(self):
        """Move the device-side histories to host floats (one synchronisation)."""
        if self._pending_hist:
            h = torch.cat(self._pending_hist, dim=0).cpu().tolist()
            self.iter_losses.extend(x[0] for x in h)
            self.iter_reg_terms.extend(x[1] for x in h)
            self._pending_hist = []
        self.epoch_losses = [float(x) for x in self.epoch_losses]
        self.epoch_reg_terms = [float(x) for x in self.epoch_reg_terms]

    def run(self, num_epochs=1):
        """lddmm.py:364-375.  Losses stay on the device during the run and are converted once at the end."""
        self.image_optimizer.zero_grad()
        for _ in range(num_epochs):
            l, r = self.epoch()
            self.epoch_losses.append(l)
            self.epoch_reg_terms.append(r)
            self._epoch += 1
        self._flush_history()
        return self.I.detach()

    # ---- checkpoint / resume: the reference's HDF5 layout (lddmm.py:238-285) ---------------

    def state_dict(self):
        """What `LDDMMAtlasBuilder.save` of the reference writes (lddmm.py:238-262), as numpy arrays under
        the reference's dataset names: `atlas`, `momenta` (this rank's minibatches concatenated along the
        batch axis; in the momenta's own dtype -- the reference always narrows to float32, which would make a
        float64 run resume inexactly) with its `batch_sizes` attribute, `epoch_losses`, `epoch_reg_terms`, `iter_losses`,
        `iter_reg_terms`."""
        self._flush_history()
        return {
            "atlas": self.I.detach().cpu().numpy(),
            "momenta": torch.cat([m.detach() for m in self.ms], dim=0).cpu().numpy(),
            "batch_sizes": [int(m.shape[0]) for m in self.ms],
            "epoch_losses": np.asarray(self.epoch_losses, dtype=np.float64),
            "epoch_reg_terms": np.asarray(self.epoch_reg_terms, dtype=np.float64),
            "iter_losses": np.asarray(self.iter_losses, dtype=np.float64),
            "iter_reg_terms": np.asarray(self.iter_reg_terms, dtype=np.float64),
        }

    def load_state_dict(self, state, load_image=True, load_momenta=True, load_losses=True):
        """Resume from `state_dict()` (reference: `load`, lddmm.py:264-285)."""
        with torch.no_grad():
            if load_image:
                atlas = torch.as_tensor(np.asarray(state["atlas"])).to(self.I.device, self.I.dtype)
                self.I.copy_(atlas.view_as(self.I))
            if load_momenta:
                szs = [int(s) for s in state["batch_sizes"]]
                if szs != [int(m.shape[0]) for m in self.ms]:
                    raise ValueError("checkpoint was written with a different shard / batch size")
                mom = np.asarray(state["momenta"])
                i = 0
                for m, s in zip(self.ms, szs):
                    m.detach_().copy_(torch.as_tensor(mom[i:i + s]).to(m.device, m.dtype))
                    i += s
        if load_losses:
            self.epoch_losses = [float(x) for x in state["epoch_losses"]]
            self.epoch_reg_terms = [float(x) for x in state["epoch_reg_terms"]]
            self.iter_losses = [float(x) for x in state["iter_losses"]]
            self.iter_reg_terms = [float(x) for x in state["iter_reg_terms"]]
        self.image_iters = 0
        self.image_optimizer.zero_grad()

    def _rank_path(self, filename):
        # every rank owns different momenta: rank r > 0 writes next to rank 0's file
        return filename if self.rank == 0 else f"{filename}.rank{self.rank}"

    def save(self, filename):
        """HDF5 with the reference's dataset names when h5py is importable (lddmm.py:251-262); otherwise
        the same arrays as an uncompressed .npz archive (numpy arrays only: read back without pickle).
        Rank r > 0 writes `<filename>.rank<r>`."""
        st = self.state_dict()
        path = self._rank_path(filename)
        try:
            import h5py
        except ImportError:
            with open(path, "wb") as fh:
                np.savez(fh, **{k: np.asarray(v) for k, v in st.items()})
            return path
        with h5py.File(path, "w") as f:
            f.create_dataset("atlas", data=st["atlas"])
            hms = f.create_dataset("momenta", shape=st["momenta"].shape, dtype=st["momenta"].dtype)
            hms[...] = st["momenta"]
            hms.attrs["batch_sizes"] = st["batch_sizes"]
            for k in ("epoch_losses", "epoch_reg_terms", "iter_losses", "iter_reg_terms"):
                f.create_dataset(k, data=st[k])
        return path

    def load(self, filename, load_image=True, load_momenta=True, load_losses=True):
        path = self._rank_path(filename)
        try:
            import h5py
        except ImportError:
            h5py = None
        is_h5 = False
        with open(path, "rb") as fh:
            is_h5 = fh.read(8) == b"\x89HDF\r\n\x1a\n"
        if is_h5:
            if h5py is None:
                raise RuntimeError(f"{path} is an HDF5 checkpoint but h5py is not importable")
            with h5py.File(path, "r") as f:
                st = {k: np.asarray(f[k]) for k in ("atlas", "momenta", "epoch_losses", "epoch_reg_terms",
                                                    "iter_losses", "iter_reg_terms")}
                st["batch_sizes"] = [int(s) for s in f["momenta"].attrs["batch_sizes"]]
        else:
            with np.load(path, allow_pickle=False) as z:
                st = {k: z[k] for k in z.files}
        self.load_state_dict(st, load_image=load_image, load_momenta=load_momenta, load_losses=load_losses)
