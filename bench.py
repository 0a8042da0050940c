#!/usr/bin/env python3
"""Benchmark of the LDDMM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process only parses its arguments and starts N
fresh worker processes (torch.distributed.run, one per GPU, rendezvous on 127.0.0.1) BEFORE anything
touches a GPU; it never initialises HIP itself.  Started by a launcher (WORLD_SIZE set), it is one of
the workers and refuses to run when WORLD_SIZE != --gpus.

Headline (BASELINE.json `metric`): LDDMM step voxels/sec, 3D 128^3 -- one "step" of this
benchmark is one `lddmm.expmap` call (10 Euler steps of the integrated EPDiff equation,
BASELINE configs[3]) over a GLOBAL batch of 32 momentum fields of 3x128^3 fp32, sharded 32/N per
GPU ("strong" scaling: the global batch is fixed), inputs resident in HBM, no data-path collective.
value = (voxels * Euler steps) processed by all ranks / wall time (max over ranks).

Also on the same JSON line:
  atlas_step    -- BASELINE configs[4], at every N: `LDDMMAtlasBuilder.iteration` (lddmm_step = 5-step expmap
                   -> interp -> loss -> backward through every operator -> momentum update, then ONE RCCL
                   all-reduce of the atlas gradient and the image update) on 160^3 subjects, global
                   minibatch 32 split N ways; voxels/s, ms per step, the all-reduce time, rank count.
  atlas_epoch   -- BASELINE configs[4] as the workload it names: 256 synthetic 160^3 subjects resident in HBM with their
                   momenta, global minibatch 32, one `LDDMMAtlasBuilder.epoch()` (8 minibatches, the forced end-of-epoch
                   image update, the history reduction); ms per epoch, voxels/s, fraction of the HBM bound, peak memory.
                   At N = 1 both atlas legs run the N-rank code over a world-size-1 RCCL process group created in this
                   process (`rccl_world_size_1`; LAGO_BENCH_FORCE_DIST=0 switches that off).
  roofline      -- the dominant hand-written kernel of the workload: algorithmic bytes / mean launch time measured
                   live with HIP events on the launch stream, in a SINGLE-STREAM pass of the same shoots that follows
                   the timed region in this process (the timed region runs the product's default, two sub-batches on two
                   HIP streams, where a launch's duration is not the kernel's own time); PMC traffic from profiles/.
  cpu_baseline  -- the CPU oracle (C port, OpenMP) on bounded samples of the same workloads, timed on
                   this host (rank 0, N = 1 only): the expmap sample plus per-operator figures; and
                   reference_cpu_path: the reference's OWN CPU code (extension/cpu/affine.cpp, oracle/_ref) timed
                   beside the HIP kernel that replaces it, BASELINE configs[0] included.
  interp_splat, fluid, other_ops -- BASELINE configs[1] / configs[2] micro-measurements (N = 1 only).
  brain_grid -- the shoot and the metric on a 176 x 208 x 176 volume (not a BASELINE config; N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def gaussian_blur(x, sigma):
    """Periodic Gaussian blur over the spatial axes via FFT (synthetic-data helper)."""
    dims = tuple(range(2, x.dim()))
    F = torch.fft.rfftn(x, dim=dims)
    for ax, d in enumerate(dims):
        n = x.shape[d]
        k = torch.fft.rfftfreq(n, device=x.device) if d == dims[-1] else torch.fft.fftfreq(n, device=x.device)
        g = torch.exp(-2.0 * (torch.pi * k * sigma) ** 2)
        shape = [1] * F.dim()
        shape[d] = g.numel()
        F = F * g.view(shape)
    return torch.fft.irfftn(F, s=[x.shape[d] for d in dims], dim=dims)


class KernelTimer:
    """Wraps lagomorph_ext entry points with HIP events recorded on torch's current stream --
    the stream the C ABI launches on -- to get per-launch device time inside the timed region."""

    def __init__(self, ext, names):
        self.ext = ext
        self.names = names
        self.orig = {}
        self.events = {n: [] for n in names}
        self.enabled = False

    def __enter__(self):
        for n in self.names:
            f = getattr(self.ext, n)
            self.orig[n] = f

            def wrapped(*a, _f=f, _n=n, **k):
                if not self.enabled:
                    return _f(*a, **k)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = _f(*a, **k)
                e.record()
                self.events[_n].append((s, e))
                return r

            setattr(self.ext, n, wrapped)
        return self

    def __exit__(self, *exc):
        for n, f in self.orig.items():
            setattr(self.ext, n, f)

    def summary(self):
        out = {}
        for n, evs in self.events.items():
            if evs:
                ms = [s.elapsed_time(e) for s, e in evs]
                out[n] = {"launches": len(ms), "mean_ms": sum(ms) / len(ms), "total_ms": sum(ms)}
        return out


def time_op(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)
    return ms[len(ms) // 2], sum(ms) / len(ms)


def micro_interp_splat(ext, dev, size, batch=8):
    """BASELINE configs[1]: 3D deform.interp + splat, batch 8, 1 x size^3 fp32."""
    g = torch.Generator(device=dev).manual_seed(1234)
    I = gaussian_blur(torch.randn((batch, 1, size, size, size), device=dev, generator=g), 2.0)
    I = I / I.std()
    u = gaussian_blur(torch.randn((batch, 3, size, size, size), device=dev, generator=g), 8.0)
    u = u * (4.0 / u.abs().max())  # smooth case: max |u| = 4 voxels
    go = torch.randn((batch, 1, size, size, size), device=dev, generator=g)
    V = batch * size ** 3
    res = {"workload": f"interp+splat batch {batch} x 1x{size}^3 fp32 (configs[1])"}
    for label, uu in (("smooth", u), ("rough", 2.0 * torch.randn_like(u))):
        # (the first ~30 launches after new allocations run 8 % slower than the steady state: warm up past them)
        fwd_med, _ = time_op(lambda: ext.interp_forward(I, uu, 1.0), reps=30, warm=30)
        r = {"fwd_ms": fwd_med, "fwd_GBps": 20.0 * V / fwd_med / 1e6}
        for mode in (1, 0):
            ext.set_splat_mode(mode)
            # the global-atomics leg is only a comparison figure: few repetitions
            bwd_med, _ = time_op(lambda: ext.interp_backward(go, I, uu, 1.0, True, True),
                                 reps=30 if mode == 1 else 4, warm=(30 if label == "smooth" else 5) if mode == 1 else 1)
            tag = "lds" if mode == 1 else "atomics"
            r[f"bwd_{tag}_ms"] = bwd_med
            r[f"bwd_{tag}_GBps"] = 36.0 * V / bwd_med / 1e6
        ext.set_splat_mode(1)
        pair = r["fwd_ms"] + min(r["bwd_lds_ms"], r["bwd_atomics_ms"])
        r["pair_ms"] = pair
        r["pair_GBps"] = 56.0 * V / pair / 1e6
        r["pair_frac_of_hbm_peak"] = r["pair_GBps"] / HBM_PEAK_GBPS
        r["pair_Gvoxel_per_s"] = V / pair / 1e6
        if label == "smooth":   # north_star's target for this pair, stated on the line
            r["target_frac"] = 0.60
            r["met"] = bool(r["pair_frac_of_hbm_peak"] >= 0.60)
            r["floor_of_scheme"] = 0.55
            r["floor_source"] = "profiles/r05_colour_flush.md"
            r["fwd_frac_of_hbm_peak"] = r["fwd_GBps"] / HBM_PEAK_GBPS
            r["bwd_frac_of_hbm_peak"] = r["bwd_lds_GBps"] / HBM_PEAK_GBPS
        res[label] = r
    # HBM bytes per launch of the two kernels from the PMC passes over tools/run_micro.py (same workload, same batch)
    tpath = next((q for q in (os.path.join(ROOT, "profiles", f) for f in ("r06_traffic_micro.json", "r05_traffic_micro.json", "r04_traffic_micro.json", "r03_traffic_micro.json", "r02_traffic_micro.json"))
                  if os.path.exists(q)), "")
    if tpath and size == 128 and batch == 8:
        t = json.load(open(tpath))
        pick = lambda prefix: next((rec["traffic_bytes"] for name, rec in t.items() if name.startswith(prefix)), None)
        res["traffic"] = {
            "source": f"profiles/{os.path.basename(tpath)} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, tools/run_micro.py)",
            "interp_forward": {"kernel": "interp_fwd3_unroll_kernel<float,false,2,true>", "bytes": pick("lago::interp_fwd3_unroll_kernel<float, false"),
                               "algorithmic_bytes": 20.0 * V},
            "interp_backward": {"kernel": "splat_shear_kernel<1024,true,true,false,0>", "bytes": pick("lago::splat_shear_kernel<1024, true, true, false, 0>"),
                                "algorithmic_bytes": 36.0 * V},
        }
    return res


def micro_fluid(lm, dev, size, batch=8):
    """BASELINE configs[2]: FluidMetric sharp/flat on 3 x size^3 momentum fields, batch 8."""
    m = torch.randn((batch, 3, size, size, size), device=dev)
    met = lm.FluidMetric([0.1, 0.0, 0.01])
    V = batch * size ** 3
    with torch.no_grad():
        sharp_ms, _ = time_op(lambda: met.sharp(m), reps=20, warm=20)
        flat_ms, _ = time_op(lambda: met.flat(m), reps=20, warm=20)
        Fm = torch.view_as_real(torch.fft.rfftn(m, dim=(-3, -2, -1), norm="ortho").contiguous())
        k_inv, _ = time_op(lambda: lm.lagomorph_ext.fluid_operator(Fm, True, met.luts["cos"], met.luts["sin"], *met.params))
        k_fwd, _ = time_op(lambda: lm.lagomorph_ext.fluid_operator(Fm, False, met.luts["cos"], met.luts["sin"], *met.params))
    kbytes = 2 * Fm.numel() * 4
    # "FFT vs finite-difference solver" (configs[2]): `flat` applied in the spatial domain by periodic stencils written with
    # torch.roll (tests/fd_fluid.py: an independent derivation from the symbol of cuda/metric.cu:236-254; valid for flat only,
    # SURVEY section 7) -- a comparator for the timing and a cross-check of the values, not an optimised kernel
    fd = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from fd_fluid import flat_fd

        with torch.no_grad():
            fd_ms, _ = time_op(lambda: flat_fd(m, met.params), reps=3, warm=1)
            dev_ = float((met.flat(m).double() - flat_fd(m.double(), met.params)).abs().max() / flat_fd(m.double(), met.params).abs().max())
        fd = {"flat_finite_difference_torch_ms": fd_ms, "flat_fft_ms": flat_ms, "fft_speedup": fd_ms / flat_ms,
              "max_rel_dev_fft_f32_vs_fd_f64": dev_,
              "note": "spatial-domain flat = l(l(v)) with periodic second / central differences through torch.roll"}
    except Exception as e:  # a comparator must not cost the line
        fd = {"error": repr(e)}
    return {
        "finite_difference_comparator": fd,
        "workload": f"FluidMetric sharp/flat batch {batch} x 3x{size}^3 fp32 (configs[2])",
        "sharp_ms": sharp_ms, "flat_ms": flat_ms,
        "sharp_GBps_ideal72.8B": 72.8 * V / sharp_ms / 1e6,
        "kernel_inverse_ms": k_inv, "kernel_forward_ms": k_fwd,
        "kernel_inverse_GBps": kbytes / k_inv / 1e6, "kernel_forward_GBps": kbytes / k_fwd / 1e6,
        "kernel_bytes": kbytes,
    }


def micro_ops(lm, dev, size, batch=8):
    """Every other operator of the path at batch 8 x 3 x size^3 fp32 (smooth displacement): median ms and
    algorithmic GB/s (each tensor counted once)."""
    ext = lm.lagomorph_ext
    g = torch.Generator(device=dev).manual_seed(99)
    sh = (batch, 3, size, size, size)
    v, w, go = (torch.randn(sh, device=dev, generator=g) for _ in range(3))
    u = gaussian_blur(torch.randn(sh, device=dev, generator=g), 8.0)
    u = u * (4.0 / u.abs().max())
    I1 = torch.randn((batch, 1, size, size, size), device=dev, generator=g)
    A = (torch.eye(3, device=dev)[None] + 0.05 * torch.randn((batch, 3, 3), device=dev, generator=g)).contiguous()
    T = torch.randn((batch, 3), device=dev, generator=g)
    V = batch * size ** 3
    half = [size // 2] * 3
    origin = [(size - 1) * 0.5] * 3
    spacing = [(size - 1) / (size // 2 - 1)] * 3
    small = torch.randn((batch, 3, *half), device=dev, generator=g)
    ops = {
        "jtv_forward(disp)": (lambda: ext.jacobian_times_vectorfield_forward(v, w, True, False), 36),
        "jtv_forward(transpose)": (lambda: ext.jacobian_times_vectorfield_forward(v, w, False, True), 36),
        "jtv_backward": (lambda: ext.jacobian_times_vectorfield_backward(go, v, w, True, False, True, True), 60),
        "jtv_adjoint_forward": (lambda: ext.jacobian_times_vectorfield_adjoint_forward(v, w), 36),
        "jtv_adjoint_backward": (lambda: ext.jacobian_times_vectorfield_adjoint_backward(go, v, w, True, True), 60),
        "interp_forward(C=3)": (lambda: ext.interp_forward(v, u, 1.0), 36),
        "interp_backward(C=3)": (lambda: ext.interp_backward(go, v, u, 1.0, True, True), 60),
        "compose": (lambda: ext.compose(u, v, -0.1, 1.0), 36),
        "ad_star(fused interp+jtv)": (lambda: ext.Ad_star(u, w), 36),
        "affine_interp_forward(C=1)": (lambda: ext.affine_interp_forward(I1, A, T), 8),
        "affine_interp_backward(C=1)": (lambda: ext.affine_interp_backward(I1, I1, A, T, True, True, True), 12),
        "regrid_forward(64^3->128^3,C=3)": (lambda: ext.regrid_forward(small, [size] * 3, [(size // 2 - 1) * 0.5] * 3,
                                                                        [(size // 2 - 1) / (size - 1)] * 3), 12 + 1.5),
        "regrid_backward(128^3->64^3,C=3)": (lambda: ext.regrid_backward(v, half, [size] * 3, [(size // 2 - 1) * 0.5] * 3,
                                                                          [(size // 2 - 1) / (size - 1)] * 3), 12 + 1.5),
    }
    out = {"workload": f"batch {batch} x 3x{size}^3 fp32, median of 20 after 20 warm-up calls", "ops": {}}
    for name, (fn, bpv) in ops.items():
        med, _ = time_op(fn, reps=20, warm=20)
        out["ops"][name] = {"ms": med, "alg_bytes_per_voxel": bpv, "GBps": bpv * V / med / 1e6,
                            "frac_of_hbm_peak": bpv * V / med / 1e6 / HBM_PEAK_GBPS}
    # BASELINE configs[0]: 2D affine_interp forward + adjoint, batch 2, 1 x 64 x 64 random images (the reference runs it
    # on its CPU path: cpu_baseline.reference_cpu_path times that code on this host).  32 KB of data: a launch-latency
    # figure, not a bandwidth one -- no roofline fraction is quoted for it.
    I0 = torch.randn((2, 1, 64, 64), device=dev, generator=g)
    A0 = (torch.eye(2, device=dev)[None] + 0.1 * torch.randn((2, 2, 2), device=dev, generator=g)).contiguous()
    T0 = torch.randn((2, 2), device=dev, generator=g)
    go0 = torch.randn((2, 1, 64, 64), device=dev, generator=g)
    f_ms, _ = time_op(lambda: ext.affine_interp_forward(I0, A0, T0), reps=100, warm=20)
    b_ms, _ = time_op(lambda: ext.affine_interp_backward(go0, I0, A0, T0, True, True, True), reps=100, warm=20)
    out["configs0"] = {"workload": "2D affine_interp forward + adjoint (d_I, d_A, d_T), batch 2, 1x64x64 fp32 (BASELINE configs[0])",
                       "forward_us": 1e3 * f_ms, "adjoint_us": 1e3 * b_ms, "pair_us": 1e3 * (f_ms + b_ms),
                       "voxels": 2 * 64 * 64, "pair_voxels_per_s": 2 * 64 * 64 / ((f_ms + b_ms) * 1e-3)}
    return out


def micro_brain_grid(lm, dev, batch=8, shape=(176, 208, 176), steps=10):
    """Not a BASELINE config: the same shoot on a real brain-MRI grid (176 x 208 x 176 = 11*16 x 13*16 x 11*16, the OASIS
    atlas-space volume), whose extents the fluid metric's LDS-tiled FFT passes cover since round 6 (radix-11 / radix-13
    levels); beside it the metric alone, on the default path and through rocFFT's 3D plan + operator kernel."""
    ext = lm.lagomorph_ext
    g = torch.Generator(device=dev).manual_seed(77)
    met = lm.FluidMetric([0.1, 0.0, 0.01])
    vox = batch * shape[0] * shape[1] * shape[2]
    with torch.no_grad():
        m = gaussian_blur(torch.randn((batch, 3) + shape, device=dev, generator=g), 4.0)
        m *= 2.5 / met.sharp(m).abs().max()
        n0 = ext.path_launches("fluid_lds")
        ref = met.sharp(m)
        tuned = ext.path_launches("fluid_lds") == n0 + 1
        t_sharp = time_op(lambda: met.sharp(m), reps=10, warm=3)[0]
        t_shoot = time_op(lambda: lm.expmap(met, m, num_steps=steps), reps=3, warm=1)[0]
        mode = ext.get_tuning()["fluid_mode"]
        try:
            ext.set_fluid_mode(0)
            out = met.sharp(m)
            dev_rocfft = float((out.double() - ref.double()).abs().max() / ref.double().abs().max())
            t_rocfft = time_op(lambda: met.sharp(m), reps=5, warm=2)[0]
        finally:
            ext.set_fluid_mode(mode)
    return {"workload": f"lddmm.expmap, {steps} Euler steps, batch {batch} x 3 x {shape[0]}x{shape[1]}x{shape[2]} fp32 (not a BASELINE config)",
            "expmap_ms": t_shoot, "Gvoxel_step_per_s": vox * steps / t_shoot / 1e6,
            "sharp_ms": t_sharp, "sharp_path": "LDS-tiled passes" if tuned else "generic passes",
            "sharp_frac_of_single_pass_ideal_at_hbm_peak": vox * 72.8 / (t_sharp * 1e-3) / 8.0e12,
            "sharp_ms_rocfft_3d_plus_operator": t_rocfft, "max_rel_dev_from_rocfft": dev_rocfft}


def micro_atlas_step(lm, dev, size, batch=8):
    """One matching step of the atlas builder (lddmm.py:300-325: expmap 5 steps -> interp -> loss ->
    backward through every operator -> momentum update) at batch 8 x size^3, momenta that shoot to
    ~3 voxels of displacement, learning rate 0 so that every timed step sees the same state."""
    metric = lm.FluidMetric([0.1, 0.0, 0.01])
    g = torch.Generator(device=dev).manual_seed(4321)
    I = gaussian_blur(torch.randn((1, 1, size, size, size), device=dev, generator=g), 3.0)
    I = (I / I.std()).requires_grad_(True)
    img = gaussian_blur(torch.randn((batch, 1, size, size, size), device=dev, generator=g), 3.0)
    img = img / img.std()
    with torch.no_grad():
        m = gaussian_blur(torch.randn((batch, 3, size, size, size), device=dev, generator=g), 4.0)
        m *= 3.0 / metric.sharp(m).abs().max()

    def step():
        lm.lddmm_step(I, m, img, metric, dataset_size=batch, integration_steps=5, learning_rate_pose=0.0)

    med, _ = time_op(step, reps=5, warm=2)
    bpv = lddmm_step_alg_bytes_per_voxel(5)
    gbps = bpv * batch * size ** 3 / med / 1e6
    return {"workload": f"lddmm_step (fwd + bwd + update) batch {batch} x {size}^3 fp32, 5 integration steps",
            "ms": med, "Gvoxel_per_s": batch * size ** 3 / med / 1e6, "alg_bytes_per_voxel": bpv,
            "achieved_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS}


# Algorithmic bytes per voxel of one lddmm_step with S integration steps (fp32, d = 3, one image channel; every
# tensor counted once per operator of an ideally fused implementation, SURVEY 8d conventions):
#   forward  per general Euler step: Ad_star 36 + sharp 72.8 + compose 36                              = 144.8
#   backward per general Euler step: compose^T 60 (go, u, v in; d_u, d_v out) + sharp 72.8 + Ad_star^T 60 = 192.8
#   the first Euler step (from the identity, closed form phi_1 = -dt sharp(m0), sharp(m0) shared with the
#         regulariser): 24 forward (v in, phi out), 24 + 36 backward (scale; sum with the regulariser's gradient)  = 84
#   once: interp(I, h) 16 + its backward 28 (atlas broadcast: I and d_I are 1/B of a voxel each) + mse fwd/bwd 8 + 12
#         + reg term: sharp 72.8 forward and backward + <v, m> 24 + momentum update 36
def lddmm_step_alg_bytes_per_voxel(steps):
    return (steps - 1) * (144.8 + 192.8) + 84 + 16 + 28 + 8 + 12 + 2 * 72.8 + 24 + 36


def atlas_leg(lm, dev, world, rank, args, force_dist=False):
    """BASELINE configs[4]: the batched atlas step at `args.atlas_size`^3, global minibatch `args.atlas_batch`
    split over the ranks, one all-reduce of the atlas gradient per image update (image_update_freq = 0:
    every iteration, lddmm.py:287-298).  Timed exactly like the headline: barrier + synchronize on both sides,
    max over ranks.  Learning rates are small so that every timed step sees a comparable state.
    `force_dist` (N = 1 with a world-size-1 RCCL process group): the builder takes every branch of the N-rank code --
    the asynchronous all-reduce from the gradient hook, the wait before the image update -- over RCCL on this one GPU."""
    coll = world > 1 or force_dist
    S, GB = args.atlas_size, args.atlas_batch
    if GB % world:
        raise SystemExit(f"bench.py: --atlas-batch {GB} is not divisible by {world} ranks")
    B = GB // world
    iters = args.atlas_warmup + args.atlas_steps
    g = torch.Generator(device=dev).manual_seed(777)  # the template is the same on every rank
    tmpl = gaussian_blur(torch.randn((1, 1, S, S, S), device=dev, generator=g), 3.0)
    tmpl = tmpl / tmpl.std()
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    subj = []
    with torch.no_grad():
        for _ in range(iters):  # this rank's shard: `iters` minibatches of B subjects = template o (id + smooth u) + noise
            u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
            u *= 3.0 / u.abs().max()
            x = lm.interp(tmpl, u) + 0.05 * torch.randn((B, 1, S, S, S), device=dev, generator=g)
            subj.append(x)
            del u
        images = torch.cat(subj)
        del subj
    builder = lm.LDDMMAtlasBuilder(images, batch_size=B, lddmm_integration_steps=5, reg_weight=1e2,
                                   learning_rate_pose=1e-3, learning_rate_image=1e-2, world_size=world, rank=rank,
                                   dataset_size=GB * iters, force_collectives=force_dist)
    # momenta that shoot to ~3 voxels, so that the gathers and splats see a realistic displacement
    with torch.no_grad():
        for b in range(len(builder.ms)):
            m = gaussian_blur(torch.randn(builder.ms[b].shape, device=dev, generator=g), 4.0)
            m *= 3.0 / builder.metric.sharp(m).abs().max()
            builder.ms[b] = m
    torch.cuda.empty_cache()

    def sync():
        torch.cuda.synchronize()
        if coll:
            dist.barrier()
        torch.cuda.synchronize()

    for b in range(args.atlas_warmup):
        builder.iteration(b)
    sync()
    t0 = time.perf_counter()
    for b in range(args.atlas_warmup, iters):
        builder.iteration(b)
    sync()
    T = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if coll:
        dist.all_reduce(T, op=dist.ReduceOp.MAX)
    T = T.item()
    # the same iterations again with the all-reduce BLOCKING behind the whole backward pass (the reference's placement,
    # lddmm.py:287-298) instead of asynchronous from the gradient hook: what the overlap hides shows as the difference
    T_block = None
    if coll:
        builder.overlap_allreduce = False
        sync()
        t0 = time.perf_counter()
        for b in range(args.atlas_warmup, iters):
            builder.iteration(b)
        sync()
        Tb = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(Tb, op=dist.ReduceOp.MAX)
        T_block = Tb.item()
        builder.overlap_allreduce = True
    # at one rank with forced collectives: the same iterations once more with the collectives switched off (the plain
    # single-process builder), i.e. what the RCCL plumbing itself costs per step
    T_plain = None
    if force_dist and world == 1:
        builder.collectives = False
        sync()
        t0 = time.perf_counter()
        for b in range(args.atlas_warmup, iters):
            builder.iteration(b)
        torch.cuda.synchronize()
        T_plain = time.perf_counter() - t0
        builder.collectives = True
    # the collective on its own: a blocking all-reduce of one (1, 1, S, S, S) fp32 gradient
    ar_ms = None
    if coll:
        buf = torch.zeros((1, 1, S, S, S), device=dev)
        for _ in range(3):
            dist.all_reduce(buf)
        ts = []
        for _ in range(10):
            sync()
            t1 = time.perf_counter()
            dist.all_reduce(buf)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        ar = torch.tensor([sorted(ts)[len(ts) // 2]], device=dev, dtype=torch.float64)
        dist.all_reduce(ar, op=dist.ReduceOp.MAX)
        ar_ms = 1e3 * ar.item()
    vox = GB * S ** 3 * args.atlas_steps
    bpv = lddmm_step_alg_bytes_per_voxel(5)
    gbps = bpv * vox / T / 1e9
    finite = bool(torch.isfinite(builder.I).all().item())
    del builder, images
    torch.cuda.empty_cache()
    return {
        "workload": f"LDDMMAtlasBuilder.iteration (lddmm_step fwd + bwd + momentum update, all-reduce of the atlas "
                    f"gradient, image update), {S}^3 fp32, global minibatch {GB} = {B} per GPU x {world}, 5 integration "
                    f"steps (BASELINE configs[4])",
        "value": vox / T, "unit": "voxels/s", "ms_per_step": 1e3 * T / args.atlas_steps, "steps": args.atlas_steps,
        "warmup": args.atlas_warmup, "n_ranks": world, "backend": dist.get_backend() if coll else None,
        "global_batch": GB, "per_gpu_batch": B, "scaling": "strong",
        "collective": ("%s all_reduce(SUM) of I.grad, %.1f MB fp32, issued asynchronously from the backward pass"
                       % ("RCCL" if dist.get_backend() == "nccl" else dist.get_backend(), 4 * S ** 3 / 1e6)) if coll else None,
        "forced_collectives_at_world_size_1": bool(force_dist and world == 1),
        "note": ("world size 1: every collective of the N-rank builder is issued through ProcessGroupNCCL (its own stream, "
                 "work.wait() = stream wait); an in-place SUM over one rank moves no data, so allreduce_ms is the call's "
                 "cost, not a bandwidth figure") if force_dist and world == 1 else None,
        "ms_per_step_without_collectives": None if T_plain is None else 1e3 * T_plain / args.atlas_steps,
        "allreduce_ms": ar_ms,
        "ms_per_step_blocking_allreduce": None if T_block is None else 1e3 * T_block / args.atlas_steps,
        "allreduce_hidden_ms_per_step": None if T_block is None else 1e3 * (T_block - T) / args.atlas_steps,
        "alg_bytes_per_voxel": bpv, "achieved_GBps": gbps,
        "frac_of_hbm_peak": gbps / (HBM_PEAK_GBPS * world), "finite": finite,
    }


def atlas_epoch_leg(lm, dev, world, rank, args, force_dist=False):
    """BASELINE configs[4] as the workload it names: `args.epoch_subjects` (256) synthetic `args.atlas_size`^3 (160^3)
    subjects resident in HBM with their momenta (4.2 GB + 12.6 GB at one rank), sharded over the ranks, global minibatch
    `args.atlas_batch` (32), and one `LDDMMAtlasBuilder.epoch()` (lddmm.py:343-362): every minibatch's matching step and
    image update (image_update_freq 0, lddmm.py:287-298), the forced end-of-epoch update (:359) and the reduction of the
    per-iteration (loss, reg) history (:333-335).  One warm-up epoch, one timed epoch between synchronised barriers,
    max over ranks."""
    coll = world > 1 or force_dist
    S, GB, NS = args.atlas_size, args.atlas_batch, args.epoch_subjects
    if GB % world or NS % GB:
        raise SystemExit(f"bench.py: --epoch-subjects {NS} / --atlas-batch {GB} do not split over {world} ranks")
    B, nb = GB // world, NS // GB
    torch.cuda.reset_peak_memory_stats(dev)
    g = torch.Generator(device=dev).manual_seed(777)  # the template is the same on every rank
    tmpl = gaussian_blur(torch.randn((1, 1, S, S, S), device=dev, generator=g), 3.0)
    tmpl = tmpl / tmpl.std()
    g = torch.Generator(device=dev).manual_seed(8765 + rank)
    images = torch.empty((nb * B, 1, S, S, S), device=dev)
    with torch.no_grad():
        for b in range(nb):  # this rank's shard: template o (id + smooth u) + noise
            u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
            u *= 3.0 / u.abs().max()
            images[b * B:(b + 1) * B] = lm.interp(tmpl, u) + 0.05 * torch.randn((B, 1, S, S, S), device=dev, generator=g)
            del u
    builder = lm.LDDMMAtlasBuilder(images, batch_size=B, lddmm_integration_steps=5, reg_weight=1e2,
                                   learning_rate_pose=1e-3, learning_rate_image=1e-2, world_size=world, rank=rank,
                                   dataset_size=NS, force_collectives=force_dist)
    with torch.no_grad():  # momenta that shoot to ~3 voxels, so that the gathers and splats see a realistic displacement
        for b in range(len(builder.ms)):
            m = gaussian_blur(torch.randn(builder.ms[b].shape, device=dev, generator=g), 4.0)
            m *= 3.0 / builder.metric.sharp(m).abs().max()
            builder.ms[b] = m
            del m
    torch.cuda.empty_cache()

    def sync():
        torch.cuda.synchronize()
        if coll:
            dist.barrier()
        torch.cuda.synchronize()

    builder.image_optimizer.zero_grad()
    builder.epoch()   # warm-up
    sync()
    t0 = time.perf_counter()
    loss, reg = builder.epoch()
    sync()
    T = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if coll:
        dist.all_reduce(T, op=dist.ReduceOp.MAX)
    T = T.item()
    peak = torch.cuda.max_memory_allocated(dev)
    reserved = torch.cuda.max_memory_reserved(dev)
    builder._flush_history()
    finite = bool(torch.isfinite(builder.I).all().item()) and all(x == x for x in builder.iter_losses)
    vox = NS * S ** 3
    bpv = lddmm_step_alg_bytes_per_voxel(5)
    gbps = bpv * vox / T / 1e9
    out = {
        "workload": f"LDDMMAtlasBuilder.epoch(): {NS} synthetic subjects of {S}^3 fp32 resident in HBM with their momenta, "
                    f"global minibatch {GB} = {B} per GPU x {world}, {nb} minibatches per epoch, 5 integration steps, image "
                    "update (with its all-reduce when there are collectives) after every minibatch, forced end-of-epoch "
                    "update, history reduction (BASELINE configs[4], lagomorph/lddmm.py:343-362)",
        "ms_per_epoch": 1e3 * T, "value": vox / T, "unit": "voxels/s", "subjects": NS, "minibatches": nb,
        "ms_per_minibatch": 1e3 * T / nb, "n_ranks": world, "backend": dist.get_backend() if coll else None,
        "forced_collectives_at_world_size_1": bool(force_dist and world == 1),
        "alg_bytes_per_voxel": bpv, "achieved_GBps": gbps, "frac_of_hbm_peak": gbps / (HBM_PEAK_GBPS * world),
        "resident_GB": {"images": images.numel() * 4 / 1e9, "momenta": sum(m.numel() for m in builder.ms) * 4 / 1e9},
        "peak_allocated_GB": peak / 1e9, "peak_reserved_GB": reserved / 1e9,
        "epoch_loss": float(loss), "epoch_reg_term": float(reg), "finite": finite, "scaling": "strong",
    }
    del builder, images
    torch.cuda.empty_cache()
    return out


def reference_cpu_path(lm, dev, size):
    """The reference's OWN CPU code beside the GPU kernel that replaces it: `affine_interp_cpu_forward`
    of extension/cpu/affine.cpp:129-169, compiled from the reference tree into oracle/_ref/lagomorph_ref_cpu.so by
    oracle/build_ref.py (the prebuilt file travels to the GPU box), timed on this host next to `affine_interp_forward`
    through HIP on the same inputs.  The reference's loop is serial (no OpenMP, no at::parallel_for): one core is all it
    ever uses.  BASELINE configs[0] (2D, batch 2, 1 x 64 x 64) and a bounded 3D sample (2 of the 8 x 1 x size^3 items)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import build_ref

    ref = build_ref.load_ref()
    if ref is None:
        return {"available": False, "note": "oracle/_ref/lagomorph_ref_cpu.so not in the tree (built from /root/reference by __graft_entry__.build())"}
    ext = lm.lagomorph_ext
    g = torch.Generator().manual_seed(5)
    out = {"available": True, "kind": "reference", "cores": 1,
           "source": "lagomorph/extension/cpu/affine.cpp:129-169 (affine_interp_cpu_forward), compiled unmodified; serial code"}
    prev = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        for tag, shape, d, reps in (("configs0_2d_batch2_1x64x64", (2, 1, 64, 64), 2, 200),
                                    (f"3d_batch2_1x{size}^3", (2, 1, size, size, size), 3, 3)):
            I = torch.randn(shape, generator=g)
            A = (torch.eye(d)[None] + 0.1 * torch.randn((shape[0], d, d), generator=g)).contiguous()
            T = torch.randn((shape[0], d), generator=g)
            want = ref.affine_interp_cpu_forward(I, A, T)   # untimed: page in
            t0 = time.perf_counter()
            for _ in range(reps):
                want = ref.affine_interp_cpu_forward(I, A, T)
            cpu_s = (time.perf_counter() - t0) / reps
            Ig, Ag, Tg = I.to(dev), A.to(dev), T.to(dev)
            gpu_ms, _ = time_op(lambda: ext.affine_interp_forward(Ig, Ag, Tg), reps=50, warm=20)
            got = ext.affine_interp_forward(Ig, Ag, Tg).cpu()
            V = I.numel()
            out[tag] = {"voxels": V, "reference_cpu_s": cpu_s, "reference_cpu_voxels_per_s": V / cpu_s,
                        "hip_ms": gpu_ms, "hip_voxels_per_s": V / (gpu_ms * 1e-3),
                        # the CPU path steps its position incrementally along a row, the CUDA path (and the HIP kernel)
                        # evaluates it per voxel: they agree to rounding only, as the reference's own
                        # test_affine_interp_gpucpu_match asserts (allclose)
                        "max_abs_diff_hip_vs_reference_cpu": float((got - want).abs().max()),
                        "max_abs_reference": float(want.abs().max())}
    finally:
        torch.set_num_threads(prev)
    return out


def cpu_baseline(size, euler_steps, sample_batch=1):
    """Times the CPU oracle (test infrastructure) on bounded samples of the benchmark's workloads on the host's
    cores (OpenMP, one thread per host CPU) and on one thread: the headline sample (`sample_batch` volumes of size^3,
    one expmap of `euler_steps` steps) and the four operators of BASELINE.md section 4 (interp, splat, jtv, sharp)
    at batch 8.  The oracle stands in for lagomorph_ext only inside this function."""
    import numpy as np

    import lagomorph_amd as lm
    from oracle.lago_oracle import OracleExt

    from lagomorph_amd import metric as lmm

    o = OracleExt()
    names = ["interp_forward", "jacobian_times_vectorfield_forward", "fluid_operator", "compose"]
    saved = {n: getattr(lm.lagomorph_ext, n) for n in names}
    from lagomorph_amd import adjrep as lma

    fused_flag, fused_ad = lmm.USE_FUSED_FLUID, lma.USE_FUSED_AD_STAR
    try:
        lmm.USE_FUSED_FLUID = False  # the oracle has the reference's three-call form only
        lma.USE_FUSED_AD_STAR = False  # ... and Ad_star as interp + jacobian_times_vectorfield
        for n in names:
            setattr(lm.lagomorph_ext, n, getattr(o, n))
        import oracle.lago_oracle as orc

        rng = np.random.default_rng(7)
        met = lm.FluidMetric([0.1, 0.0, 0.01])
        runs = {}
        ncpu = max(1, os.cpu_count() or 1)
        prev_threads = torch.get_num_threads()

        def shoot(threads, batch, steps):
            m = torch.from_numpy((0.01 * rng.standard_normal((batch, 3, size, size, size))).astype(np.float32))
            orc.set_threads(threads)
            torch.set_num_threads(threads)
            with torch.no_grad():  # untimed: thread pool start-up, FFT plans, LUTs
                lm.expmap(met, m[:1], num_steps=1)
            t0 = time.perf_counter()
            with torch.no_grad():
                lm.expmap(met, m, num_steps=steps)
            return time.perf_counter() - t0

        # thread count: every host CPU is not the fastest setting for memory-bound OpenMP loops plus pocketfft (on the
        # 256-CPU boxes of this pool 256 threads run the sample 2.4 x SLOWER than 64), so a short calibration (one
        # Euler step on a quarter of the sample) picks the best of {all, 1/2, 1/4, 1/8 of the CPUs} and the full
        # sample is timed with that; every candidate's calibration rate is reported
        cands = sorted({max(1, ncpu // d) for d in (1, 2, 4, 8)}, reverse=True)
        calib = {}
        for th in cands:
            cb = max(1, sample_batch // 4)
            calib[th] = cb * size ** 3 / shoot(th, cb, 1)
        nthreads = max(calib, key=calib.get)
        for tag, threads, batch in (("all", nthreads, sample_batch), ("one", 1, max(1, sample_batch // 4))):
            dt = shoot(threads, batch, euler_steps)
            runs[tag] = (batch * size ** 3 * euler_steps / dt, dt, batch, threads)
        # where the all-threads sample spends its time: the FFTs (torch CPU / pocketfft) against the oracle's kernels
        orc.set_threads(nthreads)
        torch.set_num_threads(nthreads)
        mm = torch.from_numpy((0.01 * rng.standard_normal((max(1, sample_batch // 4), 3, size, size, size))).astype(np.float32))
        with torch.no_grad():
            met.sharp(mm[:1])   # untimed: plans
            t0 = time.perf_counter()
            F = torch.fft.rfftn(mm, dim=(-3, -2, -1), norm="ortho")
            torch.fft.irfftn(F, s=mm.shape[-3:], dim=(-3, -2, -1), norm="ortho")
            t_fft = time.perf_counter() - t0
            t0 = time.perf_counter()
            met.sharp(mm)
            t_sharp = time.perf_counter() - t0
        fft_share = {"fft_pair_s": t_fft, "sharp_s": t_sharp, "fft_share_of_sharp": t_fft / t_sharp,
                     "sample": f"batch {mm.shape[0]} x 3x{size}^3"}
        del mm, F
        # per-operator figures, batch 8 x size^3: voxels/s and algorithmic GB/s
        nb = 8
        V = nb * size ** 3
        I1 = rng.standard_normal((nb, 1, size, size, size)).astype(np.float32)
        u3 = (2.0 * rng.standard_normal((nb, 3, size, size, size))).astype(np.float32)
        w3 = rng.standard_normal((nb, 3, size, size, size)).astype(np.float32)
        w3t = torch.from_numpy(w3)
        per_op = {}
        cases = {
            "interp_forward(C=1)": (lambda: orc.interp_forward(I1, u3, 1.0), 20),
            "interp_backward(C=1, splat + d_u)": (lambda: orc.interp_backward(I1, I1, u3, 1.0, True, True), 36),
            "jtv_forward(C=3)": (lambda: orc.jacobian_times_vectorfield_forward(u3, w3, True, False), 36),
            "sharp(3x)": (lambda: met.sharp(w3t), 72.8),
        }
        for tag, threads in (("all", nthreads), ("one", 1)):
            orc.set_threads(threads)
            torch.set_num_threads(threads)
            for name, (fn, bpv) in cases.items():
                with torch.no_grad():
                    t0 = time.perf_counter()
                    fn()
                    dt = time.perf_counter() - t0
                per_op.setdefault(name, {"alg_bytes_per_voxel": bpv})[tag] = {
                    "threads": threads, "s": dt, "voxels_per_s": V / dt, "GBps": bpv * V / dt / 1e9}
    finally:
        try:
            orc.set_threads(1)
            torch.set_num_threads(prev_threads)
        except Exception:
            pass
        lmm.USE_FUSED_FLUID = fused_flag
        lma.USE_FUSED_AD_STAR = fused_ad
        for n, f in saved.items():
            setattr(lm.lagomorph_ext, n, f)
    v_all, dt_all, b_all, th_all = runs["all"]
    v_one, dt_one, b_one, _ = runs["one"]
    return {
        "value": v_all, "unit": "voxels/s", "cores": th_all, "kind": "port",
        "sample": f"expmap {euler_steps} Euler steps, batch {b_all} x 3x{size}^3 fp32, oracle C port with OpenMP over "
                  f"the output voxels ({th_all} threads; FFTs by torch CPU/pocketfft), {dt_all:.1f} s",
        "one_thread": {"value": v_one, "cores": 1,
                       "sample": f"same, batch {b_one}, 1 thread, {dt_one:.1f} s"},
        "per_op": {"sample": f"one call each at batch {nb} x {size}^3 fp32 (rough displacement, 2 voxels rms)", "ops": per_op},
        "host_cpus": os.cpu_count(),
        "thread_calibration_voxel_steps_per_s": {str(k): v for k, v in calib.items()},
        "pocketfft": fft_share,
    }


def ensure_built(local_rank):
    """A fresh checkout carries no binaries: local rank 0 builds the HIP library and the oracle (what
    __graft_entry__.build() does), the other ranks of the node wait for the files."""
    lib = os.path.join(ROOT, "lagomorph_amd", "_lib", "liblagomorph_hip.so")
    orc = os.path.join(ROOT, "oracle", "_build", "liblago_oracle.so")
    if os.path.exists(lib) and os.path.exists(orc):
        return
    if local_rank == 0:
        import contextlib

        import __graft_entry__ as ge

        with contextlib.redirect_stdout(sys.stderr):  # stdout carries the one JSON line only
            ge.build()
        return
    t0 = time.time()
    while not (os.path.exists(lib) and os.path.exists(orc)):
        if time.time() - t0 > 1200:
            raise SystemExit("bench.py: timed out waiting for local rank 0 to build the HIP library")
        time.sleep(2.0)
    time.sleep(2.0)  # let the linker finish writing


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="momentum fields over ALL GPUs (sharded batch/N per GPU)")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--euler-steps", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-micro", action="store_true")
    ap.add_argument("--no-atlas", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the two untimed comparison shoots (all steps through the general kernels; stream split): "
                         "the profile runs use it, so that a kernel's average in the rocprofv3 summary is over the "
                         "launches of the timed workload only")
    ap.add_argument("--streams", type=int, default=0,
                    help="lddmm.EXPMAP_STREAMS for the timed region (0 = the product's default, 2).  The profile runs pass 1: "
                         "every launch of the trace / PMC passes is then a whole-batch launch on one stream, the launch the "
                         "roofline object describes")
    ap.add_argument("--atlas-size", type=int, default=160)
    ap.add_argument("--atlas-batch", type=int, default=32, help="subjects per atlas update over ALL GPUs")
    ap.add_argument("--atlas-steps", type=int, default=4)
    ap.add_argument("--atlas-warmup", type=int, default=1)
    ap.add_argument("--epoch-subjects", type=int, default=256, help="subjects of the atlas_epoch leg over ALL GPUs (configs[4]: 256)")
    ap.add_argument("--no-epoch", action="store_true", help="skip the atlas_epoch leg")
    ap.add_argument("--cpu-sample-batch", type=int, default=16,
                    help="volumes in the all-cores CPU baseline sample (the one-thread sample is a quarter of it)")
    return ap.parse_args()


# LAGO_BENCH_SHARE_GPU=1: run the N-rank code path on ONE GPU (ranks share device 0, gloo instead of RCCL) -- for
# checking the multi-rank plumbing on a single-GPU box; the JSON line is marked and is not a measurement.
SHARE_GPU = os.environ.get("LAGO_BENCH_SHARE_GPU", "") == "1"


def spawn_workers(args):
    """Parent of an N-GPU run started without a launcher: start N fresh workers, one per GPU, and hand their output
    through.  The parent makes NO GPU call at all -- not even a device count, which on ROCm may fall back to
    hipGetDeviceCount and initialise the runtime; each worker validates its own device instead
    (torch.cuda.set_device(local_rank) fails on a box with fewer GPUs).  Returns the exit code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# At N = 1 the atlas legs run the N-rank code over a world-size-1 RCCL process group created in this process (VERDICT r5
# item 1): init_process_group("nccl") as lagomorph/utils.py:161-166 does, every collective of LDDMMAtlasBuilder issued
# through ProcessGroupNCCL.  LAGO_BENCH_FORCE_DIST=0 switches it off (the atlas legs then run without a process group).
FORCE_DIST = os.environ.get("LAGO_BENCH_FORCE_DIST", "1") != "0"

_RCCL_PROBE = r"""
import socket, sys, torch, torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
t = torch.ones(1 << 20, device=dev)
w = dist.all_reduce(t, async_op=True); w.wait()
torch.cuda.synchronize()
assert float(t.sum()) == float(1 << 20)
dist.barrier(); dist.destroy_process_group()
print("RCCL_WORLD1_OK")
"""


def rccl_world1_probe(timeout=180):
    """Can a world-size-1 RCCL process group be created and used on this box?  Asked of a CHILD process with a time
    limit, before this process touches the GPU: a communicator that hangs in its set-up then costs the atlas legs their
    RCCL variant, not the whole benchmark line.  Returns (ok, detail)."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        r = subprocess.run([sys.executable, "-c", _RCCL_PROBE], capture_output=True, text=True, timeout=timeout, env=env)
    except subprocess.TimeoutExpired:
        return False, f"probe timed out after {timeout} s"
    if r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout:
        return True, "ok"
    return False, (r.stderr or r.stdout).strip()[-400:]


def free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


_JSON_FD = None


def claim_stdout():
    """stdout carries ONE JSON line: everything else that writes to file descriptor 1 -- RCCL prints a version banner
    there from C -- is sent to stderr; the line itself goes to the saved descriptor (`emit`)."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    data = (line + "\n").encode()
    fd = _JSON_FD if _JSON_FD is not None else 1
    while data:
        data = data[os.write(fd, data):]


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_workers(args))
    claim_stdout()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a run on the wrong "
                         "number of ranks")
    if args.batch % world:
        raise SystemExit(f"bench.py: --batch {args.batch} is not divisible by {world} ranks")
    force_dist, force_dist_detail = False, None
    if world == 1 and FORCE_DIST and not (args.no_atlas and args.no_epoch):
        force_dist, force_dist_detail = rccl_world1_probe()   # (a child process; nothing here has touched the GPU yet)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    ensure_built(local_rank)
    if SHARE_GPU:  # plumbing check only: all ranks on GPU 0, collectives over gloo (RCCL refuses two ranks per device)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if SHARE_GPU:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    elif force_dist:
        try:
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1, device_id=dev)
        except Exception as e:
            force_dist, force_dist_detail = False, repr(e)

    import lagomorph_amd as lm

    ext = lm.lagomorph_ext
    GBATCH, S, E = args.batch, args.size, args.euler_steps
    B = GBATCH // world
    torch.manual_seed(1234 + rank)
    metric = lm.FluidMetric([0.1, 0.0, 0.01])  # the atlas builder's default, lddmm.py:213
    with torch.no_grad():
        m = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev), 4.0)
        v = metric.sharp(m)
        m *= 5.0 / v.abs().max()  # max |expmap| ~ 5 voxels
        del v

        def step():
            return lm.expmap(metric, m, num_steps=E)

        names = ["interp_forward", "jacobian_times_vectorfield_forward", "fluid_operator", "fluid_metric", "compose",
                 "Ad_star"]
        from lagomorph_amd import lddmm as _lddmm
        if args.streams > 0:
            _lddmm.EXPMAP_STREAMS = args.streams
        default_streams = _lddmm.EXPMAP_STREAMS
        # TIMED REGION: the product's default path -- a forward-only shoot of 2+ batch items is cut into two sub-batches
        # on HIP streams of their own (lddmm.EXPMAP_STREAMS = 2, bit-identical to one stream).  HIP events of the
        # wrapped entry points are recorded on the stream each call is launched on; with two parts in flight they
        # measure a kernel BESIDE the other part's kernels (reported as breakdown_ms_per_step, not as a roofline).
        with KernelTimer(ext, names) as kt:
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            kt.enabled = True
            t0 = time.perf_counter()
            for _ in range(args.steps):
                h = step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            kt.enabled = False
            ksum_timed = kt.summary()
        hmax = h.abs().max().item()
        # ROOFLINE PASS (same process, same inputs, right after the timed region): the same shoots on
        # ONE stream, so that a launch's HIP-event duration is the time the kernel needs by itself.  `roofline` and
        # `single_stream` below come from here; the headline does not.
        _lddmm.EXPMAP_STREAMS = 1
        try:
            with KernelTimer(ext, names) as kt1:
                step()
                torch.cuda.synchronize()
                kt1.enabled = True
                ts0 = time.perf_counter()
                for _ in range(args.steps):
                    h1 = step()
                torch.cuda.synchronize()
                t_single = (time.perf_counter() - ts0) / args.steps
                kt1.enabled = False
                ksum = kt1.summary()
            single_bits = bool(torch.equal(h1, h))
            del h1
            # for comparison, untimed: the same single-stream shoot with every Euler step through the general kernels
            # (an explicit zero phiinv switches off the closed-form first step; same result bit for bit)
            t_general, same_bits = None, None
            if not args.no_extras:
                z = torch.zeros_like(m)
                lm.expmap(metric, m, num_steps=E, phiinv=z)
                torch.cuda.synchronize()
                tg0 = time.perf_counter()
                for _ in range(min(args.steps, 3)):
                    hg = lm.expmap(metric, m, num_steps=E, phiinv=z)
                torch.cuda.synchronize()
                t_general = (time.perf_counter() - tg0) / min(args.steps, 3)
                same_bits = bool(torch.equal(hg, h))
                del hg, z
        finally:
            _lddmm.EXPMAP_STREAMS = default_streams
        split_active = default_streams >= 2 and B // _lddmm.EXPMAP_MIN_ITEMS >= default_streams   # (lddmm._shoot_forward_split)
        del h
    elapsed = torch.tensor([t1 - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    T = elapsed.item()
    vox_steps = GBATCH * S ** 3 * E * args.steps
    V = B * S ** 3

    result = {
        "metric": "LDDMM step voxels/sec (3D, 128^3)",
        "value": vox_steps / T,
        "unit": "voxels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * T / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"lddmm.expmap, {E} Euler steps, global batch {GBATCH} x 3x{S}^3 fp32 sharded {B} per GPU "
                        "(BASELINE configs[3]); value counts voxels x Euler steps; shooting from the identity, the "
                        "first Euler step is evaluated in closed form (-dt sharp(m0): the bits EPDiff_step returns); "
                        + (f"the product's default path: each rank's shoot runs as {default_streams} sub-batches on HIP "
                           "streams of their own (lddmm.EXPMAP_STREAMS, bit-identical)" if split_active else
                           f"one HIP stream (the sub-batch split needs {max(2, 2 * _lddmm.EXPMAP_MIN_ITEMS)}+ items per rank "
                           "and lddmm.EXPMAP_STREAMS >= 2)"),
            "streams": default_streams if split_active else 1,
            "single_stream": {"ms_per_step": 1e3 * t_single, "same_bits": single_bits,
                              "note": "the same shoots with lddmm.EXPMAP_STREAMS = 1, after the timed region: the pass "
                                      "the roofline object's launch durations are measured in"},
            "all_steps_through_the_general_kernels": None if t_general is None else {
                "ms_per_step": 1e3 * t_general, "local_voxel_steps_per_s": B * S ** 3 * E / t_general, "same_bits": same_bits},
            "global_batch": GBATCH, "per_gpu_batch": B, "volume": [S, S, S], "euler_steps": E,
            "parallelism": f"batch-sharded x{world}, no data-path collective in expmap; the atlas step "
                           "(atlas_step below) all-reduces the atlas gradient over RCCL",
            "max_abs_displacement_vox": hmax,
        },
    }
    del m
    torch.cuda.empty_cache()
    atlas, epoch = None, None
    if not args.no_atlas:
        atlas = atlas_leg(lm, dev, world, rank, args, force_dist=force_dist)
    if not args.no_epoch:
        epoch = atlas_epoch_leg(lm, dev, world, rank, args, force_dist=force_dist)
    if rank == 0:
        if atlas is not None:
            result["atlas_step"] = atlas
        if epoch is not None:
            result["atlas_epoch"] = epoch
        if world == 1:
            result["rccl_world_size_1"] = {"requested": FORCE_DIST, "active": force_dist, "detail": force_dist_detail}
        # the dominant single kernel of the timed region (fluid_metric is three kernels and is reported
        # separately): all four candidates move 36 algorithmic bytes per voxel at C = 3
        cands = {
            "Ad_star": ("ad_star3_tile_kernel<float,512,2,5,2>", "lago::ad_star3_tile_kernel<float"),
            "compose": ("compose3_window_kernel<512,8,false>", "lago::compose3_window_kernel<512"),
            "interp_forward": ("interp_fwd3_unroll_kernel<float,false,2,true> (C=3)", "lago::interp_fwd3_unroll_kernel<float"),
            "jacobian_times_vectorfield_forward": ("jtv_fwd_kernel<float,3,true,false>", "lago::jtv_fwd_kernel<float, 3, true"),
        }
        present = {n: ksum[n] for n in cands if n in ksum and ksum[n]["launches"]}
        if present:
            op = max(present, key=lambda n: present[n]["total_ms"])
            k, (kname, prefix) = present[op], cands[op]
            bytes_per_launch = 36.0 * V  # 4*(3 + 3 + 3) bytes per voxel
            ach = bytes_per_launch / (k["mean_ms"] * 1e-3) / 1e9
            # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, collected
            # separately with rocprofv3 --pmc and condensed by tools/pmc_traffic.py into profiles/)
            traffic, tsrc = None, None
            for tname in ("r06_traffic_expmap.json", "r05_traffic_expmap.json", "r04_traffic_expmap.json", "r03_traffic_expmap.json", "r02_traffic_expmap.json", "r01_traffic.json"):
                tpath = os.path.join(ROOT, "profiles", tname)
                if os.path.exists(tpath) and B == 32 and S == 128:
                    for name, rec in json.load(open(tpath)).items():
                        if name.startswith(prefix):
                            traffic, tsrc = rec["traffic_bytes"], f"profiles/{tname} (rocprofv3 --pmc, same workload)"
                            break
                if traffic is not None:
                    break
            result["roofline"] = {
                "kernel": kname, "op": op, "bound": "hbm", "achieved": ach,
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
                "traffic_source": tsrc,
                "bytes_per_launch": bytes_per_launch, "mean_launch_ms": k["mean_ms"], "launches": k["launches"],
                "measured_in": "the single-stream pass that follows the timed region in this process (config.single_stream): "
                               "HIP events on the launch stream around each of its launches",
                # the same kernel inside the timed region itself: with the two-stream default each launch covers one
                # sub-batch and runs BESIDE the other stream's kernels, so its duration is an upper bound of its own time
                "timed_region": None if op not in ksum_timed else {
                    "mean_launch_ms": ksum_timed[op]["mean_ms"], "launches": ksum_timed[op]["launches"],
                    "bytes_per_launch": bytes_per_launch / (default_streams if split_active else 1),   # (mean over the parts: exact for even per-rank batches)
                    "frac": bytes_per_launch / (default_streams if split_active else 1) / (ksum_timed[op]["mean_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "streams": default_streams if split_active else 1},
                "others": {n: {"mean_launch_ms": present[n]["mean_ms"],
                               "frac": 36.0 * V / (present[n]["mean_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS}
                           for n in present if n != op},
            }
        total_ms = 1e3 * T
        result["breakdown_ms_per_step"] = {n: s["total_ms"] / args.steps for n, s in ksum_timed.items()}
        result["breakdown_ms_per_step"]["wall"] = total_ms / args.steps
        result["breakdown_ms_per_step"]["note"] = ("timed region; sums of per-launch HIP-event durations over BOTH streams "
                                                   "(they overlap in time, so they add up to more than the wall)"
                                                   if split_active else "timed region, one stream")
        result["breakdown_ms_per_step_single_stream"] = {n: s["total_ms"] / args.steps for n, s in ksum.items()}
        result["breakdown_ms_per_step_single_stream"]["wall"] = 1e3 * t_single
        if not args.no_micro and world == 1:
            result["interp_splat"] = micro_interp_splat(ext, dev, S)
            torch.cuda.empty_cache()
            result["fluid"] = micro_fluid(lm, dev, S)
            torch.cuda.empty_cache()
            result["other_ops"] = micro_ops(lm, dev, S)
            torch.cuda.empty_cache()
            result["atlas_step_128"] = micro_atlas_step(lm, dev, S)
            torch.cuda.empty_cache()
            result["brain_grid"] = micro_brain_grid(lm, dev)
            torch.cuda.empty_cache()
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(S, E, args.cpu_sample_batch)
            try:
                result["cpu_baseline"]["reference_cpu_path"] = reference_cpu_path(lm, dev, S)
            except Exception as e:  # the checker's absence must not cost the line
                result["cpu_baseline"]["reference_cpu_path"] = {"available": False, "error": repr(e)}
        if SHARE_GPU:
            result["debug_shared_gpu"] = "LAGO_BENCH_SHARE_GPU=1: all ranks on one GPU over gloo -- a plumbing check, NOT a measurement"
        emit(json.dumps(result))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
