#!/usr/bin/env python3
"""Benchmark of the LDDMM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Headline (BASELINE.json `metric`): LDDMM step voxels/sec, 3D 128^3 -- one "step" of this
benchmark is one `lddmm.expmap` call (10 Euler steps of the integrated EPDiff equation,
BASELINE configs[3]) over a batch of 32 momentum fields of 3x128^3 fp32 per GPU, inputs
resident in HBM.  value = (voxels * Euler steps) processed by all ranks / wall time.
The batch is sharded over ranks with no data-path collective ("weak": per-GPU batch fixed).

Also on the same JSON line:
  roofline      -- the dominant hand-written kernel of the timed region (3D interp forward,
                   C = 3): algorithmic bytes / mean launch time measured live with HIP events.
  cpu_baseline  -- the CPU oracle (scalar C port, 1 thread) on a bounded sample of the same
                   workload, timed on this host (rank 0, N = 1 only).
  interp_splat, fluid -- BASELINE configs[1] / configs[2] micro-measurements (interp HBM GB/s).
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
        fwd_med, _ = time_op(lambda: ext.interp_forward(I, uu, 1.0))
        r = {"fwd_ms": fwd_med, "fwd_GBps": 20.0 * V / fwd_med / 1e6}
        for mode in (1, 0):
            ext.set_splat_mode(mode)
            # the global-atomics leg is only a comparison figure: few repetitions
            bwd_med, _ = time_op(lambda: ext.interp_backward(go, I, uu, 1.0, True, True),
                                 reps=20 if mode == 1 else 4, warm=5 if mode == 1 else 1)
            tag = "lds" if mode == 1 else "atomics"
            r[f"bwd_{tag}_ms"] = bwd_med
            r[f"bwd_{tag}_GBps"] = 36.0 * V / bwd_med / 1e6
        ext.set_splat_mode(1)
        pair = r["fwd_ms"] + min(r["bwd_lds_ms"], r["bwd_atomics_ms"])
        r["pair_ms"] = pair
        r["pair_GBps"] = 56.0 * V / pair / 1e6
        r["pair_frac_of_hbm_peak"] = r["pair_GBps"] / HBM_PEAK_GBPS
        r["pair_Gvoxel_per_s"] = V / pair / 1e6
        res[label] = r
    return res


def micro_fluid(lm, dev, size, batch=8):
    """BASELINE configs[2]: FluidMetric sharp/flat on 3 x size^3 momentum fields, batch 8."""
    m = torch.randn((batch, 3, size, size, size), device=dev)
    met = lm.FluidMetric([0.1, 0.0, 0.01])
    V = batch * size ** 3
    with torch.no_grad():
        sharp_ms, _ = time_op(lambda: met.sharp(m), reps=10, warm=3)
        flat_ms, _ = time_op(lambda: met.flat(m), reps=10, warm=3)
        Fm = torch.view_as_real(torch.fft.rfftn(m, dim=(-3, -2, -1), norm="ortho").contiguous())
        k_inv, _ = time_op(lambda: lm.lagomorph_ext.fluid_operator(Fm, True, met.luts["cos"], met.luts["sin"], *met.params))
        k_fwd, _ = time_op(lambda: lm.lagomorph_ext.fluid_operator(Fm, False, met.luts["cos"], met.luts["sin"], *met.params))
    kbytes = 2 * Fm.numel() * 4
    return {
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
    out = {"workload": f"batch {batch} x 3x{size}^3 fp32, median of 10", "ops": {}}
    for name, (fn, bpv) in ops.items():
        med, _ = time_op(fn, reps=10, warm=3)
        out["ops"][name] = {"ms": med, "alg_bytes_per_voxel": bpv, "GBps": bpv * V / med / 1e6,
                            "frac_of_hbm_peak": bpv * V / med / 1e6 / HBM_PEAK_GBPS}
    return out


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
    return {"workload": f"lddmm_step (fwd + bwd + update) batch {batch} x {size}^3 fp32, 5 integration steps",
            "ms": med, "Gvoxel_per_s": batch * size ** 3 / med / 1e6}


def cpu_baseline(size, euler_steps, sample_batch=1):
    """Times the CPU oracle (test infrastructure) on a bounded sample of the headline workload --
    `sample_batch` volumes of size^3, one expmap of `euler_steps` steps -- on the host's cores (OpenMP,
    at most 64 threads) and on one thread.  The oracle stands in for lagomorph_ext only inside this
    function."""
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
        nthreads = max(1, min(os.cpu_count() or 1, 64))
        prev_threads = torch.get_num_threads()
        # all host cores (capped at 64 OpenMP threads) on the full sample, one thread on a quarter of it
        for tag, threads, batch in (("all", nthreads, sample_batch), ("one", 1, max(1, sample_batch // 4))):
            m = torch.from_numpy((0.01 * rng.standard_normal((batch, 3, size, size, size))).astype(np.float32))
            orc.set_threads(threads)
            torch.set_num_threads(threads)
            t0 = time.perf_counter()
            with torch.no_grad():
                lm.expmap(met, m, num_steps=euler_steps)
            dt = time.perf_counter() - t0
            runs[tag] = (batch * size ** 3 * euler_steps / dt, dt, batch, threads)
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
                  f"the voxel loops ({th_all} threads; FFTs by torch CPU/pocketfft), {dt_all:.1f} s",
        "one_thread": {"value": v_one, "cores": 1,
                       "sample": f"same, batch {b_one}, 1 thread, {dt_one:.1f} s"},
        "host_cpus": os.cpu_count(),
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="momentum fields per GPU")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--euler-steps", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-micro", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=16,
                    help="volumes in the all-cores CPU baseline sample (the one-thread sample is a quarter of it)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    ensure_built(local_rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import lagomorph_amd as lm

    ext = lm.lagomorph_ext
    B, S, E = args.batch, args.size, args.euler_steps
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
            ksum = kt.summary()
        hmax = h.abs().max().item()
        del h
    elapsed = torch.tensor([t1 - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    T = elapsed.item()
    vox_steps = world * B * S ** 3 * E * args.steps
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
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"lddmm.expmap, {E} Euler steps, batch {B} x 3x{S}^3 fp32 per GPU (BASELINE configs[3]); "
                        "value counts voxels x Euler steps",
            "global_batch": B * world, "per_gpu_batch": B, "volume": [S, S, S], "euler_steps": E,
            "parallelism": f"batch-sharded x{world}, no data-path collective",
            "max_abs_displacement_vox": hmax,
        },
    }
    if rank == 0:
        # the dominant single kernel of the timed region (fluid_metric is three kernels and is reported
        # separately): all four candidates move 36 algorithmic bytes per voxel at C = 3
        cands = {
            "Ad_star": ("ad_star3_unroll_kernel<float,2>", "lago::ad_star3_unroll_kernel<float"),
            "compose": ("compose3_unroll_kernel<float,2,false>", "lago::compose3_unroll_kernel<float"),
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
            tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
            if os.path.exists(tpath) and B == 32 and S == 128:
                for name, rec in json.load(open(tpath)).items():
                    if name.startswith(prefix):
                        traffic, tsrc = rec["traffic_bytes"], "profiles/r01_traffic.json (rocprofv3 --pmc, same workload)"
                        break
            result["roofline"] = {
                "kernel": kname, "op": op, "bound": "hbm", "achieved": ach,
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
                "traffic_source": tsrc,
                "bytes_per_launch": bytes_per_launch, "mean_launch_ms": k["mean_ms"], "launches": k["launches"],
                "others": {n: {"mean_launch_ms": present[n]["mean_ms"],
                               "frac": 36.0 * V / (present[n]["mean_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS}
                           for n in present if n != op},
            }
        total_ms = 1e3 * T
        result["breakdown_ms_per_step"] = {n: s["total_ms"] / args.steps for n, s in ksum.items()}
        result["breakdown_ms_per_step"]["wall"] = total_ms / args.steps
        if not args.no_micro and world == 1:
            result["interp_splat"] = micro_interp_splat(ext, dev, S)
            torch.cuda.empty_cache()
            result["fluid"] = micro_fluid(lm, dev, S)
            torch.cuda.empty_cache()
            result["other_ops"] = micro_ops(lm, dev, S)
            torch.cuda.empty_cache()
            result["atlas_step"] = micro_atlas_step(lm, dev, S)
            if S == 128:  # BASELINE configs[4] volume size (not a power of two: rocFFT-based fluid metric)
                torch.cuda.empty_cache()
                result["atlas_step_160"] = micro_atlas_step(lm, dev, 160)
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(S, E, args.cpu_sample_batch)
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
