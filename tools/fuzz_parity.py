#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: random shapes (thin, ragged, 2D and 3D), dtypes, batch / channel counts,
displacement scales (sub-voxel to far out of range) through the bit-exact operators against the CPU oracle, and the
scatter-adds at north_star's bound.  Runs for `seconds` (default 120), prints the first mismatch or a count.
usage: python tools/fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import lagomorph_amd as lm
from oracle import lago_oracle as orc

ext = lm.lagomorph_ext
BIG = os.environ.get("LAGO_FUZZ_BIG") == "1"
TUNE = os.environ.get("LAGO_FUZZ_TUNE", "1") == "1"
CASE = {}
if BIG:
    orc.set_threads(min(32, os.cpu_count() or 1))


def run(budget=120.0, seed=0):
    """Returns (cases, worst error per scatter-add in units of its bound, cases decided by the float64 yardstick); raises
    SystemExit with a description at the first mismatch."""
    global n
    rng = np.random.default_rng(seed)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    host = lambda t: t.detach().cpu().numpy()
    t0, n, worst = time.time(), 0, {}


    yard = {}

    def close(name, got, want, tol, truth=None, sabs=None, nterms=None):
        """|got - want| <= tol * max|want|.  Where that fails for a float32 scatter-add or reduction, two float32 sums of
        the same n terms in different orders are being compared, and what float32 allows is the summation bound
        eps * sum|terms| of the element in question, not a fraction of the largest result (a border cell onto which a
        thousand clamped samples pile up; a d_T that cancels from terms a thousand times its size): with `truth` (the
        float64 oracle's result), `sabs` (sum of |terms| per element, or a bound of it) and `nterms` (how many terms an
        element sums, an estimate) HIP must be within (4 + 2 sqrt(n)) eps sum|terms| of the float64 result everywhere --
        the random-walk size of n float32 additions (strays of a rough field reach their cell one atomic at a time).
        The reference's own atomics add in arbitrary order too."""
        want = np.asarray(want, dtype=np.float64)
        g = host(got).astype(np.float64)
        err = float(np.abs(g - want).max()) if want.size else 0.0
        sc = max(float(np.abs(want).max()) if want.size else 0.0, 1e-30)
        if err > tol * sc and truth is not None and g.dtype is not None and tol > 1e-8:
            t = np.asarray(truth(), dtype=np.float64)
            S = np.asarray(sabs(), dtype=np.float64)
            nt = np.asarray(nterms(), dtype=np.float64) if nterms is not None else np.ones_like(S)
            excess = float(np.max(np.abs(g - t) - np.maximum((4.0 + 2.0 * np.sqrt(np.maximum(nt, 0.0))) * 1.2e-7 * S, tol * sc)))
            yard[name] = yard.get(name, 0) + 1
            if excess <= 0.0:
                return
            if os.environ.get("LAGO_FUZZ_DUMP"):
                np.savez(os.environ["LAGO_FUZZ_DUMP"], name=name, hip=g, orc32=want, f64=t, **{k: v for k, v in CASE.items()})
            raise SystemExit(f"MISMATCH {name}: HIP beyond the float32 summation bound by {excess:.3g} (case {n})")
        worst[name] = max(worst.get(name, 0.0), err / (tol * sc))
        if err > tol * sc:
            raise SystemExit(f"MISMATCH {name}: err {err:.3g} scale {sc:.3g} (case {n})")

    def bits(name, got, want):
        if not np.array_equal(host(got), want):
            raise SystemExit(f"BIT MISMATCH {name} (case {n}): max diff {np.abs(host(got) - want).max():.3g}")


    def one_case():
        d = int(rng.choice([2, 3, 3]))
        sp = tuple(int(x) for x in rng.choice([2, 3, 5, 8, 17, 33, 64, 70], size=d))
        if rng.random() < 0.15:
            sp = sp[:-1] + (int(rng.choice([128, 160, 200])),)
        if BIG:   # LAGO_FUZZ_BIG=1: shapes on which the tile / window / row-tile fast paths engage (up to ~2 M voxels)
            sp = tuple(int(x) for x in rng.choice([24, 32, 40, 48, 64, 96, 128, 160], size=d))
            while np.prod(sp) > 2_200_000:
                sp = tuple(sorted(sp))[:-1] + (int(sorted(sp)[-1] // 2),)
                sp = tuple(int(x) for x in rng.permutation(sp))
        N, C = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        dt_np = np.float32 if rng.random() < 0.7 else np.float64
        tol = 1e-5 if dt_np == np.float32 else 1e-12
        amp = float(rng.choice([0.3, 1.5, 6.0, 40.0]))
        bc = bool(rng.random() < 0.3)
        I = rng.standard_normal(((1 if bc else N), C) + sp).astype(dt_np)
        u = (amp * rng.standard_normal((N, d) + sp)).astype(dt_np)
        if rng.random() < 0.5:   # smooth-ish displacement: exercises the LDS windows
            from scipy.ndimage import uniform_filter
            u = uniform_filter(u, size=[1, 1] + [3] * d, mode="nearest").astype(dt_np)
        if rng.random() < 0.2:
            u = np.round(u).astype(dt_np)
        go = rng.standard_normal((N, C) + sp).astype(dt_np)
        dtv = float(rng.choice([1.0, -1.0, 0.37, -0.1]))
        CASE.clear(); CASE.update(I=I, u=u, go=go, dtv=dtv)
        bits("interp_forward", ext.interp_forward(dev(I), dev(u), dtv), orc.interp_forward(I, u, dtv))
        dI, du = ext.interp_backward(dev(go), dev(I), dev(u), dtv, True, True)
        oI, ou = orc.interp_backward(go, I, u, dtv, True, True)
        bits("interp_backward d_u", du, ou)
        f8 = lambda a: a.astype(np.float64)
        close("interp_backward d_I", dI, oI, tol, truth=lambda: orc.interp_backward(f8(go), f8(I), f8(u), dtv, True, True)[0],
              sabs=lambda: orc.interp_backward(np.abs(f8(go)), f8(I), f8(u), dtv, True, True)[0],
              nterms=lambda: 8.0 * orc.interp_backward(np.ones_like(f8(go)), f8(I), f8(u), dtv, True, True)[0] + 8.0)
        if min(sp) >= 2:
            v = rng.standard_normal((N, d) + sp).astype(dt_np)
            w = rng.standard_normal((N, d) + sp).astype(dt_np)
            for disp, tr in ((True, False), (False, True)):
                bits("jtv_forward", ext.jacobian_times_vectorfield_forward(dev(v), dev(w), disp, tr),
                     orc.jacobian_times_vectorfield_forward(v, w, disp, tr))
            bits("jtv_adjoint_forward", ext.jacobian_times_vectorfield_adjoint_forward(dev(v), dev(w)),
                 orc.jacobian_times_vectorfield_adjoint_forward(v, w))
            uu = u[:, :d]
            k = dt_np   # deform.py:53-55 with torch's rounding: scalars rounded to the tensor dtype, three roundings
            bits("compose", ext.compose(dev(uu), dev(v), dtv, -0.3), k(dtv) * uu + k(-0.3) * orc.interp_forward(v, uu, dtv))
            if True:   # (2D and 3D)
                want = orc.jacobian_times_vectorfield_forward(uu, orc.interp_forward(w, uu, 1.0), True, False)
                bits("Ad_star", ext.Ad_star(dev(uu), dev(w)), want)
            gv = rng.standard_normal((N, d) + sp).astype(dt_np)
            for disp, tr in ((True, False), (False, True), (True, True)):
                a, b = ext.jacobian_times_vectorfield_backward(dev(gv), dev(v), dev(w), disp, tr, True, True)
                oa, ob = orc.jacobian_times_vectorfield_backward(gv, v, w, disp, tr)
                bits("jtv_backward d_v", a, oa)
                bits("jtv_backward d_w", b, ob)
            a, b = ext.jacobian_times_vectorfield_adjoint_backward(dev(gv), dev(v), dev(w), True, True)
            oa, ob = orc.jacobian_times_vectorfield_adjoint_backward(gv, v, w)
            bits("jtv_adjoint_backward d_v", a, oa)
            bits("jtv_adjoint_backward d_w", b, ob)
            if n % 4 == 0:   # the fluid metric on a random shape: tuned passes, the fused 2D kernel or the generic FFT passes
                fs = tuple(int(x) for x in rng.choice([4, 6, 8, 9, 10, 12, 14, 15, 16, 22, 25, 26, 31, 32, 34, 58, 64], size=d))
                mm = rng.standard_normal((N, d) + fs).astype(dt_np)
                pr = [0.1, float(rng.choice([0.0, 0.05])), float(rng.choice([0.01, 0.3]))]
                met = lm.FluidMetric(pr)
                ft = 1e-5 if dt_np == np.float32 else 1e-11
                try:
                    close("sharp", met.sharp(dev(mm)), orc.fluid_metric_apply(mm, pr, True), ft)
                    close("flat", met.flat(dev(mm)), orc.fluid_metric_apply(mm, pr, False), ft)
                    sc = float(rng.choice([-0.1, -0.5, 1.0 / 3.0, 2.5]))   # the operator with an output factor: the bits of a multiply
                    if not torch.equal(met.sharp(dev(mm), out_scale=sc), met.sharp(dev(mm)) * sc):
                        raise SystemExit(f"case {n}: sharp(out_scale={sc}) differs from sharp * {sc} at {fs} {dt_np}")
                except RuntimeError as e:   # fluid_mode < 3 may select rocFFT, whose guard fails LOUDLY on a wrong transform
                    if "rocFFT returned a WRONG" not in str(e):
                        raise
                    yard["rocFFT wrong, call failed loudly"] = yard.get("rocFFT wrong, call failed loudly", 0) + 1
        A = (np.eye(d)[None] + 0.3 * rng.standard_normal((N, d, d))).astype(dt_np)
        T = (2.0 * rng.standard_normal((N, d))).astype(dt_np)
        CASE.update(A=A, T=T)
        bits("affine_interp_forward", ext.affine_interp_forward(dev(I), dev(A), dev(T)), orc.affine_interp_forward(I, A, T))
        gI, gA, gT = ext.affine_interp_backward(dev(go), dev(I), dev(A), dev(T), True, True, True)
        oI, oA, oT = orc.affine_interp_backward(go, I, A, T, True, True, True)
        tr = lambda k: (lambda: orc.affine_interp_backward(f8(go), f8(I), f8(A), f8(T), True, True, True)[k])
        # sum |terms|: d_I -- the splat of |go|; d_T / d_A -- |go| |grad I| (|x - o|) summed over the image: bounded by
        # sum|go| * 2 max|I| (* the largest half extent)
        sT = float(np.abs(go).sum()) * 2.0 * float(np.abs(I).max())
        close("affine d_I", gI, oI, tol, tr(0), sabs=lambda: orc.affine_interp_backward(np.abs(f8(go)), f8(I), f8(A), f8(T), True, True, True)[0],
              nterms=lambda: 8.0 * orc.affine_interp_backward(np.ones_like(f8(go)), f8(I), f8(A), f8(T), True, True, True)[0] + 8.0)
        close("affine d_A", gA, oA, tol, tr(1), sabs=lambda: np.full(oA.shape, sT * 0.5 * max(sp)), nterms=lambda: np.full(oA.shape, float(go[0].size)))
        close("affine d_T", gT, oT, tol, tr(2), sabs=lambda: np.full(oT.shape, sT), nterms=lambda: np.full(oT.shape, float(go[0].size)))
        out = tuple(int(x) for x in rng.choice([2, 3, 7, 16, 40], size=d))
        origin = [float((s - 1) * 0.5 + rng.normal()) for s in sp]
        spacing = [float((a - 1) / max(b - 1, 1) * rng.choice([1.0, 0.7, 1.6])) or 1.0 for a, b in zip(sp, out)]
        Ic = rng.standard_normal((N, C) + sp).astype(dt_np)
        bits("regrid_forward", ext.regrid_forward(dev(Ic), list(out), origin, spacing), orc.regrid_forward(Ic, list(out), origin, spacing))
        gb = rng.standard_normal((N, C) + out).astype(dt_np)
        close("regrid_backward", ext.regrid_backward(dev(gb), list(sp), list(out), origin, spacing),
              orc.regrid_backward(gb, list(sp), list(out), origin, spacing), tol,
              truth=lambda: orc.regrid_backward(f8(gb), list(sp), list(out), origin, spacing),
              sabs=lambda: orc.regrid_backward(np.abs(f8(gb)), list(sp), list(out), origin, spacing),
              nterms=lambda: 8.0 * orc.regrid_backward(np.ones_like(f8(gb)), list(sp), list(out), origin, spacing) + 8.0)

    while time.time() - t0 < budget:
        n += 1
        # every third case under a random combination of the library's sibling implementations (lago_tuning and the
        # shim's switches: speed only, never results -- which is what this checks)
        knobs = None
        if TUNE and n % 3 == 0:
            knobs = dict(gw=int(rng.integers(0, 2)), st=int(rng.integers(0, 2)), vk=int(rng.integers(0, 2)), sm=int(rng.integers(0, 2)),
                         fm=int(rng.integers(0, 4)), mc=int(rng.integers(0, 3)), sep=int(rng.integers(0, 2)), box=int(rng.integers(0, 2)),
                         lo=int(rng.integers(0, 2)))
            ext.set_gather_window(knobs["gw"]); ext.set_stencil_tile(knobs["st"]); ext.set_vector_kernels(knobs["vk"])
            ext.set_splat_mode(knobs["sm"]); ext.set_fluid_mode(knobs["fm"]); ext.set_splat_shear_mc(knobs["mc"])
            ext.REGRID_BACKWARD_SEPARABLE = knobs["sep"]; ext.tune(affine_box=knobs["box"]); ext.set_launch_order(knobs["lo"])
        try:
            one_case()
        except SystemExit as e:
            raise SystemExit(f"{e} [tuning {knobs}]")
        finally:
            if knobs is not None:
                ext.set_gather_window(1); ext.set_stencil_tile(1); ext.set_vector_kernels(1); ext.set_splat_mode(1)
                ext.set_fluid_mode(3); ext.set_splat_shear_mc(2); ext.REGRID_BACKWARD_SEPARABLE = 1; ext.tune(affine_box=1)
                ext.set_launch_order(1)
    return n, worst, yard


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    n, worst, yard = run(budget, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"{n} random cases in {budget:.0f} s, no mismatch; largest error in units of the bound: " +
          ", ".join(f"{k} {v:.3g}" for k, v in sorted(worst.items())) + f"; decided by the float64 yardstick: {yard}")
