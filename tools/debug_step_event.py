#!/usr/bin/env python3
"""Where a float32 `lddmm_step` leaves its float64 twin: every `interp_backward` / `interp_backward_fused` call of the
step is recorded in a HIP float32 and a HIP float64 run of the SAME case, and for each call the position gradient d_u
is compared voxel by voxel.  A CELL-FACE EVENT shows as a handful of voxels that carry the whole deviation (orders of
magnitude above the median) and whose sample position x + dt u lies within float32 rounding of an integer in some
component: the reference's gradient (include/interp.h:207-327) takes one-sided differences of the cell the floor
selects, so it is a different number on the two sides of a face, and x + dt u computed in float32 rounds onto the face
(floor = k) where the float64 value lies a hair below it (floor = k - 1).  `analyse` then re-runs the float32 step with
d_u of exactly those voxels taken from the float64 run and reports what is left of the mismatch.
usage: python tools/debug_step_event.py dump.npz      (a dump of tools/fuzz_step.py)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import lagomorph_amd as lm
from lagomorph_amd import lddmm

ext = lm.lagomorph_ext
FACE_ULPS = 2.0   # a sample counts as "on a face" within this many float32 ulps of its position


def analyse(base, imgs, m, B, kw, say=None):
    """Returns {"before": {key: err}, "after": {key: err}, "on_face": n, "off_face": n}: relative-to-max errors of the
    float32 step against the float64 step (both through HIP) for the updated momenta and the atlas gradient, before and
    after the d_u values of the on-face outlier voxels are taken from the float64 run; off_face = outlier voxels (error
    above 100 x the call's median and 1e-4 of its maximum) that are NOT within FACE_ULPS of a cell face."""
    say = say or (lambda *a: None)
    o_b, o_f = ext.interp_backward, ext.interp_backward_fused

    def run(dt, log=None, patch=None):
        n = [0]

        def wrap(kind, f):
            def g(grad_out, I, u, ds, *a, **k):
                r = f(grad_out, I, u, ds, *a, **k)
                i = n[0]
                n[0] += 1
                if patch is not None and i in patch:
                    idx, val = patch[i]
                    r[1].view(-1)[idx.cuda()] = val.to(r[1].dtype).cuda()
                if log is not None:
                    log.append((kind, u.detach().cpu().double(), float(ds), r[1].detach().cpu().double()))
                return r
            return g

        ext.interp_backward, ext.interp_backward_fused = wrap("interp_backward", o_b), wrap("interp_backward_fused", o_f)
        streams = lddmm.LDDMM_STEP_STREAMS
        lddmm.LDDMM_STEP_STREAMS = 1
        try:
            Ig = base.to(dt).cuda().requires_grad_(True)
            mg, lg, rg = lm.lddmm_step(Ig, m.to(dt).cuda().clone(), imgs.to(dt).cuda(), lm.FluidMetric([0.1, 0.0, 0.01]), 3 * B, **kw)
            torch.cuda.synchronize()
        finally:
            ext.interp_backward, ext.interp_backward_fused = o_b, o_f
            lddmm.LDDMM_STEP_STREAMS = streams
        return {"m": mg.detach().cpu().double(), "I.grad": Ig.grad.detach().cpu().double()}

    rel = lambda a, b: float((a - b).abs().max() / max(float(b.abs().max()), 1e-300))
    l64 = []
    r64 = run(torch.float64, l64)
    patch, out = {}, {"on_face": 0, "off_face": 0}
    # Calls in execution order: an event in one call contaminates every later one (its jump travels on through the
    # reverse sweep), so the FIRST call with outliers is classified and patched, the step re-run, and so on.
    for it in range(len(l64) + 1):
        l32 = []
        r32 = run(torch.float32, l32, patch)
        errs = {k: rel(r32[k], r64[k]) for k in r64}
        if it == 0:
            out["before"] = errs
            say(f"float32 against float64 through HIP: {errs}   ({len(l32)} interp-backward calls)")
        out["after"] = errs
        found = False
        for i, ((kind, u32, ds, d32), (_, u64, _, d64)) in enumerate(zip(l32, l64)):
            if d32.numel() == 0 or d32.shape != d64.shape:
                continue
            e = (d32 - d64).abs()
            sc, med = float(d64.abs().max()), float(e.median())
            big = (e.view(-1) > max(100 * med, 1e-4 * sc)).nonzero().view(-1)
            if big.numel() == 0:
                continue
            say(f"call {i} {kind}: d_u max err {float(e.max()) / sc:.3g} of max, median {med / sc:.3g}; outliers: {big.numel()} of {e.numel()}")
            sp = tuple(d64.shape[2:])
            on, off = [], 0
            for j in big.tolist():
                idx = np.unravel_index(j, tuple(d64.shape))
                vox = tuple(int(v) for v in idx[2:])
                pos = [vox[c] + ds * float(u64[(idx[0], c) + vox]) for c in range(len(sp))]
                # distance to the nearest integer in units of the float32 ulp of the position
                near = min(abs(p - round(p)) / float(np.spacing(np.float32(max(abs(p), 1.0)))) for p in pos)
                if near <= FACE_ULPS:
                    on.append(j)
                else:
                    off += 1
                if len(on) + off <= 6:
                    say(f"    item {idx[0]} component {idx[1]} voxel {vox}: d_u f32 {float(d32[idx]):+.6g} f64 {float(d64[idx]):+.6g}; sample position "
                        f"{['%.9f' % p for p in pos]}: {near:.2f} float32 ulps from a cell face")
            out["off_face"] += off
            if on and not off:
                on = torch.tensor(on, dtype=torch.long)
                if i in patch:
                    on = torch.unique(torch.cat([patch[i][0], on]))
                patch[i] = (on, d64.view(-1)[on])
                found = True
            break   # (the first call with outliers only; later ones are judged after the re-run)
        if not found:
            break
    out["on_face"] = sum(int(v[0].numel()) for v in patch.values())
    if patch:
        say(f"float32 step with d_u of those {out['on_face']} on-face values taken from the float64 run: {out['after']}")
    return out


if __name__ == "__main__":
    z = np.load(sys.argv[1])
    kw = dict(integration_steps=int(z["steps"]), reg_weight=float(z["reg_weight"]), learning_rate_pose=1e-3,
              momentum_preconditioning=bool(z["precond"]))
    base, imgs, m = torch.from_numpy(z["base"]), torch.from_numpy(z["imgs"]), torch.from_numpy(z["m"])
    print(f"case: image {tuple(base.shape[2:])} momenta {tuple(m.shape[2:])} B {int(z['B'])} {kw}")
    analyse(base, imgs, m, int(z["B"]), kw, say=print)
