#!/usr/bin/env python3
"""Takes a matching-step case dumped by tools/fuzz_step.py (LAGO_FUZZ_DUMP=<file>) apart: the float32 error of the updated
momenta against the same step in float64 through HIP, for the oracle backend and for HIP under each fluid-metric
implementation (`fluid_mode` 3 hand-written passes, 2, 1, 0 rocFFT-based), and of the preconditioning `flat` alone."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import lagomorph_amd as lm
from test_gpu_lddmm_step import oracle_backend

z = np.load(sys.argv[1])
base, imgs, m = (torch.from_numpy(z[k]) for k in ("base", "imgs", "m"))
B = int(z["B"])
kw = dict(integration_steps=int(z["steps"]), reg_weight=float(z["reg_weight"]), learning_rate_pose=1e-3,
          momentum_preconditioning=bool(z["precond"]))
print("case:", tuple(base.shape), tuple(m.shape), kw)
ext = lm.lagomorph_ext


def hip(dt):
    I = base.to(dt).cuda().requires_grad_(True)
    mg, l, r = lm.lddmm_step(I, m.to(dt).cuda().clone(), imgs.to(dt).cuda(), lm.FluidMetric([0.1, 0.0, 0.01]), 3 * B, **kw)
    return mg.detach().cpu().double()


truth = hip(torch.float64)
sc = float(truth.abs().max())
with oracle_backend() as lmo:
    Ic = base.clone().requires_grad_(True)
    mc, _, _ = lmo.lddmm_step(Ic, m.clone(), imgs, lmo.FluidMetric([0.1, 0.0, 0.01]), 3 * B, **kw)
print(f"oracle backend f32 vs HIP f64: {float((mc.double() - truth).abs().max()) / sc:.3e}")
for mode in (3, 2, 1, 0):
    ext.set_fluid_mode(mode)
    try:
        before = ext.path_launches()
        e = float((hip(torch.float32) - truth).abs().max()) / sc
        after = ext.path_launches()
        print(f"HIP f32 fluid_mode {mode}: {e:.3e}   paths {[k for k in after if after[k] != before[k] and k.startswith('fluid')]}")
    finally:
        ext.set_fluid_mode(3)
# the metric alone on the step's momentum grid: flat and sharp of a smooth field, float32 against float64
met = lm.FluidMetric([0.1, 0.0, 0.01])
x = m.cuda()
for name, op in (("sharp", met.sharp), ("flat", met.flat)):
    t = op(x.double())
    for mode in (3, 0):
        ext.set_fluid_mode(mode)
        try:
            e = float((op(x.float()).double() - t).abs().max() / t.abs().max())
        finally:
            ext.set_fluid_mode(3)
        print(f"{name} of the case's momenta, f32 vs f64, fluid_mode {mode}: {e:.3e}")
    with oracle_backend() as lmo:
        eo = float((getattr(lmo.FluidMetric([0.1, 0.0, 0.01]), name)(m.float()).double() - t.cpu()).abs().max() / t.abs().max())
    print(f"{name}, oracle backend (torch CPU FFT) f32 vs f64: {eo:.3e}")
