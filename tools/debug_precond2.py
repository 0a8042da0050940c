#!/usr/bin/env python3
"""Per-call accuracy of FluidMetric.sharp / flat inside one lddmm_step: HIP float32 against HIP float64, call by call
(input error, output error, and the same float32 operator applied to the float64 run's input)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import lagomorph_amd as lm
from oracle import lago_oracle as orc

z = np.load(sys.argv[1])
base, imgs, m = torch.from_numpy(z["base"]), torch.from_numpy(z["imgs"]), torch.from_numpy(z["m"])
B, steps = int(z["B"]), int(z["steps"])
kw = dict(integration_steps=steps, reg_weight=float(z["reg_weight"]), learning_rate_pose=1e-3, momentum_preconditioning=bool(z["precond"]))
calls = {}
def hook(tag):
    log = calls.setdefault(tag, [])
    o_sharp, o_flat = lm.FluidMetric.sharp, lm.FluidMetric.flat
    def sharp(self, x):
        y = o_sharp(self, x); log.append(("sharp", x.detach().cpu().double(), y.detach().cpu().double())); return y
    def flat(self, x, out=None):
        y = o_flat(self, x) if out is None else o_flat(self, x, out=out); log.append(("flat", x.detach().cpu().double(), y.detach().cpu().double())); return y
    lm.FluidMetric.sharp, lm.FluidMetric.flat = sharp, flat
    return o_sharp, o_flat
for tag, dt in (("hip32", torch.float32), ("hip64", torch.float64)):
    o = hook(tag)
    Ig = base.to(dt).cuda().requires_grad_(True)
    lm.lddmm_step(Ig, m.to(dt).cuda().clone(), imgs.to(dt).cuda(), lm.FluidMetric([0.1, 0.0, 0.01]), 3 * B, **kw)
    lm.FluidMetric.sharp, lm.FluidMetric.flat = o
rel = lambda a, b: float((a - b).abs().max() / max(float(b.abs().max()), 1e-300))
met = lm.FluidMetric([0.1, 0.0, 0.01])
for i, ((k, xi, yi), (_, x6, y6)) in enumerate(zip(calls["hip32"], calls["hip64"])):
    alone = (met.sharp if k == "sharp" else met.flat)(x6.float().cuda()).cpu().double()
    want = orc.fluid_metric_apply(x6.float().numpy(), [0.1, 0.0, 0.01], k == "sharp")
    print(f"call {i:2d} {k:5s}: input err {rel(xi, x6):.2e}  output err {rel(yi, y6):.2e}   operator alone on the f64 input: HIP f32 {rel(alone, y6):.2e}, "
          f"pocketfft f32 {rel(torch.from_numpy(want).double(), y6):.2e}   (max |in| {float(x6.abs().max()):.3g}, max |out| {float(y6.abs().max()):.3g}, "
          f"|mean in| per plane max {float(x6.mean(dim=(2, 3, 4)).abs().max()):.3g})")
k, xi, yi = calls["hip32"][7]; _, x6, y6 = calls["hip64"][7]
e = (xi - x6).abs()
big = (e > 0.05 * e.max()).nonzero()
print("call 7 input: voxels with more than 5 % of the largest error:", big.shape[0], "of", e.numel(), big[:8].tolist(), " largest", float(e.max()), " median", float(e.median()))
