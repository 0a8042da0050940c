#!/usr/bin/env python3
"""Where does the float32 error of a preconditioned matching step come from?  Records the input and the output of
FluidMetric.flat inside lddmm_step for HIP float32, HIP float64 and the oracle backend in float32 on one case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import lagomorph_amd as lm
from test_gpu_lddmm_step import oracle_backend, smooth_np

if len(sys.argv) > 1 and sys.argv[1].endswith(".npz"):
    z = np.load(sys.argv[1])
    base, imgs, m = torch.from_numpy(z["base"]), torch.from_numpy(z["imgs"]), torch.from_numpy(z["m"])
    B, steps = int(z["B"]), int(z["steps"])
    kw = dict(integration_steps=steps, reg_weight=float(z["reg_weight"]), learning_rate_pose=1e-3, momentum_preconditioning=bool(z["precond"]))
    PRESCALED = True
else:
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
    sp = (16, 8, 20); B = 2; steps = 4; d = 3
    base = torch.from_numpy(smooth_np(rng, (1, 1) + sp, 1.5)).float(); base = base / base.std()
    imgs = (base + 0.2 * torch.from_numpy(smooth_np(rng, (B, 1) + sp, 1.0)).float()).contiguous()
    m = torch.from_numpy(smooth_np(rng, (B, d) + sp, 1.5)).float()
    kw = dict(integration_steps=steps, reg_weight=1e-2, learning_rate_pose=1e-3, momentum_preconditioning=True)
    PRESCALED = False
rec = {}

def hooked(tag, FM):
    orig = FM.flat
    def flat(self, x, out=None):
        y = orig(self, x) if out is None else orig(self, x, out=out)
        rec[tag] = (x.detach().cpu().double().clone(), y.detach().cpu().double().clone())
        return y
    return orig, flat

with oracle_backend() as lmo:
    if not PRESCALED:
        m = (m * (1.5 / lmo.FluidMetric([0.1, 0.0, 0.01]).sharp(m).abs().max())).contiguous()
    o, f = hooked("orc32", lmo.FluidMetric); lmo.FluidMetric.flat = f
    Ic = base.clone().requires_grad_(True)
    mc, lc, rc = lmo.lddmm_step(Ic, m.clone(), imgs, lmo.FluidMetric([0.1, 0.0, 0.01]), 3 * B, **kw)
    lmo.FluidMetric.flat = o
res = {"orc32": mc.double()}
for tag, dt in (("hip32", torch.float32), ("hip64", torch.float64)):
    o, f = hooked(tag, lm.FluidMetric); lm.FluidMetric.flat = f
    Ig = base.to(dt).cuda().requires_grad_(True)
    mg, lg, rg = lm.lddmm_step(Ig, m.to(dt).cuda().clone(), imgs.to(dt).cuda(), lm.FluidMetric([0.1, 0.0, 0.01]), 3 * B, **kw)
    lm.FluidMetric.flat = o
    res[tag] = mg.cpu().double()
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
print("updated momenta vs hip64:  hip32 %.3g   orc32 %.3g   (max |m| %.3g, max |m_new - m| %.3g)" % (rel(res["hip32"], res["hip64"]), rel(res["orc32"], res["hip64"]), float(m.abs().max()), float((res["hip64"] - m.double()).abs().max())))
for tag in ("hip32", "orc32"):
    xi, yi = rec[tag]; x6, y6 = rec["hip64"]
    print(f"{tag}: flat input vs f64 {rel(xi, x6):.3g} (max |in| {float(x6.abs().max()):.3g});  flat output vs f64 {rel(yi, y6):.3g} (max |out| {float(y6.abs().max()):.3g})")
    # the operator alone: float32 flat of the float64 run's input against its float64 output
x6, y6 = rec["hip64"]
met = lm.FluidMetric([0.1, 0.0, 0.01])
print("flat alone on the float64 input: HIP f32 %.3g   torch-CPU f32 (oracle backend) %.3g" % (
    rel(met.flat(x6.float().cuda()).cpu().double(), y6),
    rel(__import__('oracle.lago_oracle', fromlist=['x']).fluid_metric_apply(x6.float().numpy(), [0.1, 0.0, 0.01], False).astype(np.float64).__class__ and torch.from_numpy(__import__('oracle.lago_oracle', fromlist=['x']).fluid_metric_apply(x6.float().numpy(), [0.1, 0.0, 0.01], False)).double(), y6)))
xi, _ = rec["hip32"]; x6, _ = rec["hip64"]
e = (xi - x6).abs()
big = (e > 0.1 * e.max()).nonzero()
print("voxels carrying more than a tenth of the largest error of the flat input:", big.shape[0], "of", e.numel(), "; at", big[:6].tolist())
# sample positions of the matching term at those voxels: x + h(x); distance of each coordinate to the nearest integer
met64 = lm.FluidMetric([0.1, 0.0, 0.01])
h64 = lm.expmap(met64, m.double().cuda(), num_steps=steps).cpu()
h32 = lm.expmap(lm.FluidMetric([0.1, 0.0, 0.01]), m.float().cuda(), num_steps=steps).cpu().double()
for idx in big[:4].tolist():
    n_, c_, i_, j_, k_ = idx
    pos = [float(v + h64[n_, a, i_, j_, k_]) for a, v in enumerate((i_, j_, k_))]
    print("  voxel", idx, "sample position (float64)", ["%.7f" % p for p in pos], " |h32 - h64| there", float((h32[n_, :, i_, j_, k_] - h64[n_, :, i_, j_, k_]).abs().max()))
