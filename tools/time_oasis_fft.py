#!/usr/bin/env python3
"""`sharp` on the 176 x 208 x 176 brain grid (OASIS) and its permutations: the LDS-tiled passes with the radix-11 / radix-13
levels (round 6) against rocFFT's 3D plan + operator kernel (`fluid_mode 0`); persistent against one-shot zy kernels where
both exist (176 x 176 planes) and persistent against one-shot x pass.  Median of 20 calls, three alternations."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import time_op

ext = lm.lagomorph_ext
g = torch.Generator(device="cuda").manual_seed(3)
met = lm.FluidMetric([0.1, 0.0, 0.01])
variants = [("tuned", dict(fluid_mode=3)), ("tuned, zy one-shot", dict(fluid_mode=3, fluid_zy_persist=0)),
            ("tuned, x one-shot", dict(fluid_mode=3, fluid_xpass_persist=0)),
            ("tuned, x persistent forced", dict(fluid_mode=3, fluid_xpass_persist=2)), ("rocFFT", dict(fluid_mode=0))]
for sp, B in (((176, 208, 176), 1), ((176, 208, 176), 2), ((176, 208, 176), 4), ((176, 208, 176), 8), ((208, 176, 176), 4), ((176, 176, 208), 4), ((176, 176, 176), 4),
              ((128, 128, 128), 8)):
    x = torch.randn((B, 3) + sp, device="cuda", generator=g)
    ref = None
    t = {v[0]: [] for v in variants}
    with torch.no_grad():
        for rep in range(3):
            for name, kw in variants:
                ext.tune(**ext.default_tuning())
                ext.tune(**kw)
                before = ext.path_launches("fluid_lds")
                out = met.sharp(x)
                if name != "rocFFT":
                    assert ext.path_launches("fluid_lds") == before + 1, "not the LDS-tiled passes"
                    if ref is None:
                        ref = out.clone()
                    else:
                        assert torch.equal(out, ref), (sp, name)
                else:
                    err = float((out.double() - ref.double()).abs().max() / ref.double().abs().max())
                    assert err < 2e-6, err
                t[name].append(time_op(lambda: met.sharp(x), reps=20, warm=5)[0] * 1e3)
    ext.tune(**ext.default_tuning())
    vox = B * sp[0] * sp[1] * sp[2]
    print(f"{sp} x{B}:", flush=True)
    for name, _ in variants:
        m = sorted(t[name])[1]
        print(f"    {name:28s} {m:8.1f} us   {vox * 72.8 / m / 1e6:6.2f} TB/s of the single-pass ideal (72.8 B/voxel)", flush=True)

# the whole shoot on the brain grid: lddmm.expmap, 10 Euler steps, batch 8 (momenta scaled as bench.py scales them)
import bench

sp, B = (176, 208, 176), 8
m = bench.gaussian_blur(torch.randn((B, 3) + sp, device="cuda", generator=g), 4.0)
with torch.no_grad():
    m *= 2.5 / met.sharp(m).abs().max()
    for name, kw in (("tuned", dict(fluid_mode=3)), ("rocFFT", dict(fluid_mode=0))):
        ext.tune(**ext.default_tuning())
        ext.tune(**kw)
        ms = time_op(lambda: lm.expmap(met, m, num_steps=10), reps=5, warm=2)[0]
        print(f"expmap 10 steps, {B} x 3 x {sp}, {name}: {ms:.2f} ms = {B * sp[0] * sp[1] * sp[2] * 10 / ms / 1e6:.2f} Gvoxel-step/s", flush=True)
ext.tune(**ext.default_tuning())
