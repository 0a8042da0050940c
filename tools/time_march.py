#!/usr/bin/env python3
"""Timing of the C = 3 splat with d_u through one library (LAGO_HIP_LIBRARY selects an A/B build): modes 2 and 3 (mode 3 =
the marching-window kernel of `profiles/r06_march_window.patch`; on the shipped library every mode >= 2 is the shipped kernel),
batch B x S^3.  env: S (128), B (8), MODES ("2,3"), PARITY=1 compares mode 3 with mode 2."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur, time_op

ext = lm.lagomorph_ext
S, B = int(os.environ.get("S", 128)), int(os.environ.get("B", 8))
tag = sys.argv[1] if len(sys.argv) > 1 else ""
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1234)
sh = (B, 3, S, S, S)
u = gaussian_blur(torch.randn(sh, device=dev, generator=g), 8.0)
u = u * (4.0 / u.abs().max())
I = gaussian_blur(torch.randn(sh, device=dev, generator=g), 2.0)
go = torch.randn(sh, device=dev, generator=g)
V = B * S ** 3
forms = {"dt=1": lambda: ext.interp_backward(go, I, u, 1.0, True, True),
         "dt=-0.2": lambda: ext.interp_backward(go, I, u, -0.2, True, True)}
modes = [int(x) for x in os.environ.get("MODES", "2,3,4,5").split(",")]
if os.environ.get("PARITY") == "1":
    for name, fn in forms.items():
        ext.set_splat_shear_mc(2)
        rI, ru = fn()
        ext.set_splat_shear_mc(3)
        mI, mu = fn()
        print(f"{tag} parity {name}: d_u same bits {torch.equal(mu, ru)}, d_I rel {float((mI - rI).abs().max() / rI.abs().max()):.1e}")
for rep in range(2):
    for mode in modes:
        ext.set_splat_shear_mc(mode)
        line = f"{tag:8s} {S}^3 B={B} mode={mode}:"
        for name, fn in forms.items():
            t, _ = time_op(fn, reps=20, warm=10)
            line += f"  {name} {t*1e3:7.1f} us ({60.0*V/t/1e9/8.0:.3f})"
        print(line, flush=True)
ext.set_splat_shear_mc(2)
if os.environ.get("CENSUS") == "1":
    ext.set_splat_shear_mc(3)
    for name, fn in forms.items():
        dI, _ = fn()
        torch.cuda.synchronize()
        print(f"{tag} census {name}: x-misses {float(dI.view(B, -1)[:, 0].sum()):.0f} y-misses {float(dI.view(B, -1)[:, 1].sum()):.0f} of {V} voxels")
    ext.set_splat_shear_mc(2)
