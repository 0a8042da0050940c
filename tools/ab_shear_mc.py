#!/usr/bin/env python3
"""A/B in one process of the multi-channel forms of the sheared-window splat (lago_set_splat_shear_mc 2 / 1 / 0):
interp_backward with C = 3 and d_u, unit and non-unit step, plus the fused reverse-sweep forms.  env: S (128), B (8)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur, time_op

ext = lm.lagomorph_ext
S, B = int(os.environ.get("S", 128)), int(os.environ.get("B", 8))
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1234)
sh = (B, 3, S, S, S)
u = gaussian_blur(torch.randn(sh, device=dev, generator=g), 8.0)
u = u * (4.0 / u.abs().max())
I = gaussian_blur(torch.randn(sh, device=dev, generator=g), 2.0)
go = torch.randn(sh, device=dev, generator=g)
V = B * S ** 3
cases = {
    "interp_backward dt=1": lambda: ext.interp_backward(go, I, u, 1.0, True, True),
    "interp_backward dt=-0.2": lambda: ext.interp_backward(go, I, u, -0.2, True, True),
    "fused addgo dt=-0.2": lambda: ext.interp_backward_fused(go, I, u, -0.2, True, addgo=-0.2),
}
ref = {}
for rep in range(2):
    for mc in [int(x) for x in os.environ.get("MODES", "1,2").split(",")]:
        ext.set_splat_shear_mc(mc)
        for name, fn in cases.items():
            dI, du = fn()
            if name not in ref:
                ref[name] = (dI, du)
                same = "first"
            else:
                e = float((dI - ref[name][0]).abs().max() / ref[name][0].abs().max())
                same = f"d_u {'same bits' if torch.equal(du, ref[name][1]) else 'DIFFERS'}, d_I rel {e:.1e}"
            t, _ = time_op(fn, reps=30, warm=20)
            print(f"{S}^3 B={B} mc={mc} {name:24s}: {t*1e3:8.1f} us  {60.0*V/t/1e9:5.2f} TB/s alg   {same}", flush=True)
ext.set_splat_shear_mc(2)
