#!/usr/bin/env python3
"""interp_backward C = 1 (d_I + d_u) at 8 x 128^3 on fields of increasing roughness u = a * randn:
the shipped sheared window (margins 1, 1, 4), wider margins through the same kernel (lago_tuning.splat_shear), the
general tiled window and the reference's global atomics; bench.py's smooth field for comparison."""
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
I = gaussian_blur(torch.randn((B, 1, S, S, S), device=dev, generator=g), 2.0)
go = torch.randn((B, 1, S, S, S), device=dev, generator=g)
us = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
us = us * (4.0 / us.abs().max())
noise = torch.randn((B, 3, S, S, S), device=dev, generator=g)
fields = [("smooth(4 vox)", us)] + [(f"a={a}", a * noise) for a in (0.25, 0.5, 1.0, 2.0)] + [("smooth+0.5", us + 0.5 * noise)]
default = ext.default_tuning()
configs = [("shipped 8,6 m1,1,4", dict()),
           ("4x4 m2,2,4", dict(splat_shear=[1, 4, 4, 0, 2, 2, 4, 1024])),
           ("4x4 m2,2,6", dict(splat_shear=[1, 4, 4, 0, 2, 2, 6, 1024])),
           ("3x3 m3,3,8", dict(splat_shear=[1, 3, 3, 0, 3, 3, 8, 1024])),
           ("2x2x64 m3,3,8", dict(splat_shear=[1, 2, 2, 64, 3, 3, 8, 1024])),
           ("tiled window", dict(splat_shear=[0, 8, 6, 0, 1, 1, 4, 1024])),
           ("global atomics", dict(splat_mode=0))]
V = B * S ** 3
ref = {}
print(f"{'field':16s} " + " ".join(f"{c[0]:>20s}" for c in configs))
for name, u in fields:
    row = f"{name:16s} "
    for cname, kw in configs:
        ext.tune(**default)
        ext.tune(**kw)
        try:
            dI, du = ext.interp_backward(go, I, u, 1.0, True, True)
            if name not in ref:
                ref[name] = (dI, du)
            ok = torch.equal(du, ref[name][1]) and float((dI - ref[name][0]).abs().max() / ref[name][0].abs().max()) < 1e-5
            t, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, True), reps=6, warm=3)
            row += f" {t*1e3:9.1f} us ({36.0*V/t/1e9/8.0:.3f}){'' if ok else ' !!'}"
        except RuntimeError as e:
            row += f" {'error':>20s}"
    print(row, flush=True)
ext.tune(**default)
