#!/usr/bin/env python3
"""Sheared-window splat (splat_shear_kernel) on the configs[1] workload (8 x 1x128^3 fp32): parity against the general
tiled kernel (d_u bit for bit, d_I to rounding), then a sweep of tile shapes / margins / threads."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur, time_op

ext = lm.lagomorph_ext
dev = torch.device("cuda")
S, B = int(os.environ.get("S", 128)), 8
g = torch.Generator(device=dev).manual_seed(1234)
I = gaussian_blur(torch.randn((B, 1, S, S, S), device=dev, generator=g), 2.0)
I = I / I.std()
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
u = u * (4.0 / u.abs().max())
go = torch.randn((B, 1, S, S, S), device=dev, generator=g)
V = B * S ** 3

ext.set_splat_shear(0)
ref_I, ref_u = ext.interp_backward(go, I, u, 1.0, True, True)
t_old, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, True), reps=20, warm=3)
t_old_nu, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, False), reps=20, warm=3)
print(f"general tiled kernel: {t_old*1e3:.1f} us (d_I + d_u), {t_old_nu*1e3:.1f} us (d_I only)")
res = []
for nt in (1024, 512):
    for (tx, ty, tz) in ((4, 8, 0), (8, 4, 0)):
        for (mx, mz) in ((1, 4),):
            cfg = dict(tx=tx, ty=ty, tz=tz, mx=mx, my=mx, mz=mz, nthreads=nt)
            ext.set_splat_shear(1, **cfg)
            try:
                dI, du = ext.interp_backward(go, I, u, 1.0, True, True)
                ok_u = torch.equal(du, ref_u)
                err = float((dI - ref_I).abs().max() / ref_I.abs().max())
                t, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, True), reps=8, warm=2)
                t2, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, False), reps=8, warm=2)
                res.append((t, t2, cfg, ok_u, err))
            except RuntimeError as e:
                print("cfg", cfg, "failed:", str(e)[:100])
res.sort(key=lambda r: r[0])
for t, t2, cfg, ok_u, err in res[:24]:
    print(f"{t*1e3:7.1f} us  (d_I only {t2*1e3:6.1f})  {36.0*V/t/1e6:6.0f} GB/s  d_u bits {'ok' if ok_u else 'DIFF'}  d_I relerr {err:.1e}  {cfg}")
print("worst:", [(round(r[0] * 1e3, 1), r[2]) for r in res[-3:]])
bad = [r for r in res if not r[3] or r[4] > 1e-5]
print("parity failures:", len(bad), bad[:3])
best = res[0][2]
rough = 2.0 * torch.randn_like(u)
for label, uu, dtt in (("rough", rough, 1.0), ("smooth dt=-0.2", u, -0.2), ("smooth dt=0.7 C=3", u, 0.7), ("smooth dt=1 C=3 bc", u, 1.0)):
    C = 3 if "C=3" in label else 1
    Ic = I if C == 1 else torch.randn((1 if "bc" in label else B, 3, S, S, S), device=dev, generator=g)
    gc = go if C == 1 else torch.randn((B, 3, S, S, S), device=dev, generator=g)
    ext.set_splat_shear(0)
    ext.tune(splat_mc=0)
    a = ext.interp_backward(gc, Ic, uu, dtt, True, True)
    ta, _ = time_op(lambda: ext.interp_backward(gc, Ic, uu, dtt, True, True), reps=5, warm=1)
    ext.tune(splat_mc=1)
    tm, _ = time_op(lambda: ext.interp_backward(gc, Ic, uu, dtt, True, True), reps=5, warm=1)
    ext.tune(splat_mc=0)
    ext.set_splat_shear(1, **best)
    ext.tune(splat_shear_mc=0)
    tb0, _ = time_op(lambda: ext.interp_backward(gc, Ic, uu, dtt, True, True), reps=5, warm=1)
    ext.tune(splat_shear_mc=1)
    b = ext.interp_backward(gc, Ic, uu, dtt, True, True)
    tb, _ = time_op(lambda: ext.interp_backward(gc, Ic, uu, dtt, True, True), reps=5, warm=1)
    ext.tune(splat_mc=1)
    print(f"{label}: general {ta*1e3:.1f} us (multi-channel form {tm*1e3:.1f}), sheared {tb*1e3:.1f} us (d_u per channel {tb0*1e3:.1f}), d_u bits {'ok' if torch.equal(a[1], b[1]) else 'DIFF'}, "
          f"d_I relerr {float((a[0]-b[0]).abs().max()/a[0].abs().max()):.1e}")
