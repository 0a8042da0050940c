#!/usr/bin/env python3
"""Sheared-window splat, multi-channel form (C = 3, d_I and d_u): tile sweep at S^3 (S from the environment, default
160 = the atlas step of BASELINE configs[4]); parity against the general tiled kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lagomorph_amd as lm
from bench import gaussian_blur, time_op

ext = lm.lagomorph_ext
dev = torch.device("cuda")
S, B, C = int(os.environ.get("S", 160)), int(os.environ.get("B", 8)), int(os.environ.get("C", 3))
g = torch.Generator(device=dev).manual_seed(1234)
I = gaussian_blur(torch.randn((B, C, S, S, S), device=dev, generator=g), 2.0)
u = gaussian_blur(torch.randn((B, 3, S, S, S), device=dev, generator=g), 8.0)
u = u * (3.0 / u.abs().max())
go = torch.randn((B, C, S, S, S), device=dev, generator=g)
V = B * S ** 3
ext.set_splat_shear(0)
ref_I, ref_u = ext.interp_backward(go, I, u, 1.0, True, True)
t0, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, True), reps=10, warm=3)
print(f"S={S} B={B} C={C}: general tiled kernel {t0*1e3:.1f} us")
cfgs = []
tiles = [tuple(int(x) for x in t.split("x")) for t in os.environ.get("TILES", "4x8x0,5x8x0,6x8x0,7x8x0,8x8x0,6x6x0,5x10x0,4x10x0,6x4x0,8x4x0,8x6x0,10x4x0,12x4x0,3x8x0").split(",")]
for nt in (1024,):
    for (tx, ty, tz) in tiles:
        cfgs.append(dict(tx=tx, ty=ty, tz=tz, mx=1, my=1, mz=4, nthreads=nt))
res = []
for cfg in cfgs:
    ext.set_splat_shear(1, **cfg)
    try:
        dI, du = ext.interp_backward(go, I, u, 1.0, True, True)
        ok_u = torch.equal(du, ref_u)
        err = float((dI - ref_I).abs().max() / ref_I.abs().max())
        t, _ = time_op(lambda: ext.interp_backward(go, I, u, 1.0, True, True), reps=10, warm=3)
        res.append((t * 1e3, cfg, ok_u, err))
    except Exception as e:  # noqa
        print("failed", cfg, str(e)[:100])
for t, cfg, ok_u, err in sorted(res, key=lambda r: r[0]):
    print(f"  {t:7.1f} us  {(24.0 + 12.0 * C) * V / t / 1e3:6.0f} GB/s  d_u {'bits ok' if ok_u else 'DIFFERS'}  d_I relerr {err:.1e}  {cfg}")
